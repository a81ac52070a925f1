#!/bin/bash
# round 4: 16x16x4 main loop -- parity of the GEMM-level tests, then A/B of the launch variants and of the C3 step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4b
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_layer.py tests/test_hip_sparse_backward.py tests/test_hip_edge_cases.py tests/test_hip_entry_points.py -x -q > $O/pytest_gemm.log 2>&1
rc=$?
tail -3 $O/pytest_gemm.log
[ $rc = 0 ] || exit $rc
for shape in "512 16384" "512 8192" "1024 16384" "512 65536"; do
  echo "== mi16 $shape" >> $O/variants.txt
  timeout -k 10 200 python tools/gemm_variants.py $shape >> $O/variants.txt 2>&1 || exit 1
  echo "== mi4 $shape" >> $O/variants.txt
  MOBOCMF_HIP_LIB=$PWD/abtest/libmi4.so timeout -k 10 200 python tools/gemm_variants.py $shape >> $O/variants.txt 2>&1 || exit 1
done
for i in 1 2; do
  MOBOCMF_HIP_LIB=$PWD/abtest/libmi4.so timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null | python -c "import sys,json; print('mi4 ', round(json.loads(sys.stdin.read())['value'], 2))" >> $O/ab.txt
  timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null | python -c "import sys,json; print('mi16', round(json.loads(sys.stdin.read())['value'], 2))" >> $O/ab.txt
done
cat $O/ab.txt
