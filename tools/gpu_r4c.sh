#!/bin/bash
# round 4: 16x16x4 main loop A/B (variants at two shapes + C3 step), no tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r4c}
mkdir -p $O
rm -f $O/variants.txt $O/ab.txt
for shape in "512 16384" "512 8192" "1024 16384"; do
  echo "== new $shape" >> $O/variants.txt
  timeout -k 10 200 python tools/gemm_variants.py $shape >> $O/variants.txt 2>&1 || exit 1
  echo "== mi4 $shape" >> $O/variants.txt
  MOBOCMF_HIP_LIB=$PWD/abtest/libmi4.so timeout -k 10 200 python tools/gemm_variants.py $shape >> $O/variants.txt 2>&1 || exit 1
done
grep -E "^==|lower store   |lower colstats   |lower dA|weighted syrk" $O/variants.txt
for i in 1 2; do
  MOBOCMF_HIP_LIB=$PWD/abtest/libmi4.so timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null | python -c "import sys,json; print('mi4 ', round(json.loads(sys.stdin.read())['value'], 2))" >> $O/ab.txt
  timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null | python -c "import sys,json; print('new ', round(json.loads(sys.stdin.read())['value'], 2))" >> $O/ab.txt
done
cat $O/ab.txt
