#!/bin/bash
# same-box A/B of two builds: the product library ("base") vs abtest/lib$1.so ("var"): launch variants at three shapes, then the C3 step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$1; O=gpurun_out/ab_$V; mkdir -p $O; rm -f $O/*.txt
for shape in "512 16384" "512 8192" "1024 16384"; do
  echo "== base $shape" >> $O/variants.txt
  timeout -k 10 200 python tools/gemm_variants.py $shape >> $O/variants.txt 2>&1 || exit 1
  echo "== var $shape" >> $O/variants.txt
  MOBOCMF_HIP_LIB=$PWD/abtest/lib$V.so timeout -k 10 200 python tools/gemm_variants.py $shape >> $O/variants.txt 2>&1 || exit 1
done
grep -E "^==|lower store   |lower colstats   |lower dA|weighted syrk" $O/variants.txt | cut -c1-100
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null | python -c "import sys,json; print('base', round(json.loads(sys.stdin.read())['value'], 2))" >> $O/ab.txt
  MOBOCMF_HIP_LIB=$PWD/abtest/lib$V.so timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null | python -c "import sys,json; print('var ', round(json.loads(sys.stdin.read())['value'], 2))" >> $O/ab.txt
done
cat $O/ab.txt
