#!/bin/bash
# GPU call: 64-row tiles -- parity, then A/B timing against 128-row tiles.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_entry_points.py tests/test_hip_layer.py tests/test_hip_edge_cases.py -m gpu -q -x > $O/pytest.log 2>&1
rc=$?
tail -4 $O/pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $O/pytest.log | head -20; exit $rc; fi
for tr in 128 64; do
  for shape in "512 65536" "512 8192" "1024 65536" "1024 8192"; do
    echo "== TILE_ROWS=$tr shape $shape" >> $O/variants.txt
    TILE_ROWS=$tr timeout -k 10 300 python tools/gemm_variants.py $shape 2>&1 | grep -E "lower|upper|syrk" >> $O/variants.txt || exit 1
  done
done
cat $O/variants.txt
for tr in 128 64; do
  timeout -k 10 300 python bench.py --tile-rows $tr --no-cpu-baseline --no-roofline > $O/bench_$tr.json 2> $O/bench_$tr.err || exit 1
  timeout -k 10 300 python bench.py --tile-rows $tr --no-cpu-baseline --no-roofline --surrogates 1 > $O/bench1_$tr.json 2>> $O/bench_$tr.err || exit 1
  timeout -k 10 300 python bench.py --tile-rows $tr --no-cpu-baseline --no-roofline --config C5 > $O/benchC5_$tr.json 2>> $O/bench_$tr.err || exit 1
  python -c "
import json
for n in ('bench','bench1','benchC5'):
    d=json.load(open('$O/%s_$tr.json'%n)); print('$tr',n,round(d['value'],1),[round(v,1) for v in d['repeat_values']])"
done
