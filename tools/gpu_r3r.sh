#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3r
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_entry_points.py -m gpu -q -x -k "mid_gemm" > $O/pytest_mid.log 2>&1
rc=$?
tail -2 $O/pytest_mid.log
if [ $rc -ne 0 ]; then grep -E "^E " $O/pytest_mid.log | head -20; exit $rc; fi
python tools/mid_k_sweep.py 2>&1 | grep -E "waves=|floor"
for w in 32 8; do
for mx in 512 1024; do
for a in "--surrogates 1" "--config C5" ""; do
  timeout -k 10 300 python bench.py $a --mid-gemm-waves $w --mid-gemm-max $mx --no-cpu-baseline --no-roofline --no-dense-leg > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python -c "
import json
d=json.load(open('$O/b.json')); print('waves=$w max=$mx $a |',round(d['value'],1),[round(v,1) for v in d['repeat_values']])"
done
done
done
