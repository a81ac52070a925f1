#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3m
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_entry_points.py -m gpu -q -x -k "mid_gemm" > $O/pytest_mid.log 2>&1
rc=$?
tail -3 $O/pytest_mid.log
if [ $rc -ne 0 ]; then grep -E "^E " $O/pytest_mid.log | head -20; exit $rc; fi
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
rc=$?
tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $O/pytest.log | head -20; exit $rc; fi
for a in "--surrogates 1" "--config C5" "--config C2" ""; do
  timeout -k 10 300 python bench.py $a --no-cpu-baseline --no-roofline --no-dense-leg > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python -c "
import json
d=json.load(open('$O/b.json')); print('$a |',round(d['value'],1),[round(v,1) for v in d['repeat_values']])"
done
