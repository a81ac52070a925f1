#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3g
rm -rf $O && mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_sparse_backward.py -m gpu -q -x > $O/pytest_sparse.log 2>&1
rc=$?
tail -3 $O/pytest_sparse.log
if [ $rc -ne 0 ]; then grep -E "^E |^tests.*(Error|FAILED)" $O/pytest_sparse.log | head -30; exit $rc; fi
for db in "" "--dense-backward"; do
  for a in "--surrogates 1" "--config C5" "--config C2" ""; do
    timeout -k 10 300 python bench.py $a $db --no-cpu-baseline --no-roofline > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
    python -c "
import json
d=json.load(open('$O/b.json')); print('$db $a |',round(d['value'],1),[round(v,1) for v in d['repeat_values']])"
  done
done
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
rc=$?
tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $O/pytest.log | head -20; exit $rc; fi
