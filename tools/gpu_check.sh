#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/check
rm -rf $O && mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
rc=$?
tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $O/pytest.log | head -20; exit $rc; fi
for a in "--surrogates 1" "--config C5" "--config C2" "--config C1" ""; do
  timeout -k 10 300 python bench.py $a --no-cpu-baseline --no-roofline --no-dense-leg > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python -c "
import json
d=json.load(open('$O/b.json')); print('$a |',round(d['value'],1),[round(v,1) for v in d['repeat_values']])"
done
bash tools/gpu_timeline.sh > /dev/null 2>&1; grep "step span\|sum_partials_multi\|gram_bwd" gpurun_out/tl/timeline.txt | head
