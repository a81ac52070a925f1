#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3i
rm -rf $O && mkdir -p $O
timeout -k 10 600 python tools/r3_sweep.py > $O/sweep.txt 2>&1 || { tail -20 $O/sweep.txt; exit 1; }
cat $O/sweep.txt
for a in "--surrogates 1" "--config C5" ""; do
  timeout -k 10 300 python bench.py $a --no-cpu-baseline --no-roofline --no-dense-leg > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python -c "
import json
d=json.load(open('$O/b.json')); print('$a |',round(d['value'],1),[round(v,1) for v in d['repeat_values']])"
done
