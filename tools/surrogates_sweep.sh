#!/bin/bash
# per-surrogate step time against the number of surrogates (streams) per GPU
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in 1 2 3 4 5 6; do
  timeout -k 10 300 python bench.py --surrogates $n --no-cpu-baseline --no-roofline --no-dense-leg > gpurun_out/b_s.json 2> gpurun_out/b_s.err || { tail -5 gpurun_out/b_s.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/b_s.json')); print('$n surrogates |', round(d['value'],1), 'steps/s |', round($n*1000.0/d['value'],3), 'ms per surrogate step')"
done
