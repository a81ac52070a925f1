#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for p in 0.0625 0.125 0.25 0.5 0.75; do
  timeout -k 10 300 python bench.py --top-fraction $p --no-cpu-baseline --no-roofline > gpurun_out/b_u.json 2> gpurun_out/b_u.err || { tail -5 gpurun_out/b_u.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/b_u.json')); print('top fraction $p |', round(d['value'],1), 'steps/s | reference layout', round(d['reference_layout']['value'],1) if d['reference_layout'] else None, '| columns', d['dead_work']['panel_columns'])"
done
