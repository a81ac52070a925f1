#!/bin/bash
# full GPU suite, then a C3 bench line (no CPU baseline) -- the round's standard check
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-suite}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
rc=$?
tail -6 $O/pytest_gpu.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('C3', round(d['value'],1), 'ref layout', d['reference_layout'] and round(d['reference_layout']['value'],1))"
for c in C1 C2; do timeout -k 10 200 python bench.py --config $c --steps 200 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', round(d['value'],1), 'ms/step/surrogate', round(1e3/d['per_surrogate_steps_per_s'],4))"; done
