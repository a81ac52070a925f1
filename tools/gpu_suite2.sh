#!/bin/bash
# full GPU suite, conditioned-training timing, C1/C2/C3 bench lines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-suite2}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
rc=$?
tail -6 $O/pytest_gpu.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/cond_bench.py 400 2>&1 | grep -v amdgpu.ids | tee $O/cond_bench.txt
for c in C3 C1 C2; do timeout -k 10 200 python bench.py --config $c --steps 200 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', round(d['value'],1), 'ms/step/surrogate', round(1e3/d['per_surrogate_steps_per_s'],4))"; done | tee $O/bench_small.txt
