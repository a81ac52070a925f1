#!/bin/bash
# cooperative step: all its tests, the fitter / conditioned / BO-iteration tests that may now train through it, bench C2 both ways
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/coop
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_coop_step.py tests/test_hip_tiny_step.py tests/test_hip_conditioned.py tests/test_hip_bo_iteration.py -x -q > $O/pytest3.log 2>&1
rc=$?
tail -12 $O/pytest3.log
[ $rc = 0 ] || exit $rc
for a in "--config C2" "--config C2 --layer-path" "--config C2 --surrogates 4" "--config C2 --surrogates 1"; do
  timeout -k 10 200 python bench.py $a --steps 300 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a |', round(d['value'],1), 'steps/s |', round(d['ms_per_step']*1e3,1), 'us per bench step |', d['step_issue'], '| repeats', [round(v) for v in d['repeat_values']])"
done | tee $O/bench_C2.txt
for a in "--config C1" "--config C1 --surrogates 1"; do
  timeout -k 10 200 python bench.py $a --steps 500 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a |', round(d['value'],1), 'steps/s |', round(d['ms_per_step']*1e3,1), 'us per bench step |', d['step_issue'], '| repeats', [round(v) for v in d['repeat_values']])"
done | tee $O/bench_C1.txt
