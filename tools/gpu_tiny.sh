#!/bin/bash
# the one-launch step: its tests, the fitter / BO-iteration tests that now train through it, C1 through both paths, the
# Forrester walk-through.  Output: gpurun_out/tiny/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tiny
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_tiny_step.py tests/test_hip_bo_iteration.py tests/test_hip_conditioned.py -x -q > $O/pytest.log 2>&1
rc=$?
tail -5 $O/pytest.log
[ $rc = 0 ] || exit $rc
for a in "--config C1" "--config C1 --layer-path" "--config C1 --surrogates 1" "--config C1 --surrogates 1 --layer-path" "--config C1 --surrogates 8"; do
  timeout -k 10 200 python bench.py $a --steps 500 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a |', round(d['value'],1), 'steps/s |', round(d['ms_per_step']*1e3,1), 'us per bench step |', d['step_issue'], '| repeats', [round(v) for v in d['repeat_values']])"
done | tee $O/bench_C1.txt
timeout -k 10 300 python examples/example_acquisition_mfdgp_forrester.py > $O/forrester_walkthrough.txt 2>&1; tail -8 $O/forrester_walkthrough.txt
