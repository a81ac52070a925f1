#!/bin/bash
# round 5, first check: the changed one-launch tests, the one-rank RCCL test, a short bench line with the one-rank process group
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a
rm -rf $O && mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_hip_rccl_single_rank.py tests/test_hip_tiny_step.py tests/test_hip_entry_points.py tests/test_hip_conditioned.py -x -q -s > $O/pytest.log 2>&1
rc=$?
tail -8 $O/pytest.log
[ $rc = 0 ] || exit $rc
timeout -k 10 120 python tools/rccl_single_rank.py > $O/rccl_1rank.json 2> $O/rccl_1rank.err; tail -2 $O/rccl_1rank.json
timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-dense-leg > $O/bench_C3_short.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench_C3_short.json'))
print({k: d[k] for k in ('value','exchange_ms','exchange_first_ms','rccl_ranks','backend','librccl_mapped','rccl_error','finite')})"
