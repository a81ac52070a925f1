#!/bin/bash
# round 4: full GPU suite (tuning refactor + skewed 16x16x4 pipeline + new oracle tests), then A/B vs the 4x4x4 build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r4e}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
rc=$?
tail -15 $O/pytest_gpu.log
[ $rc = 0 ] || exit $rc
bash tools/gpu_r4c.sh ${1:-r4e}
