#!/bin/bash
# the cooperative step: its tests, then the timing sweep.  Output: gpurun_out/coop/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/coop
rm -rf $O && mkdir -p $O
timeout -k 10 ${1:-600} python -m pytest tests/test_hip_coop_step.py -x -q > $O/pytest.log 2>&1
rc=$?
tail -15 $O/pytest.log
[ $rc = 0 ] || exit $rc
timeout -k 10 400 python tools/coop_sweep.py ${2:-} > $O/coop_sweep.txt 2>&1; cat $O/coop_sweep.txt
