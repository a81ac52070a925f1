#!/bin/bash
# cooperative step: its tests, the quick sweep, the phase stamps (abtest/libcstamps.so built beforehand)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/coop
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_coop_step.py -x -q > $O/pytest.log 2>&1
rc=$?
tail -5 $O/pytest.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/coop_sweep.py --quick --wgs 0,8,16,32 --no-layer-path > $O/coop_sweep_quick.txt 2>&1; cat $O/coop_sweep_quick.txt
bash tools/gpu_coop_stamps.sh
