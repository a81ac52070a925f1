#!/bin/bash
# cooperative step: tests, quick sweep, phase stamps (abtest/libcstamps.so built beforehand)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/coop
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_coop_step.py -x -q > $O/pytest.log 2>&1
rc=$?
tail -5 $O/pytest.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/coop_sweep.py --quick --wgs 0,8,16,32,48 --no-layer-path > $O/coop_sweep_quick.txt 2>&1; cat $O/coop_sweep_quick.txt
export MOBOCMF_HIP_LIB=$PWD/abtest/libcstamps.so
{ for a in "2 2 64 64 1 1 0" "2 2 128 512 8 1 16" "2 2 128 512 8 4 32"; do timeout -k 10 120 python tools/coop_stamps.py $a; echo; done; } > $O/coop_stamps.txt 2>&1
cat $O/coop_stamps.txt
