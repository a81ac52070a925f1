#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final_pre
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/final_pre/pytest.log 2>&1
rc=$?
tail -3 gpurun_out/final_pre/pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " gpurun_out/final_pre/pytest.log | head -20; exit $rc; fi
bash tools/collect_profiles.sh
