#!/bin/bash
# one-launch Cholesky: accuracy against the launch-pair forms and rocSOLVER, timing, the layer tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/chol
mkdir -p $O
timeout -k 10 200 python tools/chol_accuracy.py > $O/chol_accuracy.txt 2>&1; rc=$?; cat $O/chol_accuracy.txt
[ $rc = 0 ] || exit $rc
timeout -k 10 200 python tools/chol_bench.py > $O/chol_bench.txt 2>&1; rc=$?; cat $O/chol_bench.txt
[ $rc = 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_hip_layer.py tests/test_hip_baselines.py tests/test_hip_model.py -x -q -m gpu > $O/pytest.log 2>&1
rc=$?
tail -5 $O/pytest.log
exit $rc
