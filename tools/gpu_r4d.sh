#!/bin/bash
# round 4: skewed 16x16x4 pipeline -- GEMM-level parity, then A/B vs the 4x4x4 build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r4d}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_layer.py tests/test_hip_sparse_backward.py tests/test_hip_edge_cases.py tests/test_hip_entry_points.py -x -q > $O/pytest_gemm.log 2>&1
rc=$?
tail -3 $O/pytest_gemm.log
[ $rc = 0 ] || exit $rc
bash tools/gpu_r4c.sh ${1:-r4d}
