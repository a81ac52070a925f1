#!/bin/bash
# the one-launch Cholesky with the workgroup counts a two-layer batch gets (25 trailing + 16 inverse) against other splits
# (diagnostic build abtest/libpcstamps.so: MOBOCMF_DEBUG_POTRF_NT / _NI)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/chol
export MOBOCMF_HIP_LIB=$PWD/abtest/libpcstamps.so
{ for a in "16 8" "20 8" "20 12" "25 8" "25 12" "32 12" "32 16" "21 8" "21 12" "14 10"; do set -- $a; echo "== $1 trailing + $2 inverse workgroups"; MOBOCMF_DEBUG_POTRF_NT=$1 MOBOCMF_DEBUG_POTRF_NI=$2 timeout -k 10 120 python tools/chol_bench.py 2>&1 | grep "n= 512\|n= 768\|n=1024"; done; } > gpurun_out/chol/wgs_split.txt
cat gpurun_out/chol/wgs_split.txt
