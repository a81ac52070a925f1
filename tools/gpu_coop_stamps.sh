#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/coop
mkdir -p $O
export MOBOCMF_HIP_LIB=$PWD/abtest/libcstamps.so
{ for a in "2 2 64 64 1 1 0" "2 2 128 512 8 1 16" "2 2 128 512 8 4 32"; do timeout -k 10 120 python tools/coop_stamps.py $a; echo; done; } > $O/coop_stamps.txt 2>&1
cat $O/coop_stamps.txt
