#!/bin/bash
# phase stamps of the one-launch Cholesky's panel workgroup (abtest/libpcstamps.so built beforehand: tools/build_variant.sh pcstamps -DPC_STAMPS)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/chol
mkdir -p $O
export MOBOCMF_HIP_LIB=$PWD/abtest/libpcstamps.so
for n in 512 1024; do timeout -k 10 120 python tools/chol_stamps.py $n > $O/stamps$n.txt 2>&1 || exit 1; done
grep "last inverse row" $O/stamps1024.txt | tr -s ' ' | cut -d' ' -f2 | tr '\n' ' '; echo
grep "last inverse row" $O/stamps512.txt | tr -s ' ' | cut -d' ' -f2 | tr '\n' ' '; echo
