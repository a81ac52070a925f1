#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3k
rm -rf $O && mkdir -p $O
for v in "" pad32 pad64; do
  if [ -n "$v" ]; then export MOBOCMF_HIP_LIB=$PWD/abtest/lib$v.so; else unset MOBOCMF_HIP_LIB; fi
  echo "== variant '$v'"
  timeout -k 10 300 python tools/tile_sweep.py 512x8192 512x16384 1024x8192 1024x16384 2>&1 | grep "M="
done
