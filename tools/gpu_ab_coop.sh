#!/bin/bash
# A/B of two builds of the library on the cooperative step: $1 = abtest/lib<name>.so variant name
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/coop
mkdir -p $O
for rep in 1 2; do
  echo "== main build"; timeout -k 10 200 python tools/coop_sweep.py --quick --wgs 0 --no-layer-path 2>&1 | grep "one launch"
  echo "== variant $1"; MOBOCMF_HIP_LIB=$PWD/abtest/lib$1.so timeout -k 10 200 python tools/coop_sweep.py --quick --wgs 0 --no-layer-path 2>&1 | grep "one launch"
done | tee $O/ab_$1.txt
