#!/bin/bash
# Builds libmobocmf_hip.so for gfx950 (cross-compiles without a GPU).
# An object is reused only when the hash of everything that went into it -- compiler version, flags, its source, common.h
# and the public header -- equals the stamp written next to it (file times say nothing after a checkout or a flag
# change).  `build.sh -B` rebuilds everything.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $EXTRA_HIPCC_FLAGS"
force=0
[ "$1" = "-B" ] && force=1
ccver=$($HIPCC --version 2>/dev/null | sha256sum | cut -c1-16)
objs=""
pids=""
for f in gemm_f64 chol gram elementwise rff tiny_step coop_step api; do
  want=$( (echo "$ccver $FLAGS"; cat $f.hip common.h small_step_common.h tile16.h ../../include/mobocmf_hip.h) | sha256sum | cut -c1-32)
  have=$(cat $f.o.stamp 2>/dev/null || true)
  if [ $force = 1 ] || [ ! -f $f.o ] || [ "$want" != "$have" ]; then
    rm -f $f.o $f.o.stamp           # a failed compile must not leave a stale object for the link step
    ( $HIPCC $FLAGS -c $f.hip -o $f.o && echo "$want" > $f.o.stamp ) &
    pids="$pids $!"
  fi
  objs="$objs $f.o"
done
for p in $pids; do wait $p; done      # set -e: any failed compile aborts the build
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libmobocmf_hip.so $objs
echo "built $(pwd)/libmobocmf_hip.so"
