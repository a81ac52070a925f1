#!/bin/bash
# Builds libmobocmf_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result"
objs=""
pids=""
for f in gemm_f64 chol gram elementwise rff api; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ common.h -nt $f.o ] || [ ../../include/mobocmf_hip.h -nt $f.o ]; then
    rm -f $f.o                      # a failed compile must not leave a stale object for the link step
    $HIPCC $FLAGS -c $f.hip -o $f.o &
    pids="$pids $!"
  fi
  objs="$objs $f.o"
done
for p in $pids; do wait $p; done      # set -e: any failed compile aborts the build
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libmobocmf_hip.so $objs
echo "built $(pwd)/libmobocmf_hip.so"
