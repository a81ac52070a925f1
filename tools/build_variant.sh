#!/bin/bash
# Builds a variant of the library into abtest/lib<name>.so with extra compiler flags (diagnostic builds, A/B experiments):
#   bash tools/build_variant.sh stamps -DGEMM_STAMPS
# Run it with MOBOCMF_HIP_LIB=$PWD/abtest/lib<name>.so.  abtest/ is scratch (git-ignored) but travels with gpurun.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p abtest/obj_$name
objs=""
pids=""
for f in gemm_f64 chol gram elementwise rff tiny_step coop_step api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c mobocmf_amd/csrc/$f.hip -o abtest/obj_$name/$f.o &
  pids="$pids $!"
  objs="$objs abtest/obj_$name/$f.o"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abtest/lib$name.so $objs
echo "built abtest/lib$name.so"
