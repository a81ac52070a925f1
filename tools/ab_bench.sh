#!/bin/bash
# Same-box A/B of two library builds (box-to-box variation of the bench is ~4 %, so decisions need this):
#   1. here (build container):  bash tools/ab_bench.sh build [<git-rev, default HEAD>]   -> abtest/libold.so from that revision,
#                                                                                          the working tree's build stays in place
#   2. on the GPU box:          gpurun -- 'bash tools/ab_bench.sh run "--config C3" 3'   -> old / new, interleaved, N rounds
# abtest/ is scratch (git-ignored); delete it afterwards.
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
  rev=${2:-HEAD}
  mkdir -p abtest/src
  for f in gemm_f64 chol gram elementwise rff api; do git show $rev:mobocmf_amd/csrc/$f.hip > abtest/src/$f.hip; done
  git show $rev:mobocmf_amd/csrc/common.h > abtest/src/common.h
  mkdir -p abtest/include && git show $rev:include/mobocmf_hip.h > abtest/include/mobocmf_hip.h
  objs=""
  for f in gemm_f64 chol gram elementwise rff api; do
    sed -i 's|"../../include/mobocmf_hip.h"|"../include/mobocmf_hip.h"|' abtest/src/common.h abtest/src/$f.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -c abtest/src/$f.hip -o abtest/src/$f.o &
    objs="$objs abtest/src/$f.o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abtest/libold.so $objs
  bash mobocmf_amd/csrc/build.sh
  echo "built abtest/libold.so from $rev"
elif [ "$1" = "run" ]; then
  args=${2:-}
  n=${3:-2}
  for i in $(seq $n); do
    MOBOCMF_HIP_LIB=$PWD/abtest/libold.so python bench.py $args --steps 40 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('old', round(json.loads(sys.stdin.read())['value'], 2))"
    python bench.py $args --steps 40 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('new', round(json.loads(sys.stdin.read())['value'], 2))"
  done
else
  echo "usage: $0 build [rev] | run \"<bench args>\" [rounds]"; exit 2
fi
