#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "" gnoexp gnostore; do
  if [ -n "$v" ]; then export MOBOCMF_HIP_LIB=$PWD/abtest/lib$v.so; else unset MOBOCMF_HIP_LIB; fi
  bash tools/gpu_timeline.sh > /dev/null 2>&1
  echo "== variant '$v'"; grep "gram_fwd_kernel<1, 8, 8>\|gram_fwd_kernel<0, 8, 0>" gpurun_out/tl/timeline.txt | grep "4096 wg\|1024 wg" 
done
