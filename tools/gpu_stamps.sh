#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/stamps; mkdir -p $O
for n in 16384 8192; do
  echo "######## M = 512, N' = $n" >> $O/stamps.txt
  MOBOCMF_HIP_LIB=$PWD/abtest/${STAMPLIB:-libstamps.so} timeout -k 10 200 python tools/gemm_stamps.py 512 $n 2>&1 | grep -v amdgpu.ids >> $O/stamps.txt || exit 1
done
cat $O/stamps.txt | head -120
