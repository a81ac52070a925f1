#!/bin/bash
# the mid-size product kernel with other LDS row paddings (variant builds abtest/libpad_<A>_<B>.so: -DMS_PAD=<A> -DMD_PADB=<B>)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_mid32
{ echo "== product build (pads 2 / 2)"; timeout -k 10 100 python tools/mid32_bench.py 2>&1 | grep "M ="
for f in abtest/libpad_*.so; do echo "== $f"; MOBOCMF_HIP_LIB=$PWD/$f timeout -k 10 100 python tools/mid32_bench.py 2>&1 | grep "M ="; done; } > gpurun_out/pmc_mid32/pads.txt
cat gpurun_out/pmc_mid32/pads.txt
