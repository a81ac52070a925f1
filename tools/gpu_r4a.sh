#!/bin/bash
# round 4, first call: clean v_mfma_f64_16x16x4 peak + baseline bench on this box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4a
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak 2>/dev/null
timeout -k 10 120 ./tools/mfma_peak > $O/mfma_peak.txt 2>&1 && timeout -k 10 120 ./tools/mfma_peak >> $O/mfma_peak.txt 2>&1
cat $O/mfma_peak.txt
timeout -k 10 400 python bench.py --steps 100 --warmup 10 > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
