#!/bin/bash
# step timeline of one surrogate (single stream) under rocprofv3: args = extra bench flags
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tl
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 bench.py --surrogates 1 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-dense-leg "$@" > $O/run.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python tools/step_timeline.py $(ls $O/p/*/*kernel_trace.csv | head -1) > $O/timeline.txt
rm -rf $O/p
tail -5 $O/timeline.txt
