#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3d
rm -rf $O && mkdir -p $O
for cfg in C2 C3 C5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/p_$cfg -- python3 bench.py --config $cfg --surrogates 1 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline > $O/run_$cfg.json 2> $O/run_$cfg.err || { tail -3 $O/run_$cfg.err; exit 1; }
  python tools/step_timeline.py $(ls $O/p_$cfg/*/*kernel_trace.csv | head -1) > $O/timeline_$cfg.txt
  tail -30 $O/timeline_$cfg.txt | head -3
  rm -rf $O/p_$cfg
done
