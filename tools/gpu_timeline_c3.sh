#!/bin/bash
# step timeline of ONE C3 / C5 surrogate on a single stream under rocprofv3 (the chain with the one-launch Cholesky)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tl
rm -rf $O && mkdir -p $O
for cfg in C3 C5; do
  rocprofv3 --kernel-trace --output-format csv -d $O/p_$cfg -- python3 bench.py --config $cfg --surrogates 1 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-dense-leg > /dev/null 2> $O/err_$cfg.txt || { tail -5 $O/err_$cfg.txt; exit 1; }
  python tools/step_timeline.py $(ls $O/p_$cfg/*/*kernel_trace.csv | head -1) > $O/${cfg}_step_timeline.txt
  rm -rf $O/p_$cfg
  grep "step span" $O/${cfg}_step_timeline.txt
done
