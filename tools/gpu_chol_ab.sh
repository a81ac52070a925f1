#!/bin/bash
# one-launch Cholesky + inverse (potrf_cols 0) against the launch pair per 64 columns (4) in whole training steps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/chol
mkdir -p $O
: > $O/ab.txt
for cols in 0 4; do
  for cfg in "--config C3" "--config C3 --surrogates 1" "--config C5"; do
    timeout -k 10 300 python bench.py $cfg --potrf-cols $cols --steps 100 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('potrf_cols $cols  $cfg ', round(d['value'],1), 'steps/s', round(d['ms_per_step'],4), 'ms/step')" >> $O/ab.txt || exit 1
  done
done
cat $O/ab.txt
