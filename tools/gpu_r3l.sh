#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3l
rm -rf $O && mkdir -p $O
for w in 384 512; do
for a in "--surrogates 1" ""; do
  timeout -k 10 300 python bench.py $a --small-gemm-max $w --no-cpu-baseline --no-roofline --no-dense-leg > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python -c "
import json
d=json.load(open('$O/b.json')); print('small_gemm_max=$w $a |',round(d['value'],1),[round(v,1) for v in d['repeat_values']])"
done
done
