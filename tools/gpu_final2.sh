#!/bin/bash
# after publish_profiles.py: the bench line again (quoting the PMC traffic of the published kernel source)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final2
rm -rf $O && mkdir -p $O
python bench.py > $O/bench_C3.json 2> $O/bench_C3.err || { tail -5 $O/bench_C3.err; exit 1; }
python -c "
import json
d=json.load(open('$O/bench_C3.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])"


