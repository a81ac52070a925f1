#!/bin/bash
# SQ counters of the one-launch step (bench.py --config C1): is it latency-bound, as DESIGN.md 3.5 says?  Separate passes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tinypmc
rm -rf $O && mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -- python3 bench.py --config C1 --steps 30 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-dense-leg > /dev/null 2> $O/$tag.err || echo "pass failed: $set"
  f=$(ls $O/$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "tiny_step" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v = sorted(v)
    print("%-24s median per launch %14.0f  (%d launches)" % (k, v[len(v) // 2], len(v)))
PY
  rm -rf $O/$tag
done | tee $O/tiny_pmc.txt
