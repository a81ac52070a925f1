#!/bin/bash
# GPU call 1 of round 3: full GPU suite, the bench line, per-dispatch counters in-step vs isolated.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3a
rm -rf $O && mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
rc=$?
tail -5 $O/pytest.log
if [ $rc -ge 124 ]; then echo "pytest killed ($rc)"; exit $rc; fi
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
python -c "import json;d=json.load(open('$O/bench.json'));print(d['value'],d['repeat_values'],d['roofline']['weighted_frac']);print(json.dumps(d.get('per_kernel_instep_ms'),indent=1))"
timeout -k 10 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/clk -- python3 tools/instep_clock_run.py > $O/clk.log 2>&1 || { echo "clk run failed"; tail -5 $O/clk.log; exit 1; }
python tools/instep_clock.py $(ls $O/clk/*/*counter_collection.csv | head -1) $(ls $O/clk/*/*kernel_trace.csv | head -1) > $O/instep_clock.md 2> $O/instep_clock.err
cat $O/instep_clock.md
rm -f $O/clk/*/*agent_info.csv
du -sh $O
