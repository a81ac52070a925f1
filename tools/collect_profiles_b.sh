#!/bin/bash
# Round artefacts for profiles/, part B: PMC traffic / SQ counters of the dominant kernel, the other configurations, the callers
# either side of the path.  Output: gpurun_out/finalb/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/finalb
rm -rf $O && mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 tools/pmc_gemm.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 tools/pmc_gemm.py > /dev/null 2>&1
python tools/pmc_summarize.py $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) $O/pmc_gemm.json > /dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 tools/pmc_gemm.py > /dev/null 2>&1 || true
echo "pmc done"
for a in "--config C1" "--config C1 --layer-path" "--config C2" "--config C5" "--config C4 --surrogates 1 --steps 3 --warmup 1 --repeats 1" "--surrogates 1" "--launch" "--dense-backward" "--no-prune-rows" "--no-prune-rows --dense-backward" "--surrogates 1 --no-prune-rows --dense-backward" "--config C5 --no-prune-rows --dense-backward"; do
  python bench.py $a --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('reference_layout'); print('$a', '|', round(d['value'],1), 'steps/s |', round(d['ms_per_step'],3), 'ms per bench step |', d['config']['surrogates_per_gpu'], 'surrogates | repeats', [round(v,1) for v in d['repeat_values']], '| reference layout', r and round(r['value'],1))"
done > $O/other_configs.txt
echo "configs done"
python tools/acq_bench.py 50 > $O/acq_bench.txt 2>&1
python tools/cond_bench.py 400 > $O/cond_bench.txt 2>&1
python tools/size_sweep.py > $O/size_sweep.txt 2>&1
python examples/example_acquisition_mfdgp_forrester.py > $O/forrester_walkthrough.txt 2>&1 || true
du -sh $O
