#!/bin/bash
# Round artefacts for profiles/ (run on the GPU box through gpurun): bench line, rocprofv3 kernel trace + stats of a
# single-stream run of the same step (no roofline launches mixed in), the GEMM variants in isolation under rocprofv3,
# PMC traffic of the dominant kernel, the other configurations.  Output: gpurun_out/final/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf $O && mkdir -p $O   # (also delete the LOCAL gpurun_out/final before a new call: merged files accumulate)
python bench.py > $O/bench_C3.json 2> $O/bench_C3.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline --no-roofline --no-dense-leg > $O/bench_C3_profiled_run.json 2> $O/prof.err
python tools/summarize_profile.py $(ls $O/prof/*/*kernel_trace.csv | head -1) $O/bench_C3_kernel_summary.md > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python3 bench.py --surrogates 1 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-dense-leg > $O/bench_1surrogate_profiled_run.json 2> $O/prof1.err
python tools/summarize_profile.py $(ls $O/prof1/*/*kernel_trace.csv | head -1) $O/single_stream_kernel_summary.md > /dev/null
python tools/step_timeline.py $(ls $O/prof1/*/*kernel_trace.csv | head -1) > $O/single_stream_step_timeline.txt
echo "kernel traces done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rk -- python3 tools/gemm_variants.py > $O/gemm_variants_under_rocprof.txt 2>&1
python tools/summarize_profile.py $(ls $O/rk/*/*kernel_trace.csv | head -1) $O/gemm_variants_kernel_summary.md > /dev/null
python tools/gemm_variants.py > $O/gemm_variants.txt 2>&1
python tools/tile_sweep.py > $O/tile_sweep.txt 2>&1
python tools/gemm_variants.py 1024 16384 > $O/gemm_variants_M1024.txt 2>&1
python tools/gemm_variants.py 512 8192 > $O/gemm_variants_layer0.txt 2>&1
python tools/gemm_variants.py 512 65536 > $O/gemm_variants_reflayout.txt 2>&1
echo "variants done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 tools/pmc_gemm.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 tools/pmc_gemm.py > /dev/null 2>&1
python tools/pmc_summarize.py $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) $O/pmc_gemm.json > /dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 tools/pmc_gemm.py > /dev/null 2>&1 || true
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/clk -- python3 tools/instep_clock_run.py > $O/clk.log 2>&1 && python tools/instep_clock.py $(ls $O/clk/*/*counter_collection.csv | head -1) $(ls $O/clk/*/*kernel_trace.csv | head -1) > $O/instep_clock.md || true
rm -rf $O/clk
echo "pmc done"
for a in "--config C1" "--config C2" "--config C5" "--config C4 --surrogates 1 --steps 3 --warmup 1 --repeats 1" "--surrogates 1" "--config C2 --eager" "--launch" "--dense-backward" "--no-prune-rows" "--no-prune-rows --dense-backward" "--surrogates 1 --no-prune-rows --dense-backward" "--config C5 --no-prune-rows --dense-backward"; do
  python bench.py $a --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('reference_layout'); print('$a', '|', round(d['value'],1), 'steps/s |', round(d['ms_per_step'],3), 'ms per bench step |', d['config']['surrogates_per_gpu'], 'surrogates | repeats', [round(v,1) for v in d['repeat_values']], '| reference layout', r and round(r['value'],1))"
done > $O/other_configs.txt
python tools/acq_bench.py 50 > $O/acq_bench.txt 2>&1
python tools/cond_bench.py 400 > $O/cond_bench.txt 2>&1
python tools/size_sweep.py > $O/size_sweep.txt 2>&1
./tools/mfma_peak > $O/mfma_peak.txt 2>&1 || true
python examples/example_acquisition_mfdgp_forrester.py > $O/forrester_walkthrough.txt 2>&1 || true
rocprofv3 --kernel-trace --output-format csv -d $O/c5 -- python3 bench.py --config C5 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-dense-leg > /dev/null 2>&1
python tools/step_timeline.py $(ls $O/c5/*/*kernel_trace.csv | head -1) > $O/C5_step_timeline.txt
rm -rf $O/c5
python tools/step_timeline.py $(ls $O/prof1/*/*kernel_trace.csv | head -1) | tail -40 > /dev/null
for cfg in C2 C1; do
  rocprofv3 --kernel-trace --output-format csv -d $O/p_$cfg -- python3 bench.py --config $cfg --surrogates 1 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-dense-leg > /dev/null 2>&1
  python tools/step_timeline.py $(ls $O/p_$cfg/*/*kernel_trace.csv | head -1) > $O/${cfg}_step_timeline.txt
  rm -rf $O/p_$cfg
done
# in-kernel stamps of the GEMM launches (diagnostic build: bash tools/build_variant.sh stamps -DGEMM_STAMPS beforehand)
if [ -f abtest/libstamps.so ]; then MOBOCMF_HIP_LIB=$PWD/abtest/libstamps.so python tools/gemm_stamps.py > $O/gemm_stamps.txt 2>&1 || true; fi
rm -rf $O/prof/*/*agent_info.csv $O/prof1/*/*agent_info.csv
du -sh $O
