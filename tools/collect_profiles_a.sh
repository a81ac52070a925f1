#!/bin/bash
# Round artefacts for profiles/, part A (run on the GPU box through gpurun): the bench line, rocprofv3 kernel trace + stats of
# the same command, a single-stream run of the step, the GEMM variants in isolation (with and without rocprofv3), step
# timelines of the other configurations and of a conditioned-training iteration.  Output: gpurun_out/final/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf $O && mkdir -p $O   # (also delete the LOCAL gpurun_out/final before a new call: merged files accumulate)
python bench.py > $O/bench_C3.json 2> $O/bench_C3.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline --no-dense-leg > $O/bench_C3_profiled_run.json 2> $O/prof.err
python tools/summarize_profile.py $(ls $O/prof/*/*kernel_trace.csv | head -1) $O/bench_C3_kernel_summary.md > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python3 bench.py --surrogates 1 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-dense-leg > $O/bench_1surrogate_profiled_run.json 2> $O/prof1.err
python tools/summarize_profile.py $(ls $O/prof1/*/*kernel_trace.csv | head -1) $O/single_stream_kernel_summary.md > /dev/null
python tools/step_timeline.py $(ls $O/prof1/*/*kernel_trace.csv | head -1) > $O/single_stream_step_timeline.txt
echo "kernel traces done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rk -- python3 tools/gemm_variants.py > $O/gemm_variants_under_rocprof.txt 2>&1
python tools/summarize_profile.py $(ls $O/rk/*/*kernel_trace.csv | head -1) $O/gemm_variants_kernel_summary.md > /dev/null
python tools/gemm_variants.py > $O/gemm_variants.txt 2>&1
python tools/gemm_variants.py 1024 16384 > $O/gemm_variants_M1024.txt 2>&1
python tools/gemm_variants.py 512 8192 > $O/gemm_variants_layer0.txt 2>&1
python tools/gemm_variants.py 512 65536 > $O/gemm_variants_reflayout.txt 2>&1
python tools/tile_sweep.py > $O/tile_sweep.txt 2>&1
echo "variants done"
for cfg in C5 C2 C1; do
  # (--layer-path: C1 would otherwise run as one launch per step -- profiles/r04_tiny_step.txt covers that; the timeline is the layer path's)
  rocprofv3 --kernel-trace --output-format csv -d $O/p_$cfg -- python3 bench.py --config $cfg --surrogates 1 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-dense-leg --layer-path > /dev/null 2>&1
  python tools/step_timeline.py $(ls $O/p_$cfg/*/*kernel_trace.csv | head -1) > $O/${cfg}_step_timeline.txt
  rm -rf $O/p_$cfg
done
EPOCHS=60 rocprofv3 --kernel-trace --output-format csv -d $O/p_cond -- python3 tools/cond_bench.py 40 > /dev/null 2>&1
# cond_bench.py ends with 40 graph-replayed then 40 eager conditioned iterations: -60 lands in the replayed ones
python tools/step_timeline.py $(ls $O/p_cond/*/*kernel_trace.csv | head -1) -60 > $O/cond_iteration_timeline.txt || true
rm -rf $O/p_cond
rm -rf $O/prof/*/*agent_info.csv $O/prof1/*/*agent_info.csv
du -sh $O
