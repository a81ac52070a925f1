#!/bin/bash
# Round artefacts for profiles/ (run on the GPU box through gpurun): bench line, rocprofv3 kernel trace + stats of the
# same command, PMC traffic of the dominant kernel, the other configurations.  Output: gpurun_out/final/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf $O && mkdir -p $O   # (also delete the LOCAL gpurun_out/final before a new call: merged files accumulate)
python bench.py > $O/bench_C3.json 2> $O/bench_C3.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_C3_profiled_run.json 2> $O/prof.err
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 tools/pmc_gemm.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 tools/pmc_gemm.py > /dev/null 2>&1
echo "pmc done"
for a in "--config C1" "--config C2" "--config C5" "--config C4 --surrogates 1 --steps 3 --warmup 1" "--surrogates 1" "--config C2 --eager"; do
  python bench.py $a --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', '|', round(d['value'],1), 'steps/s |', round(d['ms_per_step'],3), 'ms per bench step |', d['config']['surrogates_per_gpu'], 'surrogates')"
done > $O/other_configs.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rk -- python3 tools/gemm_bench.py > $O/gemm_bench.txt 2>&1   # roofline kernel in isolation: stats must agree with the HIP-event time
python tools/acq_bench.py 50 > $O/acq_bench.txt 2>&1
python tools/cond_bench.py 400 > $O/cond_bench.txt 2>&1
python tools/size_sweep.py > $O/size_sweep.txt 2>&1
du -sh $O
