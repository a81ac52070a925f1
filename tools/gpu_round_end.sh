#!/bin/bash
# round-end evidence on the GPU box (profiles/r05_*): whole -m gpu suite; the bench line; the kernel trace of the TIMED configuration
# only (pruned rows, 3 streams, HIP-graph replay: no reference-layout leg, no roofline launches, no CPU baseline in the trace); the
# PMC passes of the dominant kernel; the cooperative step's sweep and phase stamps; RCCL at one rank.  Output: gpurun_out/rend/
# the one-launch Cholesky's accuracy, timing, stamps and A/B.  Optional argument: "a" = the first half (suite, bench line, timed-only
# trace, PMC), "b" = the second half (small configurations, cooperative step, RCCL, Cholesky), "nosuite" = everything but the suite;
# a gpurun call is limited to 20 minutes, the whole script needs about that.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/rend
mkdir -p $O
if [ "$1" != "b" ]; then
if [ "$1" != "nosuite" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
  rc=$?
  tail -4 $O/pytest_gpu.log
  [ $rc = 0 ] || exit $rc
fi
# the record the driver will produce, with the CPU baseline (a few minutes)
timeout -k 10 600 python bench.py > $O/bench_C3.json 2> $O/bench_C3.err && tail -c 400 $O/bench_C3.json
# per-kernel time in the configuration that is timed (the program directly after `--`)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_timed -- python3 bench.py --no-dense-leg --no-roofline --no-cpu-baseline --no-rccl --steps 100 > $O/bench_C3_profiled_timed_only.json 2> /dev/null &&
python tools/summarize_profile.py $(ls $O/p_timed/*/*kernel_trace.csv | head -1) $O/bench_C3_timed_only_kernel_summary.md > /dev/null &&
cp $(ls $O/p_timed/*/*kernel_stats.csv | head -1) $O/bench_C3_timed_only_kernel_stats.csv
rm -rf $O/p_timed
# HBM traffic of the dominant kernel (separate passes, as the guide prescribes)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 tools/pmc_gemm.py > /dev/null 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 tools/pmc_gemm.py > /dev/null 2>&1 &&
python tools/pmc_summarize.py $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) $O/pmc_gemm.json
rm -rf $O/pmc_fetch $O/pmc_write
fi
[ "$1" = "a" ] && exit 0
# the one-launch steps: C1 / C2 through them and through the layer path, the cooperative step's sweep and phase stamps
for a in "--config C1" "--config C1 --layer-path" "--config C2" "--config C2 --layer-path" "--config C2 --surrogates 4" "--config C2 --surrogates 1"; do
  timeout -k 10 200 python bench.py $a --steps 300 --no-cpu-baseline --no-roofline --no-dense-leg --no-rccl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a |', round(d['value'],1), 'steps/s |', round(d['ms_per_step']*1e3,1), 'us per bench step |', d['step_issue'], '| repeats', [round(v) for v in d['repeat_values']])"
done | tee $O/bench_small_configs.txt
timeout -k 10 500 python tools/coop_sweep.py > $O/coop_sweep.txt 2>&1; tail -60 $O/coop_sweep.txt
if [ -f abtest/libcstamps.so ]; then
  { for a in "2 2 64 64 1 1 0" "2 2 128 512 8 1 0" "2 2 128 512 8 3 0"; do MOBOCMF_HIP_LIB=$PWD/abtest/libcstamps.so timeout -k 10 120 python tools/coop_stamps.py $a; echo; done; } > $O/coop_stamps.txt 2>&1
fi
timeout -k 10 120 python tools/rccl_single_rank.py > $O/rccl_1rank.json 2> $O/rccl_1rank.err; tail -c 300 $O/rccl_1rank.json

# the one-launch Cholesky + inverse: accuracy, time per call, phase stamps of the panel workgroup, whole steps against the launch pairs
timeout -k 10 200 python tools/chol_accuracy.py 2>&1 | grep -v amdgpu.ids > $O/chol_accuracy.txt
timeout -k 10 200 python tools/chol_bench.py 2>&1 | grep -v amdgpu.ids > $O/chol_bench.txt; cat $O/chol_bench.txt
if [ -f abtest/libpcstamps.so ]; then
  for n in 512 1024; do MOBOCMF_HIP_LIB=$PWD/abtest/libpcstamps.so timeout -k 10 120 python tools/chol_stamps.py $n 2>&1 | grep -v amdgpu.ids > $O/chol_stamps$n.txt; done
fi
for cols in 0 4; do
  for cfg in "--config C3" "--config C3 --surrogates 1" "--config C5"; do
    timeout -k 10 300 python bench.py $cfg --potrf-cols $cols --steps 100 --no-cpu-baseline --no-roofline --no-dense-leg --no-rccl 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('potrf_cols $cols  $cfg ', round(d['value'],1), 'steps/s', round(d['ms_per_step'],4), 'ms/step')"
  done
done | tee $O/chol_ab.txt
