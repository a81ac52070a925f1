#!/bin/bash
# round-end check on the GPU box: whole -m gpu suite, then the PMC passes of the dominant kernel and a conditioned-iteration
# timeline.  Output: gpurun_out/rend/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/rend
rm -rf $O && mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
rc=$?
tail -4 $O/pytest_gpu.log
[ $rc = 0 ] || exit $rc
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 tools/pmc_gemm.py > /dev/null 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 tools/pmc_gemm.py > /dev/null 2>&1 &&
python tools/pmc_summarize.py $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) $O/pmc_gemm.json &&
EPOCHS=60 rocprofv3 --kernel-trace --output-format csv -d $O/p_cond -- python3 tools/cond_bench.py 40 > /dev/null 2>&1 &&
python tools/step_timeline.py $(ls $O/p_cond/*/*kernel_trace.csv | head -1) -60 > $O/cond_iteration_timeline.txt &&
rm -rf $O/p_cond $O/pmc_fetch $O/pmc_write &&
python bench.py > $O/bench_C3.json 2> $O/bench_C3.err && cat $O/bench_C3.json
