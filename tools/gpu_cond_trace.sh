cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ct; rm -rf $O; mkdir -p $O
EPOCHS=60 rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 tools/cond_bench.py 300 > /dev/null 2>&1
python tools/cond_trace.py $(ls $O/p/*/*kernel_trace.csv | head -1)
rm -rf $O/p
