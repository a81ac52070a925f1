#!/bin/bash
# Evidence for the one-launch step (profiles/r04_tiny_step.txt): C1 through both paths, the size sweep behind the
# eligibility rule, the phase stamps, conditioned training, the Forrester walk-through, a kernel trace of the C1 bench.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tinyprof
rm -rf $O && mkdir -p $O
{
echo "== bench.py --config C1 (the reference's own size: Forrester 1D, 2 fidelities, M = N = 16, S = 4), 500 steps x 5 repeats"
for a in "--config C1" "--config C1 --layer-path" "--config C1 --surrogates 1" "--config C1 --surrogates 1 --layer-path" "--config C1 --surrogates 8"; do
  timeout -k 10 200 python bench.py $a --steps 500 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a |', round(d['value'],1), 'ELBO steps/s |', round(d['ms_per_step']*1e3,1), 'us per bench step |', d['step_issue'], '| repeats', [round(v) for v in d['repeat_values']])"
done
echo
echo "== tools/tiny_sweep.py: ONE surrogate, one launch vs the layer path (HIP-graph replay); the rule's estimate = util/tiny_step.py estimated_us"
python tools/tiny_sweep.py 2>/dev/null
echo
echo "== tools/tiny_stamps.py (library built with -DTINY_STAMPS): microseconds per phase of one C1 surrogate, launch order; the last entry is Adam"
MOBOCMF_HIP_LIB=$PWD/abtest/libtstamps.so python tools/tiny_stamps.py C1 2>/dev/null
echo
echo "== tools/cond_bench.py 2000: conditioned training (N1), Forrester sizes, 3 surrogates, 50 Pareto points, 10 x~"
python tools/cond_bench.py 2000 2>/dev/null
echo
echo "== tools/acq_small_bench.py: acquisition phase at Forrester sizes (search = 200 projected-Adam iterations x 5 restarts, 2 evaluations of 6 models each)"
python tools/acq_small_bench.py 2>/dev/null
echo
echo "== examples/example_acquisition_mfdgp_forrester.py (the reference's walk-through at its own schedule)"
python examples/example_acquisition_mfdgp_forrester.py 2>/dev/null | grep -E "schedule|seconds|Pareto set|next evaluation"
} > $O/tiny_step.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --config C1 --steps 50 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-dense-leg > /dev/null 2> $O/prof.err
python tools/summarize_profile.py $(ls $O/prof/*/*kernel_trace.csv | head -1) $O/C1_one_launch_kernel_summary.md > /dev/null
rm -rf $O/prof
cat $O/tiny_step.txt | head -40
