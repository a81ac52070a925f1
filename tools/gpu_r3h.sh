#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3h
rm -rf $O && mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_sparse_backward.py -m gpu -q -x > $O/pytest_sparse.log 2>&1
rc=$?
tail -3 $O/pytest_sparse.log
if [ $rc -ne 0 ]; then grep -E "^E |^tests.*(Error|FAILED)" $O/pytest_sparse.log | head -30; exit $rc; fi
for a in "--surrogates 1" "--config C5" "--config C2" "--config C1" ""; do
  timeout -k 10 300 python bench.py $a --no-cpu-baseline --no-roofline > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python -c "
import json
d=json.load(open('$O/b.json')); print('$a |',round(d['value'],1),[round(v,1) for v in d['repeat_values']], 'ref layout', d['reference_layout'] and round(d['reference_layout']['value'],1), d['dead_work']['panel_columns'], d['dead_work']['backward_active_fraction'])"
done
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
rc=$?
tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then grep -E "^E " $O/pytest.log | head -20; exit $rc; fi
timeout -k 10 600 python bench.py > $O/bench_full.json 2> $O/bench_full.err || { tail -5 $O/bench_full.err; exit 1; }
python -c "
import json
d=json.load(open('$O/bench_full.json')); print(d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['weighted_frac'], d['per_kernel_instep_ms']['kernels']); print(d['cpu_baseline']); print(d['parity'])"
