#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3q
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_entry_points.py -m gpu -q -x -k "mid_gemm" > $O/pytest_mid.log 2>&1
rc=$?
tail -2 $O/pytest_mid.log
if [ $rc -ne 0 ]; then grep -E "^E " $O/pytest_mid.log | head -20; exit $rc; fi
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from mobocmf_amd import functional as F
dev = torch.device("cuda")
def timeit(fn, iters=50):
    for _ in range(10): fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3
F.set_mid_gemm_max(1024)
for M in (512, 640, 768, 1024):
    A = torch.tril(torch.randn(M, M, dtype=torch.float64, device=dev)); B = torch.randn(M, M, dtype=torch.float64, device=dev)
    C = torch.empty(M, M, dtype=torch.float64, device=dev)
    row = []
    for tri, tb in ((0, False), (1, False), (0, True)):
        F.set_mid_gemm_waves(4); t4 = timeit(lambda: F.gemm_f64(A, B, C, tri=tri, trans_b=tb))
        F.set_mid_gemm_waves(8); t8 = timeit(lambda: F.gemm_f64(A, B, C, tri=tri, trans_b=tb))
        row.append("tri=%d tb=%d: 4 waves %.1f us, 8 waves %.1f us" % (tri, tb, t4, t8))
    print("M=%d | %s" % (M, " | ".join(row)), flush=True)
PY
for w in 8 4; do
for mx in 512 1024; do
for a in "--surrogates 1" "--config C5" ""; do
  timeout -k 10 300 python bench.py $a --mid-gemm-waves $w --mid-gemm-max $mx --no-cpu-baseline --no-roofline --no-dense-leg > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python -c "
import json
d=json.load(open('$O/b.json')); print('waves=$w max=$mx $a |',round(d['value'],1),[round(v,1) for v in d['repeat_values']])"
done
done
done
