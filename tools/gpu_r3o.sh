#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "" nomma noloadf noboth; do
  if [ -n "$v" ]; then export MOBOCMF_HIP_LIB=$PWD/abtest/lib$v.so; else unset MOBOCMF_HIP_LIB; fi
  python - <<PY
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
row = []
for M in (512, 1024):
    A = torch.randn(M, M, dtype=torch.float64, device=dev); B = torch.randn(M, M, dtype=torch.float64, device=dev)
    C = torch.empty(M, M, dtype=torch.float64, device=dev)
    row.append("M=%d %.1f us" % (M, timeit(lambda: F.gemm_f64(A, B, C))))
print("variant '$v':", " | ".join(row), flush=True)
PY
done
