#!/bin/bash
# Samples the shader clock (rocm-smi) while (a) the one-surrogate step and (b) the isolated GEMM loop run: evidence for the
# "in-step 0.335 ms vs isolated 0.307 ms" gap of the top-layer GEMMs.
cd "$(dirname "$0")/.."
sample() { for i in 1 2 3 4 5 6; do /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; sleep 0.4; done; }
echo "== idle"; /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -3
echo "== bench.py --surrogates 1 (steps)"
python bench.py --surrogates 1 --steps 1500 --warmup 3 --repeats 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1 &
P=$!; sleep 6; sample; wait $P
echo "== bench.py (3 surrogates)"
python bench.py --steps 600 --warmup 3 --repeats 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1 &
P=$!; sleep 6; sample; wait $P
echo "== tools/gemm_instep_vs_isolated.py (GEMM launches alone, looped)"
python - <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from mobocmf_amd import functional as F
dev = torch.device("cuda")
M, N = 512, 65536
L = torch.tril(torch.randn(M, M, dtype=torch.float64, device=dev)); B = torch.randn(M, N, dtype=torch.float64, device=dev)
C = torch.empty(M, N, dtype=torch.float64, device=dev); a = torch.randn(M, dtype=torch.float64, device=dev)
p1 = torch.empty(8, N, dtype=torch.float64, device=dev); p2 = torch.empty(8, N, dtype=torch.float64, device=dev)
t0 = time.time()
while time.time() - t0 < 9:
    for _ in range(200):
        F.gemm_f64_epilogue(L, B, C, 1, 1, colsq_part=p1, coldot_part=p2, avec=a)
    torch.cuda.synchronize()
PY
P=$!; sleep 5; sample; wait $P
