"""Runs only the dominant kernel of the step -- A = L^-1 K_mn with the column-statistics epilogue, lower-triangular, M x N' x M
(argv: M N', default 512 16384 = the widest panel of the C3 step with dead rows pruned), launched exactly as
mobocmf_layer_forward launches it -- a few times: used under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)
to measure its HBM traffic per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mobocmf_amd import functional as F

dev = torch.device("cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
A = torch.tril(torch.randn(M, M, dtype=torch.float64, device=dev))
B = torch.randn(M, N, dtype=torch.float64, device=dev)
C = torch.empty(M, N, dtype=torch.float64, device=dev)
avec = torch.randn(M, dtype=torch.float64, device=dev)
p1 = torch.empty(4 * (M // 128), N, dtype=torch.float64, device=dev)
p2 = torch.empty(4 * (M // 128), N, dtype=torch.float64, device=dev)
for _ in range(5):
    F.gemm_f64_epilogue(A, B, C, 1, 1, colsq_part=p1, coldot_part=p2, avec=avec)
torch.cuda.synchronize()
