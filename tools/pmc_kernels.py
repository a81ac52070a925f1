"""Runs the three GEMM variants at the headline shapes a few times (for rocprofv3 --pmc passes: LDS bank conflicts, MFMA
busy cycles): NN lower-triangular A, NN dense, NT (A A^T, K = 65536, the weighted-syrk operand layout)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mobocmf_amd import functional as F

dev = torch.device("cuda")
M, N = 512, 65536
A = torch.tril(torch.randn(M, M, dtype=torch.float64, device=dev))
B = torch.randn(M, N, dtype=torch.float64, device=dev)
C = torch.empty(M, N, dtype=torch.float64, device=dev)
O = torch.empty(M, M, dtype=torch.float64, device=dev)
for _ in range(3):
    F.gemm_f64(A, B, C, tri=1)
    F.gemm_f64(A, B, C, tri=0)
    F.gemm_f64(B, B, O, trans_b=True)
torch.cuda.synchronize()
