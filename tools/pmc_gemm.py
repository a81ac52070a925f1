"""Runs only the dominant kernel (A = L^-1 K_mn, 512 x 65536 x 512 lower-triangular) a few times: used under
rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) to measure its HBM traffic per launch."""
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
for _ in range(5):
    F.gemm_f64(A, B, C, tri=1)
torch.cuda.synchronize()
