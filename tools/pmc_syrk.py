"""Runs only the weighted syrk of the layer backward (H = A diag(w) A^T, 512 x 65536) a few times: used under
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) to measure its HBM traffic per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mobocmf_amd import functional as F

dev = torch.device("cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
A = torch.randn(M, N, dtype=torch.float64, device=dev)
w = torch.randn(N, dtype=torch.float64, device=dev)
H = torch.empty(M, M, dtype=torch.float64, device=dev)
for _ in range(5):
    F.syrk_weighted(A, w, H)
torch.cuda.synchronize()
