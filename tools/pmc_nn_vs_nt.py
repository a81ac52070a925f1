#!/usr/bin/env python3
"""Workload for the NN-vs-NT counter comparison: the lower-triangular NN panel product (plain store) and the weighted syrk at
M = 512, N' = 65536, five launches each (run under rocprofv3 --pmc ... --kernel-trace; tools/pmc_table.py makes the table)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mobocmf_amd import functional as F

dev = torch.device("cuda")
M, N = 512, 65536
L = torch.tril(torch.randn(M, M, dtype=torch.float64, device=dev))
B = torch.randn(M, N, dtype=torch.float64, device=dev)
C = torch.empty(M, N, dtype=torch.float64, device=dev)
w = torch.randn(N, dtype=torch.float64, device=dev)
H = torch.empty(M, M, dtype=torch.float64, device=dev)
for _ in range(5):
    F.gemm_f64_epilogue(L, B, C, 1, 0)
torch.cuda.synchronize()
for _ in range(5):
    F.syrk_weighted(B, w, H)
torch.cuda.synchronize()
