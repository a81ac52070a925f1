import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobocmf_amd import functional as F
dev = torch.device("cuda")
M, N = 512, 1024
torch.manual_seed(0)
A = torch.tril(torch.randn(M, M, dtype=torch.float64, device=dev))
B = torch.randn(M, N, dtype=torch.float64, device=dev)
C = F.gemm_f64(A, B, tri=1)
ref = A @ B
err = (C - ref).abs()
print("max err", err.max().item())
e16 = err.reshape(M // 16, 16, N // 128, 128).amax(dim=(1, 3))
print((e16 > 1e-9).int())
