"""Times the f64 GEMM kernel variants at the headline shapes (HIP events) and checks them against torch.matmul."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobocmf_amd import functional as F

dev = torch.device("cuda")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); st.record()
    for _ in range(iters): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / iters

M, N = 512, 65536
A = torch.tril(torch.randn(M, M, dtype=torch.float64, device=dev))
B = torch.randn(M, N, dtype=torch.float64, device=dev)
C = torch.empty(M, N, dtype=torch.float64, device=dev)
ref = A @ B
for tri, name, fl in ((1, "lower-tri A (A=Linv K)", M * M * N), (0, "dense", 2 * M * M * N)):
    F.gemm_f64(A, B, C, tri=tri)
    err = float((C - ref).abs().max() / ref.abs().max())
    ms = timeit(lambda: F.gemm_f64(A, B, C, tri=tri))
    print(f"NN {name:26s}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s (algorithmic)  relerr {err:.1e}")
AU = A.t().contiguous()
F.gemm_f64(AU, B, C, tri=2)
err = float((C - AU @ B).abs().max() / ref.abs().max())
ms = timeit(lambda: F.gemm_f64(AU, B, C, tri=2))
print(f"NN upper-tri A              : {ms:.3f} ms  {M*M*N / ms / 1e9:.1f} TFLOP/s  relerr {err:.1e}")
# NT: M x M output, contraction over N (no split-K through this entry: shows the per-CU rate)
P = torch.randn(M, 8192, dtype=torch.float64, device=dev); Q = torch.randn(M, 8192, dtype=torch.float64, device=dev)
O = F.gemm_f64(P, Q, trans_b=True)
err = float((O - P @ Q.t()).abs().max() / (P @ Q.t()).abs().max())
print(f"NT 512x512x8192 relerr {err:.1e}")
t = timeit(lambda: torch.matmul(A, B))
print(f"torch.matmul (rocBLAS dgemm, dense): {t:.3f} ms  {2*M*M*N / t / 1e9:.1f} TFLOP/s")
