import sys, torch
sys.path.insert(0, '.')
from mobocmf_amd import functional as F
dev = torch.device("cuda")
def timeit(fn, iters=100):
    for _ in range(10): fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3
F.set_mid_gemm_max(1024)
for waves in (4, 8, 32):
    F.set_mid_gemm_waves(waves)
    for M in (128, 512):
        row = []
        for K in (32, 64, 128, 256, 512, 1024):
            A = torch.randn(M, K, dtype=torch.float64, device=dev); B = torch.randn(K, M, dtype=torch.float64, device=dev)
            C = torch.empty(M, M, dtype=torch.float64, device=dev)
            row.append("K=%d %.1f" % (K, timeit(lambda: F.gemm_f64(A, B, C))))
        print("waves=%d M=N=%d us per launch: %s" % (waves, M, " | ".join(row)), flush=True)
# a trivially small kernel for the launch floor
x = torch.zeros(256, device=dev)
print("launch floor (x.add_(1) on 256 floats): %.1f us" % timeit(lambda: x.add_(1.0)))
