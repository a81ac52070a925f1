import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobocmf_amd import functional as F
dev = torch.device("cuda:0")
M, N = 512, 65536
ns = 3
As = [torch.tril(torch.randn(M, M, dtype=torch.float64, device=dev)) for _ in range(ns)]
Bs = [torch.randn(M, N, dtype=torch.float64, device=dev) for _ in range(ns)]
Ps = [torch.randn(M, 8192, dtype=torch.float64, device=dev) for _ in range(ns)]
ref = [F.gemm_f64(As[i], Bs[i], tri=1).clone() for i in range(ns)]
refd = [F.gemm_f64(As[i], Bs[i]).clone() for i in range(ns)]
reft = [F.gemm_f64(Ps[i], Ps[(i + 1) % ns], trans_b=True).clone() for i in range(ns)]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
bad = 0
for rep in range(10):
    outs = []
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            o1 = F.gemm_f64(As[i], Bs[i], tri=1)
            o2 = F.gemm_f64(Ps[i], Ps[(i + 1) % ns], trans_b=True)
            o3 = F.gemm_f64(As[i], Bs[i])
            outs.append((o1, o2, o3))
    torch.cuda.synchronize()
    for i, (o1, o2, o3) in enumerate(outs):
        e = [(o1 != ref[i]).sum().item(), (o2 != reft[i]).sum().item(), (o3 != refd[i]).sum().item()]
        if any(e):
            bad += 1
            print("rep", rep, "stream", i, "mismatching elements (tri, nt, dense):", e)
print("bad", bad)
