import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobocmf_amd import functional as F
dev = torch.device("cuda:0")
M, N, S, d = 512, 8192, 8, 8
ns = 3
torch.manual_seed(0)
def mk():
    x = torch.rand(N, d, dtype=torch.float64, device=dev)
    f = torch.randn(N * S, dtype=torch.float64, device=dev)
    Zx = x[:M].clone(); zf = 0.1 * torch.randn(M, dtype=torch.float64, device=dev)
    hyp = torch.tensor([1, 1, 1, 0.01, 1] + [1.4] * d + [1.4] * d, dtype=torch.float64, device=dev)
    m = 0.1 * torch.randn(M, dtype=torch.float64, device=dev)
    LS = (0.1 * torch.eye(M, dtype=torch.float64, device=dev) + 0.01 * torch.tril(torch.randn(M, M, dtype=torch.float64, device=dev)))
    w = torch.randn(N * S, dtype=torch.float64, device=dev)
    return [x, f, Zx, zf, hyp, m, LS, w]
def run(p):
    x, f, Zx, zf, hyp, m, LS, w = p
    leaves = [t.detach().clone().requires_grad_(True) for t in (f, zf, hyp, m, LS)]
    mean, var, kl = F.layer_forward(x, leaves[0], Zx, leaves[1], leaves[2], leaves[3], leaves[4], 1, xdiv=S)
    loss = (w * mean).sum() + (w * w * var).sum() + 0.3 * kl
    loss.backward()
    return [mean.detach(), var.detach(), kl.detach()] + [t.grad for t in leaves]
P = [mk() for _ in range(ns)]
ref = [run(p) for p in P]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
names = ["mean", "var", "kl", "g_f", "g_zf", "g_hyp", "g_m", "g_LS"]
bad = 0
for rep in range(6):
    outs = []
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            outs.append(run(P[i]))
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        for n, a, b in zip(names, o, ref[i]):
            ne = (a != b).sum().item()
            nanc = (~torch.isfinite(a)).sum().item()
            if ne or nanc:
                bad += 1
                print("rep", rep, "stream", i, n, "mismatch", ne, "nonfinite", nanc, "maxrel", ((a - b).abs().max() / b.abs().max()).item())
print("bad", bad)
