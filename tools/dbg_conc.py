import sys, torch
sys.path.insert(0, '.')
from mobocmf_amd import functional as F
dev = torch.device("cuda:0")
M, N, S, d, ns = 512, 8192, 8, 8, 3
g = torch.Generator(device=dev); g.manual_seed(0)
r = lambda *s: torch.randn(*s, dtype=torch.float64, device=dev, generator=g)
def mk():
    x = torch.rand(N, d, dtype=torch.float64, device=dev, generator=g)
    hyp = torch.tensor([1, 1, 1, 0.01, 1] + [1.4] * (2 * d), dtype=torch.float64, device=dev)
    LS = 0.1 * torch.eye(M, dtype=torch.float64, device=dev) + 0.01 * torch.tril(r(M, M))
    return [x, r(N * S), x[:M].clone(), 0.1 * r(M), hyp, 0.1 * r(M), LS, r(N * S)]
def run(p):
    x, f, Zx, zf, hyp, m, LS, w = p
    leaves = [t.detach().clone().requires_grad_(True) for t in (f, zf, hyp, m, LS)]
    mean, var, kl = F.layer_forward(x, leaves[0], Zx, leaves[1], leaves[2], leaves[3], leaves[4], 1, xdiv=S)
    ((w * mean).sum() + (w * w * var).sum() + 0.3 * kl).backward()
    return [mean.detach(), var.detach(), kl.detach()] + [t.grad for t in leaves]
names = ["mean", "var", "kl", "g_f", "g_zf", "g_hyp", "g_m", "g_LS"]
P = [mk() for _ in range(ns)]
for knob in sys.argv[1:] or ["default"]:
    if knob == "nomid": F.set_mid_gemm_max(0)
    ref = [run(p) for p in P]
    ref2 = [run(p) for p in P]
    torch.cuda.synchronize()
    for i, (o, rf) in enumerate(zip(ref2, ref)):
        for nme, a, b in zip(names, o, rf):
            if not torch.equal(a, b):
                bad = (a != b).nonzero()
                print(knob, "serial-vs-serial", i, nme, "mismatch count", bad.shape[0], "first", bad[:3].tolist(), a.flatten()[bad[0][0] if a.dim()==1 else 0].item())
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    for rep in range(3):
        outs = []
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs.append(run(P[i]))
        torch.cuda.synchronize()
        for i, (o, rf) in enumerate(zip(outs, ref)):
            for nme, a, b in zip(names, o, rf):
                if not torch.equal(a, b):
                    bad = (a != b).nonzero()
                    print(knob, "rep", rep, "sur", i, nme, "mismatch count", bad.shape[0], "first", bad[:3].tolist(), "max abs diff", float((a - b).abs().max()))
    print(knob, "done")
