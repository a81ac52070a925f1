import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mobocmf_amd.util import synthetic
dev = torch.device("cuda:0")
cfg = dict(synthetic.CONFIGS["C3"])
def run(multi, steps=1):
    sur = bench.build_surrogates(cfg, [0, 1, 2], dev)
    gens = []
    for i in range(3):
        g = torch.Generator(device=dev); g.manual_seed(100 + i); gens.append(g)
    streams = [torch.cuda.Stream(device=dev) for _ in sur] if multi else [torch.cuda.current_stream(dev)] * 3
    torch.cuda.synchronize()
    for k in range(steps):
        bench.one_step(sur, cfg, gens, streams)
        torch.cuda.synchronize()
    return sur
a = run(False); b = run(True)
for i in range(3):
    for (n, p), (_, q) in zip(a[i][0].named_parameters(), b[i][0].named_parameters()):
        dg = (p.grad - q.grad).abs().max().item() / max(p.grad.abs().max().item(), 1e-300)
        bad = (~torch.isfinite(q.grad)).sum().item()
        if dg > 1e-12 or bad or dg != dg:
            print(i, n, tuple(p.shape), 'grad relerr', dg, 'nonfinite', bad)
    for (n, p), (_, q) in zip(a[i][0].named_parameters(), b[i][0].named_parameters()):
        dp = (p - q).abs().max().item()
        if dp > 0 or not torch.isfinite(q).all(): print(i, n, 'param diff', dp, 'nonfinite', (~torch.isfinite(q)).sum().item())
    for key in ('exp_avg', 'exp_avg_sq'):
        for (pa, sa), (pb, sb) in zip(a[i][2].state.items(), b[i][2].state.items()):
            d = (sa[key] - sb[key]).abs().max().item()
            if d > 0 or d != d: print(i, key, tuple(pa.shape), d)
print('done')
