import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mobocmf_amd.util import synthetic
dev = torch.device("cuda:0")
cfg = dict(synthetic.CONFIGS["C3"])
def trial(name, use_gen=True, adam=True, zero_none=True, nsur=2, sync_between=False):
    sur = bench.build_surrogates(cfg, list(range(nsur)), dev)
    streams = [torch.cuda.Stream(device=dev) for _ in sur]
    gens = [torch.Generator(device=dev) for _ in sur]
    fixed = torch.randn(cfg["N"] * cfg["S"], dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    res = []
    for step in range(3):
        ls = []
        for (model, elbo, opt, (x, y, fid)), gen, st in zip(sur, gens, streams):
            with torch.cuda.stream(st):
                opt.zero_grad(set_to_none=zero_none)
                eps = [None, torch.randn(cfg["N"] * cfg["S"], dtype=torch.float64, device=dev, generator=gen) if use_gen else fixed]
                out = model(x, eps=eps)
                r = elbo(out, y.T, fid)
                (-r[0]).backward()
                if adam: opt.step()
                ls.append(r[0].detach())
            if sync_between: torch.cuda.synchronize()
        torch.cuda.synchronize()
        res.append([float(l) for l in ls])
    print(name, res)
trial("3 sur baseline", nsur=3)
trial("3 sur fixed eps", nsur=3, use_gen=False)
trial("3 sur no adam", nsur=3, adam=False)
trial("3 sur sync between", nsur=3, sync_between=True)
