import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mobocmf_amd.util import synthetic
dev = torch.device("cuda:0")
cfg = dict(synthetic.CONFIGS["C3"])
sur = bench.build_surrogates(cfg, [0, 1, 2], dev)
gens = []
for i in range(3):
    g = torch.Generator(device=dev); g.manual_seed(100 + i); gens.append(g)
streams = [torch.cuda.Stream(device=dev) for _ in sur]
torch.cuda.synchronize()
bench.one_step(sur, cfg, gens, streams)
torch.cuda.synchronize()
def fwd(i, st):
    model, elbo, opt, (x, y, fid) = sur[i]
    with torch.cuda.stream(st):
        eps = [None, torch.ones(cfg["N"] * cfg["S"], dtype=torch.float64, device=dev)]
        with torch.no_grad():
            out = model(x, eps=eps)
    return out
print("alone on own stream:")
for i in range(3):
    out = fwd(i, streams[i]); torch.cuda.synchronize()
    print(i, [bool(torch.isfinite(o.mean).all()) and bool(torch.isfinite(o.variance).all()) for o in out],
          sur[i][0].hidden_layer_1._info.item(), sur[i][0].hidden_layer_0._info.item())
print("alone on default stream:")
for i in range(3):
    out = fwd(i, torch.cuda.current_stream()); torch.cuda.synchronize()
    print(i, [bool(torch.isfinite(o.mean).all()) and bool(torch.isfinite(o.variance).all()) for o in out])
print("concurrent:")
outs = [fwd(i, streams[i]) for i in range(3)]
torch.cuda.synchronize()
for i, out in enumerate(outs):
    print(i, [bool(torch.isfinite(o.mean).all()) and bool(torch.isfinite(o.variance).all()) for o in out])
