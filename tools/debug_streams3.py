import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mobocmf_amd.util import synthetic
from mobocmf_amd import functional as F
dev = torch.device("cuda:0")
cfg = dict(synthetic.CONFIGS["C2"])
sur = bench.build_surrogates(cfg, [0, 1], dev)
streams = [torch.cuda.Stream(device=dev) for _ in sur]
orig_b = F._LayerFn.backward
orig_f = F._LayerFn.forward
def fwd(ctx, *a):
    print("fwd stream", hex(torch.cuda.current_stream().cuda_stream))
    return orig_f(ctx, *a)
def bwd(ctx, *g):
    print("bwd stream", hex(torch.cuda.current_stream().cuda_stream))
    return orig_b(ctx, *g)
F._LayerFn.forward = staticmethod(fwd)
F._LayerFn.backward = staticmethod(bwd)
print("streams", [hex(s.cuda_stream) for s in streams], "default", hex(torch.cuda.current_stream().cuda_stream))
gens = [torch.Generator(device=dev) for _ in sur]
bench.one_step(sur, cfg, gens, streams)
torch.cuda.synchronize()
print(sorted(F._scratch.keys()))
