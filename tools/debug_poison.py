import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mobocmf_amd.util import synthetic
from mobocmf_amd import functional as F
dev = torch.device("cuda:0")
cfg = dict(synthetic.CONFIGS["C3"])
def poison():
    # recycle ~6 GB of NaN-filled blocks of assorted sizes through the caching allocator
    F._scratch.clear()
    ts = [torch.full((n,), float("nan"), dtype=torch.float64, device=dev) for n in
          [100 * 2**20, 100 * 2**20, 100 * 2**20, 110 * 2**20, 64 * 2**20, 2**20, 2**16, 2**12, 513, 65536, 8192] * 2]
    del ts
sur = bench.build_surrogates(cfg, [1], dev)
g = torch.Generator(device=dev); g.manual_seed(101)
st = [torch.cuda.current_stream(dev)]
for k in range(3):
    poison()
    l = bench.one_step(sur, cfg, [g], st)
    torch.cuda.synchronize()
    print(k, float(l[0]), [n for n, p in sur[0][0].named_parameters() if not torch.isfinite(p.grad).all()])
