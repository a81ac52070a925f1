import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mobocmf_amd.util import synthetic
dev = torch.device("cuda:0")
cfg = dict(synthetic.CONFIGS["C3"])
sur = bench.build_surrogates(cfg, [0, 1], dev)
streams = [torch.cuda.Stream(device=dev) for _ in sur]
torch.cuda.synchronize()
eps = [None, torch.randn(cfg["N"] * cfg["S"], dtype=torch.float64, device=dev)]
torch.cuda.synchronize()
def fwd(i):
    model, elbo, opt, (x, y, fid) = sur[i]
    out = model(x, eps=eps)
    return elbo(out, y.T, fid)[0]
def scenario(name, s0_parts):
    with torch.cuda.stream(streams[0]):
        sur[0][2].zero_grad(set_to_none=True)
        l0 = fwd(0)
        if "bwd" in s0_parts: (-l0).backward()
        if "adam" in s0_parts: sur[0][2].step()
    with torch.cuda.stream(streams[1]):
        with torch.no_grad():
            l1 = fwd(1)
    torch.cuda.synchronize()
    print(name, float(l0), float(l1))
for rep in range(2):
    scenario("s0 fwd           || s1 fwd", [])
    scenario("s0 fwd+bwd       || s1 fwd", ["bwd"])
    scenario("s0 fwd+bwd+adam  || s1 fwd", ["bwd", "adam"])
