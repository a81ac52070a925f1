import sys, torch
sys.path.insert(0, '.')
from mobocmf_amd import functional as F
dev = torch.device("cuda")
def timeit(fn, iters=50):
    for _ in range(10): fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3
g = torch.Generator(device=dev); g.manual_seed(0)
d, M = 8, 512
for nbase, S in ((2048, 8), (8192, 8)):
    x = torch.rand(nbase, d, dtype=torch.float64, device=dev, generator=g)
    f = torch.randn(nbase * S, dtype=torch.float64, device=dev, generator=g)
    Z = torch.rand(M, d, dtype=torch.float64, device=dev, generator=g); zf = torch.randn(M, dtype=torch.float64, device=dev, generator=g)
    hyp = torch.tensor([1, 1, 1, 0.01, 1] + [1.4] * (2 * d), dtype=torch.float64, device=dev)
    xr = x.repeat_interleave(S, 0)
    t = timeit(lambda: F.gram(1, Z, zf, xr, f, hyp))
    print("gram(kind 1) %d x %d via mobocmf_gram_forward (xdiv=1 path!): %.1f us" % (M, nbase * S, t))
