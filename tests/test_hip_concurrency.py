"""Layers of independent surrogates run concurrently on separate HIP streams (bench.py, the fitter): results must be
bit-identical to serial execution.  Regression test for an inter-workgroup race in the Cholesky panel kernel
(a late-scheduled workgroup re-read a diagonal block that block 0 had already overwritten in place)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_three_streams_bitwise_equal_to_serial():
    from mobocmf_amd import functional as F
    dev = torch.device("cuda:0")
    M, N, S, d, ns = 512, 8192, 8, 8, 3
    g = torch.Generator(device=dev)
    g.manual_seed(0)
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

    P = [mk() for _ in range(ns)]
    ref = [run(p) for p in P]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    for rep in range(4):
        outs = []
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs.append(run(P[i]))
        torch.cuda.synchronize()
        for o, rf in zip(outs, ref):
            for a, b in zip(o, rf):
                assert torch.equal(a, b)
