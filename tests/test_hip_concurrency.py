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


def test_two_threads_two_streams_two_tunings():
    """SURVEY 8(b) "re-entrant and safe to call from several Python threads on distinct streams": two host threads run the same
    layer forward + backward concurrently, each on its own stream and under its OWN tuning (tile height, pairing, syrk workgroup
    budget, block skipping, mid-size kernel form).  The library has no knobs of its own -- they travel in the descriptor -- so
    each thread must reproduce, bit for bit, what its tuning gives when it runs alone."""
    import threading
    from mobocmf_amd import functional as F
    dev = torch.device("cuda:0")
    M, N, S, d = 512, 4096, 4, 6
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    r = lambda *s: torch.randn(*s, dtype=torch.float64, device=dev, generator=g)
    x = torch.rand(N, d, dtype=torch.float64, device=dev, generator=g)
    hyp = torch.tensor([1, 1, 1, 0.01, 1] + [1.2] * (2 * d), dtype=torch.float64, device=dev)
    LS = 0.1 * torch.eye(M, dtype=torch.float64, device=dev) + 0.01 * torch.tril(r(M, M))
    P = [x, r(N * S), x[:M].clone(), 0.1 * r(M), hyp, 0.1 * r(M), LS, r(N * S)]
    P[7][N * S // 4:] = 0.0          # three quarters of the columns carry no upstream gradient: the block skipping has work
    tunings = [dict(tile_rows=64, pair_mode=2, syrk_workgroups=64, sparse_backward=1, mid_gemm_waves=32),
               dict(tile_rows=128, pair_mode=1, syrk_workgroups=512, sparse_backward=0, mid_gemm_waves=8)]

    def run(tn):
        xx, f, Zx, zf, hy, m, L_S, w = P
        leaves = [t.detach().clone().requires_grad_(True) for t in (f, zf, hy, m, L_S)]
        with F.tuning(**tn):
            mean, var, kl = F.layer_forward(xx, leaves[0], Zx, leaves[1], leaves[2], leaves[3], leaves[4], 1, xdiv=S)
        ((w * mean).sum() + (w * w * var).sum() + 0.3 * kl).backward()      # the backward carries the forward's snapshot
        return [mean.detach(), var.detach(), kl.detach()] + [t.grad for t in leaves]

    ref = [run(tn) for tn in tunings]
    torch.cuda.synchronize()
    # the two tunings really select different launches (different summation orders somewhere), yet agree to rounding
    assert any(not torch.equal(a, b) for a, b in zip(ref[0], ref[1]))
    for a, b in zip(ref[0], ref[1]):
        assert float((a - b).abs().max()) <= 1e-7 * max(float(b.abs().max()), 1e-300)      # ~cond(K_mm) * eps on the gradients
    streams = [torch.cuda.Stream(device=dev) for _ in tunings]
    outs, errs = [None, None], []

    def worker(i):
        try:
            with torch.cuda.stream(streams[i]):
                for _ in range(5):
                    outs[i] = run(tunings[i])
            streams[i].synchronize()
        except Exception as e:      # surfaced in the main thread
            errs.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    torch.cuda.synchronize()
    for i in range(2):
        for a, b in zip(outs[i], ref[i]):
            assert torch.equal(a, b), i
