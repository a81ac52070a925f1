"""GPU parity: one variational layer through the C-ABI vs the CPU oracle (forward + all gradients)."""
import numpy as np
import pytest
import torch

from oracle import mfdgp_oracle as O

pytestmark = pytest.mark.gpu


def _mk(kind, d, M, nbase, xdiv, seed, near=False):
    rng = np.random.default_rng(seed)
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    x = t(rng.random((nbase, d)))
    Zx = x[:M].clone() if near and nbase >= M else t(rng.random((M, d)))
    m = t(0.3 * rng.standard_normal(M))
    L_S = t(0.2 * np.eye(M) + 0.05 * np.tril(rng.standard_normal((M, M))))
    if kind == 0:
        hyp = {"ls": t(0.4 + 0.5 * rng.random(d)), "alpha": t(0.7 + rng.random())}
        f = zf = None
    else:
        hyp = {"ls1": t(0.8 + rng.random(d)), "a1": t(0.6 + rng.random()), "lsf": t(0.7 + rng.random()),
               "af": t(0.5 + rng.random()), "nu": t(0.5 + rng.random()), "ls2": t(0.3 + 0.4 * rng.random(d)),
               "a2": t(0.05 + 0.1 * rng.random())}
        f = t(rng.standard_normal(nbase * xdiv))
        zf = t(0.5 * rng.standard_normal(M))
    return x, f, Zx, zf, hyp, m, L_S


def _pack(kind, hyp):
    if kind == 0:
        return torch.cat([hyp["alpha"].reshape(1), hyp["ls"]])
    return torch.cat([hyp["a1"].reshape(1), hyp["af"].reshape(1), hyp["nu"].reshape(1), hyp["a2"].reshape(1),
                      hyp["lsf"].reshape(1), hyp["ls1"], hyp["ls2"]])


def _oracle(kind, x, f, Zx, zf, hyp, m, L_S, xdiv, branch, w):
    leaves = [x, m, L_S] + [hyp[k] for k in hyp] + ([f, zf] if kind == 1 else [])
    for l in leaves:
        l.requires_grad_(True)
    xr = x.repeat_interleave(xdiv, 0)
    Xt = xr if kind == 0 else torch.cat([xr, f[:, None]], 1)
    Zt = Zx if kind == 0 else torch.cat([Zx, zf[:, None]], 1)
    mean, var, _ = O.layer_moments(hyp, Xt, Zt, m, L_S, training=(branch == 0), shortcut=False)
    kl = O.kl_layer(hyp, Zt, m, L_S)
    loss = (w[0] * mean).sum() + (w[1] * var).sum() + w[2] * kl
    loss.backward()
    return mean.detach(), var.detach(), kl.detach()


CASES = [
    # kind, d, M, nbase, xdiv, branch
    (0, 2, 8, 12, 1, 0),
    (1, 2, 8, 12, 1, 0),
    (1, 2, 8, 12, 3, 0),
    (1, 1, 16, 16, 4, 1),
    (0, 5, 130, 300, 1, 0),
    (1, 5, 130, 100, 4, 0),
    (1, 8, 300, 257, 2, 1),
    (0, 32, 64, 200, 1, 0),
    (1, 32, 64, 90, 2, 0),
    (1, 6, 700, 450, 2, 0),      # Mp = 768: the chain's M x M products on the mid-size kernel between its 512 and 1024 cases
    (0, 6, 600, 900, 1, 1),      # Mp = 640
]


def _close(a, b, rtol, name):
    a, b = a.cpu().double(), b.cpu().double()
    scale = max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max()) / scale
    assert err < rtol, f"{name}: max err / max|ref| = {err:.3e}"


@pytest.mark.parametrize("kind,d,M,nbase,xdiv,branch", CASES)
def test_layer_forward_backward_matches_oracle(kind, d, M, nbase, xdiv, branch):
    from mobocmf_amd import functional as F
    x, f, Zx, zf, hyp, m, L_S = _mk(kind, d, M, nbase, xdiv, seed=kind * 100 + M)
    Np = nbase * xdiv
    rng = np.random.default_rng(7)
    w = [torch.tensor(rng.standard_normal(Np)), torch.tensor(rng.standard_normal(Np)), torch.tensor(0.37)]
    mean_o, var_o, kl_o = _oracle(kind, x, f, Zx, zf, hyp, m, L_S, xdiv, branch, w)

    dev = torch.device("cuda")
    g = lambda t, rg=True: None if t is None else t.detach().to(dev).requires_grad_(rg)
    xg, fg, Zg, zfg, mg, LSg = g(x), g(f), g(Zx, False), g(zf), g(m), g(L_S)
    hg = _pack(kind, {k: v.detach() for k, v in hyp.items()}).to(dev).requires_grad_(True)
    mean, var, kl = F.layer_forward(xg, fg, Zg, zfg, hg, mg, LSg, kind, xdiv=xdiv, branch=branch, want_dx=True)
    _close(mean, mean_o, 1e-9, "mean")
    _close(var, var_o, 1e-8, "var")
    _close(kl, kl_o, 1e-10, "kl")
    loss = (w[0].to(dev) * mean).sum() + (w[1].to(dev) * var).sum() + w[2].to(dev) * kl
    loss.backward()
    _close(mg.grad, m.grad, 1e-7, "g_m")
    _close(LSg.grad, torch.tril(L_S.grad), 1e-7, "g_LS")
    _close(hg.grad, _pack(kind, {k: v.grad for k, v in hyp.items()}), 1e-7, "g_hyp")
    _close(xg.grad, x.grad, 1e-7, "g_x")
    if kind == 1:
        _close(fg.grad, f.grad, 1e-7, "g_f")
        _close(zfg.grad, zf.grad, 1e-7, "g_zf")


def test_clamp_branch_and_min_variance_gradients():
    """Rows whose k_nn - q is clamped (train branch) and rows at the variance floor pass no gradient."""
    from mobocmf_amd import functional as F
    kind, d, M, nbase = 0, 2, 10, 10
    x, f, Zx, zf, hyp, m, L_S = _mk(kind, d, M, nbase, 1, seed=3, near=True)   # X == Z rows: k_nn - q ~ jitter
    L_S = 1e-7 * torch.eye(M, dtype=torch.float64)                               # tiny S: variance near the floor
    w = [torch.ones(nbase, dtype=torch.float64), torch.ones(nbase, dtype=torch.float64), torch.tensor(0.0)]
    mean_o, var_o, kl_o = _oracle(kind, x, f, Zx, zf, hyp, m, L_S, 1, 0, w)
    dev = torch.device("cuda")
    hg = _pack(kind, {k: v.detach() for k, v in hyp.items()}).to(dev).requires_grad_(True)
    mg = m.detach().to(dev).requires_grad_(True)
    LSg = L_S.detach().to(dev).requires_grad_(True)
    mean, var, kl = F.layer_forward(x.detach().to(dev), None, Zx.to(dev), None, hg, mg, LSg, kind)
    _close(mean, mean_o, 1e-7, "mean")
    assert torch.allclose(var.cpu(), var_o, rtol=1e-4, atol=1e-12)
    (mean.sum() + var.sum()).backward()
    _close(mg.grad, m.grad, 1e-6, "g_m")


@pytest.mark.parametrize("kind,d,M,nbase,ls", [(0, 2, 8, 40, 0.15), (1, 2, 8, 40, 0.15), (0, 8, 400, 600, 0.3), (1, 8, 400, 600, 0.3)],
                         ids=["kind0_small", "kind1_small", "kind0_ksliced_dual_launch", "kind1_ksliced_dual_launch"])
def test_clamped_columns_take_the_separate_syrk_path(kind, d, M, nbase, ls):
    """clamp(k_nn - q, 0) active in SOME columns (forced robustly with a negative jitter on a well-conditioned
    K_mm: q > k_nn at the inducing rows): the backward must then use Hc = A diag(c gv) A^T != H -- through the small
    operands' pair of launches and through the k-sliced DUAL launch (H and Hc slabs from one grid, M > 384)."""
    from mobocmf_amd import functional as F
    jit = -2e-3
    x, f, Zx, zf, hyp, m, L_S = _mk(kind, d, M, nbase, 1, seed=21)
    for k in ("ls", "ls1", "ls2"):
        if k in hyp:
            hyp[k] = hyp[k] * 0.0 + ls                        # short lengthscale: K_mm ~ diagonal, PD with jitter < 0
    x = torch.cat([Zx, x[M:]], 0)                             # first M data rows sit on the inducing inputs
    if kind == 1:
        f = torch.cat([zf, f[M:]], 0)
    rng = np.random.default_rng(5)
    w = [torch.tensor(rng.standard_normal(nbase)), torch.tensor(rng.standard_normal(nbase)), torch.tensor(0.2)]
    leaves = [m, L_S] + [hyp[k] for k in hyp] + ([f, zf] if kind == 1 else [])
    for l in leaves:
        l.requires_grad_(True)
    Xt = x if kind == 0 else torch.cat([x, f[:, None]], 1)
    Zt = Zx if kind == 0 else torch.cat([Zx, zf[:, None]], 1)
    mean_o, var_o, _ = O.layer_moments(hyp, Xt, Zt, m, L_S, jitter=jit, training=True, shortcut=False)
    kl_o = O.kl_layer(hyp, Zt, m, L_S, jitter=jit)
    ((w[0] * mean_o).sum() + (w[1] * var_o).sum() + w[2] * kl_o).backward()
    A = torch.linalg.solve_triangular(torch.linalg.cholesky(O.gram(hyp, Zt, Zt) + jit * torch.eye(M)), O.gram(hyp, Zt, Xt), upper=False)
    nclamp = int(((O.gram_diag(hyp, Xt) - (A * A).sum(0)) <= 0).sum())
    assert 0 < nclamp < nbase, nclamp
    dev = torch.device("cuda")
    gdev = lambda t, rg=True: None if t is None else t.detach().to(dev).requires_grad_(rg)
    fg, zfg, mg, LSg = gdev(f), gdev(zf), gdev(m), gdev(L_S)
    hg = _pack(kind, {k: v.detach() for k, v in hyp.items()}).to(dev).requires_grad_(True)
    mean, var, kl = F.layer_forward(x.to(dev), fg, Zx.to(dev), zfg, hg, mg, LSg, kind, jitter=jit)
    _close(mean, mean_o.detach(), 1e-9, "mean")
    _close(var, var_o.detach(), 1e-8, "var")
    ((w[0].to(dev) * mean).sum() + (w[1].to(dev) * var).sum() + w[2].to(dev) * kl).backward()
    _close(mg.grad, m.grad, 1e-7, "g_m")
    _close(LSg.grad, torch.tril(L_S.grad), 1e-7, "g_LS")
    _close(hg.grad, _pack(kind, {k: v.grad for k, v in hyp.items()}), 1e-7, "g_hyp")
    if kind == 1:
        _close(fg.grad, f.grad, 1e-7, "g_f")
        _close(zfg.grad, zf.grad, 1e-7, "g_zf")


def test_not_pd_is_reported():
    from mobocmf_amd import functional as F
    dev = torch.device("cuda")
    x, f, Zx, zf, hyp, m, L_S = _mk(0, 2, 8, 12, 1, seed=1)
    Zx[3] = Zx[2]                                       # duplicate inducing row + negative jitter -> not PD
    hg = _pack(0, hyp).to(dev)
    info = torch.zeros((), dtype=torch.int32, device=dev)
    F.layer_forward(x.to(dev), None, Zx.to(dev), None, hg, m.to(dev), L_S.to(dev), 0, jitter=-1e-3, info_out=info)
    assert F.check_info(info) > 0
    info2 = torch.zeros((), dtype=torch.int32, device=dev)
    F.layer_forward(x.to(dev), None, Zx.to(dev), None, hg, m.to(dev), L_S.to(dev), 0, jitter=1e-6, info_out=info2)
    assert F.check_info(info2) == 0


@pytest.mark.parametrize("kind,d,M,nbase,xdiv", [(1, 3, 40, 50, 3), (0, 2, 8, 12, 1), (1, 5, 130, 30, 25), (1, 8, 512, 200, 25),
                                               (0, 8, 512, 5000, 1), (1, 4, 130, 700, 7)],
                         ids=["small", "tiny_kind0", "S25_ragged", "reference_5000x5000", "kind0_5000", "three_panels"])
def test_predictive_covariance_matches_oracle(kind, d, M, nbase, xdiv):
    """K10 (north star: "MFMA for the K_nm K_mm^-1 K_mn predictive-covariance contraction"): the full eval-branch covariance
    K_nn - A^T A + C^T C vs the oracle's ``full_cov`` -- at the reference's own shape too: (T S)^2 = 5000^2 with its default
    S = 25 (mfdgp.py:22-25, :248) at M = 512, d = 8 (oracle evaluated on the GPU through rocBLAS / rocSOLVER there), through
    the column-panel loop (700 x 7 rows = three panels) and with rows / columns that are no multiple of the tile."""
    from mobocmf_amd import functional as F
    x, f, Zx, zf, hyp, m, L_S = _mk(kind, d, M, nbase, xdiv, seed=11 + M)
    dev = torch.device("cuda")
    odev = dev if nbase * xdiv > 1000 else torch.device("cpu")      # big cases: the torch oracle on rocBLAS
    mv = lambda t: None if t is None else t.to(odev)
    xo, fo, Zxo, zfo, mo, LSo = (mv(t) for t in (x, f, Zx, zf, m, L_S))
    hypo = {k: v.to(odev) for k, v in hyp.items()}
    xr = xo.repeat_interleave(xdiv, 0)
    Xt = xr if kind == 0 else torch.cat([xr, fo[:, None]], 1)
    Zt = Zxo if kind == 0 else torch.cat([Zxo, zfo[:, None]], 1)
    mean_o, var_o, ex = O.layer_moments(hypo, Xt, Zt, mo, LSo, training=False, full_cov=True, shortcut=False)
    g = lambda t: None if t is None else t.to(dev)
    mean, cov = F.predictive_covariance(g(x), g(f), g(Zx), g(zf), _pack(kind, hyp).to(dev), g(m), g(L_S), kind, xdiv=xdiv)
    assert cov.shape == (nbase * xdiv, nbase * xdiv)
    _close(mean, mean_o, 1e-8, "mean")
    _close(cov, ex["cov"], 1e-8, "cov")
    assert torch.equal(cov, cov.T)                                   # mirrored, not recomputed
    # the frozen-chain form (acquisition optimisation against a fixed model) gives the same matrix
    fc = F.freeze_chain(g(Zx), g(zf), _pack(kind, hyp).to(dev), g(m), g(L_S), kind, branch=1)
    _, cov2 = F.predictive_covariance(g(x), g(f), g(Zx), g(zf), _pack(kind, hyp).to(dev), None, None, kind, xdiv=xdiv, chain=fc)
    assert torch.equal(cov, cov2)
