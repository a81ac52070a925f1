"""Edge cases of the C-ABI path: smallest sizes, ragged (non-tile-multiple) sizes, single row, many replicas,
argument validation, and elementwise entry points vs plain torch float64 expressions."""
import numpy as np
import pytest
import torch

from oracle import mfdgp_oracle as O
from tests.test_hip_layer import _close, _mk, _oracle, _pack

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.mark.parametrize("kind,d,M,nbase,xdiv", [(0, 1, 1, 1, 1), (1, 1, 1, 1, 1), (0, 3, 2, 1, 1), (1, 2, 3, 1, 25),
                                                 (1, 7, 129, 1, 5), (0, 4, 127, 385, 1), (1, 2, 65, 129, 2),
                                                 # either side of the small-problem kernel thresholds (K = Mp <= 256,
                                                 # <= 512 workgroups; Cholesky panels with <= 16 / > 16 real rows)
                                                 (1, 3, 100, 2100, 4), (0, 6, 200, 5000, 1), (1, 2, 16, 40, 2),
                                                 (1, 2, 17, 40, 2), (0, 3, 80, 64, 1), (1, 4, 256, 96, 3),
                                                 (1, 4, 257, 96, 3)])
def test_tiny_and_ragged_shapes(kind, d, M, nbase, xdiv):
    from mobocmf_amd import functional as F
    x, f, Zx, zf, hyp, m, L_S = _mk(kind, d, M, nbase, xdiv, seed=M + nbase)
    Np = nbase * xdiv
    w = [torch.ones(Np, dtype=torch.float64), 0.5 * torch.ones(Np, dtype=torch.float64), torch.tensor(1.0)]
    mean_o, var_o, kl_o = _oracle(kind, x, f, Zx, zf, hyp, m, L_S, xdiv, 0, w)
    g = lambda t, rg=True: None if t is None else t.detach().to(DEV).requires_grad_(rg)
    fg, zfg, mg, LSg = g(f), g(zf), g(m), g(L_S)
    hg = _pack(kind, {k: v.detach() for k, v in hyp.items()}).to(DEV).requires_grad_(True)
    mean, var, kl = F.layer_forward(x.detach().to(DEV), fg, Zx.to(DEV), zfg, hg, mg, LSg, kind, xdiv=xdiv)
    # 5e-9: K_mm of these random inducing sets has cond ~ 1e7 .. 1e9; two backward-stable Cholesky factorisations (this
    # one, the oracle's LAPACK one) then differ by cond * eps in a = L^-1 m -- the 4-column and the 1-column panel kernels
    # give 1.1e-9 and 0.8e-9 on the (0, 3, 80, 64, 1) case at equal backward error 2e-16 (profiles/r03_chol_accuracy.txt)
    _close(mean, mean_o, 5e-9, "mean")
    _close(var, var_o, 1e-8, "var")
    _close(kl, kl_o, 1e-9, "kl")
    (mean.sum() + 0.5 * var.sum() + kl).backward()
    _close(mg.grad, m.grad, 1e-7, "g_m")
    _close(hg.grad, _pack(kind, {k: v.grad for k, v in hyp.items()}), 1e-7, "g_hyp")


def test_argument_validation():
    from mobocmf_amd import _lib
    from mobocmf_amd import functional as F
    x, f, Zx, zf, hyp, m, L_S = _mk(1, 2, 8, 12, 1, seed=0)
    t = lambda a: a.to(DEV)
    h = _pack(1, hyp).to(DEV)
    with pytest.raises(_lib.MobocmfError):            # kind 1 without f
        F.layer_forward(t(x), None, t(Zx), t(zf), h, t(m), t(L_S), 1)
    with pytest.raises(_lib.MobocmfError):            # wrong hyper-parameter length
        F.layer_forward(t(x), t(f), t(Zx), t(zf), h[:-1], t(m), t(L_S), 1)
    with pytest.raises(_lib.MobocmfError):            # float32 is refused: the path is float64 end to end
        F.layer_forward(t(x).float(), t(f), t(Zx), t(zf), h, t(m), t(L_S), 1)
    with pytest.raises(_lib.MobocmfError):            # CPU tensors are refused: no CPU fallback
        F.layer_forward(x, f, Zx, zf, _pack(1, hyp), m, L_S, 1)


def test_elementwise_entry_points_match_torch():
    from mobocmf_amd import functional as F
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, dtype=torch.float64, device=DEV, generator=g)
    # propagate (mfdgp_hidden_layer.py:272-274) with 3 replicas per row, incl. gradients
    mean, var, eps = r(50).requires_grad_(True), (r(50).abs() + 0.1).requires_grad_(True), r(150)
    out = F.propagate(mean, var, eps, 3)
    ref = mean.repeat_interleave(3) + var.repeat_interleave(3).sqrt() * eps
    assert torch.allclose(out, ref, rtol=1e-14)
    gm, gv = torch.autograd.grad((out * eps).sum(), [mean, var])
    gm_r, gv_r = torch.autograd.grad((ref * eps).sum(), [mean, var])
    assert torch.allclose(gm, gm_r, rtol=1e-12) and torch.allclose(gv, gv_r, rtol=1e-12)
    # masked expected log-likelihood (variational_elbo_mf.py:31-35), empty mask -> exactly 0
    y, fid = r(50), (torch.arange(50, device=DEV) % 3 == 0).double()
    mu, v, tau = r(150).requires_grad_(True), (r(150).abs() + 0.2).requires_grad_(True), torch.tensor([0.3], dtype=torch.float64, device=DEV, requires_grad=True)
    val = F.elbo_data(mu, v, y, fid, tau, 1.0, div=3)
    yy, mm = y.repeat_interleave(3), fid.repeat_interleave(3) == 1
    ref = (-0.5 * (((yy - mu) ** 2 + v) / tau + torch.log(tau) + np.log(2 * np.pi)))[mm].sum() / 3
    assert torch.allclose(val, ref, rtol=1e-12)
    ga = torch.autograd.grad(val, [mu, v, tau])
    gb = torch.autograd.grad(ref, [mu, v, tau])
    for a, b in zip(ga, gb):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-14)
    assert float(F.elbo_data(mu, v, y, fid, tau, 7.0, div=3)) == 0.0
    # acquisition moments (mfdgp.py:258-260) and the fused Adam update vs torch.optim.Adam
    mt, vt = r(40).requires_grad_(True), (r(40).abs() + 0.1).requires_grad_(True)
    mus, vs = F.acq_moments(mt, vt, 8)
    mus_r = mt.reshape(5, 8).mean(1)
    vs_r = (vt + mt ** 2).reshape(5, 8).mean(1) - mus_r ** 2
    assert torch.allclose(mus, mus_r, rtol=1e-13) and torch.allclose(vs, vs_r, rtol=1e-12)
    ga = torch.autograd.grad((mus * 2 + vs.log()).sum(), [mt, vt])
    gb = torch.autograd.grad((mus_r * 2 + vs_r.log()).sum(), [mt, vt])
    assert all(torch.allclose(a, b, rtol=1e-10) for a, b in zip(ga, gb))
    p = r(1000)
    p_ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-2)
    ea, es = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        grad = r(1000)
        p_ref.grad = grad.clone()
        opt.step()
        F.adam_step(p, grad, ea, es, step, 1e-2)
    assert torch.allclose(p, p_ref.detach(), rtol=1e-12, atol=1e-14)


def test_no_uninitialised_workspace_or_output_reads(monkeypatch):
    """With every workspace and output NaN-filled before use (MOBOCMF_POISON), ragged shapes still give the same finite
    ELBO, gradients and acquisition moments: no kernel reads memory it (or an earlier kernel of the call) did not write."""
    from mobocmf_amd import functional as F
    from mobocmf_amd.mlls import VariationalELBOMF
    from tests.test_hip_model import build_model
    from mobocmf_amd.util import synthetic
    from tests.helpers import to_t
    cfg = dict(d=3, L=3, M=37, N=141, S=3, seed=13)
    prob = synthetic.make_problem(**cfg)
    t = lambda a: to_t(a).to("cuda")
    x, y, fid = t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None]
    eps = [None] + [t(e) for e in prob["eps"][1:]]
    res = []
    for poison in (False, True):
        monkeypatch.setattr(F, "_POISON", poison)
        model = build_model(prob, S_train=3, S_acq=3)
        elbo = VariationalELBOMF(model, cfg["N"], cfg["L"])
        r = elbo(model(x, eps=eps), y.T, fid)
        (-r[0]).backward()
        g = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
        model.eval()
        X = x[:11].clone().requires_grad_(True)
        mus, vs = model.predict_for_acquisition(X, 2)
        (mus.sum() + vs.sum()).backward()
        res.append((r[0].detach(), r[1].detach(), g, mus.detach(), vs.detach(), X.grad.clone()))
    for a, b in zip(*res):
        assert bool(torch.isfinite(b).all()) and torch.equal(a, b)


def _random_cases(n=24, seed=2024):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        kind = int(rng.integers(0, 2))
        M = int(rng.integers(1, 301))
        d = int(rng.integers(3 if M > 60 else 1, 13))        # many inducing points in 1-2 dims = ill-conditioned K_mm
        nbase = int(rng.integers(1, 401))
        xdiv = int(rng.choice([1, 1, 2, 3, 5, 8])) if kind == 1 else 1
        out.append((kind, d, M, nbase, xdiv, int(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("kind,d,M,nbase,xdiv,branch", _random_cases())
def test_randomised_shapes_match_oracle(kind, d, M, nbase, xdiv, branch):
    """Seeded random shapes across every kernel-selection threshold: forward moments, KL and all gradients vs the oracle."""
    from mobocmf_amd import functional as F
    x, f, Zx, zf, hyp, m, L_S = _mk(kind, d, M, nbase, xdiv, seed=1000 + M + nbase)
    Np = nbase * xdiv
    rng = np.random.default_rng(M)
    w = [torch.tensor(rng.standard_normal(Np)), torch.tensor(rng.standard_normal(Np)), torch.tensor(0.7)]
    mean_o, var_o, kl_o = _oracle(kind, x, f, Zx, zf, hyp, m, L_S, xdiv, branch, w)
    g = lambda t, rg=True: None if t is None else t.detach().to(DEV).requires_grad_(rg)
    fg, zfg, mg, LSg = g(f), g(zf), g(m), g(L_S)
    hg = _pack(kind, {k: v.detach() for k, v in hyp.items()}).to(DEV).requires_grad_(True)
    mean, var, kl = F.layer_forward(x.detach().to(DEV), fg, Zx.to(DEV), zfg, hg, mg, LSg, kind, xdiv=xdiv, branch=branch)
    _close(mean, mean_o, 1e-7, "mean")
    _close(var, var_o, 1e-7, "var")
    _close(kl, kl_o, 1e-8, "kl")
    ((mean * w[0].to(DEV)).sum() + (var * w[1].to(DEV)).sum() + w[2].to(DEV) * kl).backward()
    _close(mg.grad, m.grad, 1e-6, "g_m")
    _close(LSg.grad, torch.tril(L_S.grad), 1e-6, "g_LS")
    _close(hg.grad, _pack(kind, {k: v.grad for k, v in hyp.items()}), 1e-6, "g_hyp")
    if kind == 1:
        _close(fg.grad, f.grad, 1e-6, "g_f")
        _close(zfg.grad, zf.grad, 1e-6, "g_zf")


def test_fused_adam_matches_torch_adam():
    """mobocmf_adam_multi (one launch for all parameter tensors, step count on the device) vs torch.optim.Adam: 45 tensors
    (more than one 40-tensor table), one without gradient, several steps."""
    from mobocmf_amd.functional import FusedAdam
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(1,), (), (7, 5), (130,), (64, 64)] * 9
    ps = [torch.randn(s, dtype=torch.float64, device=DEV, generator=g) for s in shapes]
    a = [p.clone().requires_grad_(True) for p in ps]
    b = [p.clone().requires_grad_(True) for p in ps]
    oa, ob = FusedAdam(a, lr=3e-3), torch.optim.Adam(b, lr=3e-3)
    for it in range(6):
        grads = [torch.randn(s, dtype=torch.float64, device=DEV, generator=g) for s in shapes]
        for k, (pa, pb, gr) in enumerate(zip(a, b, grads)):
            pa.grad = None if k == 3 else gr.clone()
            pb.grad = None if k == 3 else gr.clone()
        oa.step()
        ob.step()
    assert int(oa.steps_done) == 6
    for pa, pb in zip(a, b):
        assert float((pa - pb).abs().max()) <= 1e-13 * max(1.0, float(pb.abs().max()))
    assert torch.equal(a[3], ps[3])


def test_shortcut_variance_matches_torch():
    from mobocmf_amd import functional as F
    g = torch.Generator(device="cuda").manual_seed(5)
    for M in (1, 7, 64, 130):
        L = torch.randn(M, M, dtype=torch.float64, device=DEV, generator=g)
        L[0, 0] = 1e-7                                        # a row under the floor
        w = torch.randn(M, dtype=torch.float64, device=DEV, generator=g)
        a = L.clone().requires_grad_(True)
        b = L.clone().requires_grad_(True)
        va = F.shortcut_var(a)
        Lt = torch.tril(b)
        vb = (Lt * Lt).sum(1).clamp_min(F.MIN_VARIANCE)
        assert torch.allclose(va, vb, rtol=1e-14, atol=0)
        (va * w).sum().backward()
        (vb * w).sum().backward()
        assert torch.allclose(a.grad, b.grad, rtol=1e-14, atol=0)
