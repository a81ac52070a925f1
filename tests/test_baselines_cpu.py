"""the exact-GP comparison baselines (mobocmf/models/mfgp.py, mfgp_lin.py, MESMOC_MFGP.py) on the host
mirror: kernels vs an independent numpy evaluation of the reference's formulas, exact conditioning vs dense numpy algebra,
the marginal likelihood improves under fit(), the RFF sample interpolates, MES values vs a numpy restatement."""
import math

import numpy as np
import scipy.stats as st
import torch

from mobocmf_amd.acquisition_functions.MESMOC_MFGP import MESMOC_MFGP, _MES_MFGP
from mobocmf_amd.models.mfgp import MFGP
from mobocmf_amd.models.mfgp_lin import MFGP_lin


def _data(n=14, d=2, nf=2, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.random((n, d))
    fid = (np.arange(n) % nf).astype(float)
    y = np.sin(3 * x.sum(1)) + 0.3 * fid * np.cos(2 * x[:, 0])
    X = torch.as_tensor(np.concatenate([x, fid[:, None]], 1))
    return X, torch.as_tensor(y)[:, None], x, fid, y


def _np_rbf(a, b, ls):
    a, b = a / ls, b / ls
    return np.exp(-0.5 * ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1))


def test_mfgp_kernel_conditioning_and_fit():
    X, Y, x, fid, y = _data()
    m = MFGP(X, Y, 2)
    ks, kn = m.covar_module.cov_funct_signal, m.covar_module.cov_funct_noise
    K = m.covar_module(X, X).detach().numpy()
    want = float(ks.outputscale) * _np_rbf(x, x, ks.base_kernel.lengthscale.detach().numpy().ravel()) + \
        np.minimum(fid[:, None], fid[None, :]) * float(kn.outputscale) * _np_rbf(x, x, kn.base_kernel.lengthscale.detach().numpy().ravel())
    assert np.abs(K - want).max() < 1e-12                       # mfgp.py:170-184 (min fidelity counted from 0)
    assert abs(float(m.likelihood.noise) - 0.1) < 1e-7 and abs(float(ks.outputscale) - 1.0) < 1e-6      # set in float32, then .double(): as the reference
    Xt = torch.as_tensor(np.random.default_rng(1).random((5, 2)))
    p = m.predict(Xt, 1)
    Kn = want + float(m.likelihood.noise) * np.eye(len(y))
    xt = Xt.numpy()
    Ks = float(ks.outputscale) * _np_rbf(xt, x, ks.base_kernel.lengthscale.detach().numpy().ravel()) + \
        np.minimum(1.0, fid[None, :]) * float(kn.outputscale) * _np_rbf(xt, x, kn.base_kernel.lengthscale.detach().numpy().ravel())
    assert np.abs(p.mean.detach().numpy() - Ks @ np.linalg.solve(Kn, y)).max() < 1e-10
    Kss = float(ks.outputscale) + float(kn.outputscale)
    assert np.abs(p.variance.detach().numpy() - (Kss - np.einsum("ij,ji->i", Ks, np.linalg.solve(Kn, Ks.T)))).max() < 1e-10
    assert m.training                                             # predict() restores train mode (mfgp.py:59-60)
    sign, logdet = np.linalg.slogdet(Kn)
    mll = -0.5 * y @ np.linalg.solve(Kn, y) - 0.5 * logdet - 0.5 * len(y) * math.log(2 * math.pi)
    assert abs(float(m.marginal_log_likelihood()) - mll) < 1e-9
    m.fit(num_iters=60, lr=0.05)
    assert float(m.marginal_log_likelihood()) > mll
    f = m.sample_from_posterior(1, nFeatures=300, rng=np.random.default_rng(2))
    hi = fid == 1
    with torch.no_grad():
        mu = m.predict(torch.as_tensor(x[hi]), 1).mean.numpy()
    assert np.abs(f(x[hi]) - mu).max() < 1.5                      # a posterior draw stays near the posterior mean at the data
    assert f(x[0], gradient=True).shape == (600, 2) or f(x[0], gradient=True).shape == (2,)


def test_torch_statement_matches_exact_gp_oracle():
    """The package's differentiable torch statement of both baselines vs oracle/exact_gp_oracle.py (numpy, dense inverses):
    kernel matrix, marginal likelihood, predictive moments at every fidelity, off-default hyper-parameters."""
    from oracle import exact_gp_oracle as EO
    for cls, nf, n, d in ((MFGP, 2, 14, 2), (MFGP, 3, 60, 4), (MFGP_lin, 3, 15, 2), (MFGP_lin, 5, 80, 3)):
        X, Y, x, fid, y = _data(n=n, d=d, nf=nf, seed=n)
        m = cls(X, Y, nf)
        with torch.no_grad():
            m.covar_module.cov_funct_signal.outputscale = 0.8
            m.covar_module.cov_funct_noise.outputscale = 0.35
            m.covar_module.cov_funct_noise.base_kernel.lengthscale = 0.7 * m.covar_module.cov_funct_signal.base_kernel.lengthscale.reshape(-1)
            m.likelihood.noise = 0.02
            if cls is MFGP_lin:
                m.covar_module.rho.copy_(torch.linspace(0.6, 1.4, nf - 1))
        hyp, Xn, noise = EO.hyp_of(m), X.numpy(), float(m.likelihood.noise)
        kind = "MFGP" if cls is MFGP else "MFGP_lin"
        K_or = EO.mf_kernel(Xn, Xn, hyp) if cls is MFGP else EO.mf_kernel_lin(Xn, Xn, hyp, nf)
        assert np.abs(m.covar_module(X, X).detach().numpy() - K_or).max() < 1e-12
        mll_or = EO.marginal_log_likelihood(kind, Xn, y, hyp, noise, nf)
        assert abs(float(m.marginal_log_likelihood()) - mll_or) < 1e-9 * max(1.0, abs(mll_or))
        Xt = np.random.default_rng(1).random((9, d))
        for f in range(nf):
            mu, var = EO.predict(kind, Xn, y, hyp, noise, nf, Xt, f)
            with torch.no_grad():
                p = m.predict(torch.as_tensor(Xt), f)
            assert np.abs(p.mean.numpy() - mu).max() < 1e-9 * max(1.0, np.abs(mu).max())
            assert np.abs(p.variance.numpy() - var).max() < 1e-9 * max(1.0, np.abs(var).max())


def test_mfgp_lin_kernel_and_mean_function():
    X, Y, x, fid, y = _data(n=15, nf=3, seed=3)
    m = MFGP_lin(X, Y, 3)
    rho = m.covar_module.rho.detach().numpy()
    cum = np.concatenate([[1.0], np.cumprod(rho)])
    ks, kn = m.covar_module.cov_funct_signal, m.covar_module.cov_funct_noise
    sig = np.outer(cum[fid.astype(int)], cum[fid.astype(int)])
    noise = (np.minimum(fid[:, None], fid[None, :]) + 1 >= 2).astype(float)          # range(3, nf - 1) is empty at nf = 3
    want = sig * float(ks.outputscale) * _np_rbf(x, x, ks.base_kernel.lengthscale.detach().numpy().ravel()) + \
        noise * float(kn.outputscale) * _np_rbf(x, x, kn.base_kernel.lengthscale.detach().numpy().ravel())
    assert np.abs(m.covar_module(X, X).detach().numpy() - want).max() < 1e-12
    mf = m.get_mean_function_high_fidelity()
    pts = np.random.default_rng(4).random((3, 2))
    vals = mf(pts)
    with torch.no_grad():
        assert np.abs(vals - m.predict(torch.as_tensor(pts), 2).mean.numpy()).max() < 1e-12
    g = mf(pts, gradient=True)
    h = 1e-6
    for k in range(2):
        e = np.zeros(2)
        e[k] = h
        fd = (mf(pts + e) - mf(pts - e)) / (2 * h)
        assert np.abs(g[:, k] - fd).max() < 1e-5
    l0 = float(m.marginal_log_likelihood())
    m.fit(num_iters=40, lr=0.05)
    assert float(m.marginal_log_likelihood()) > l0 and any(p.grad is not None for p in [m.covar_module.rho])


def test_mes_acquisition_values_and_search():
    X, Y, x, fid, y = _data(n=16, seed=5)
    obj = MFGP(X, Y, 2)
    con = MFGP(X, torch.as_tensor(np.cos(2 * x.sum(1)))[:, None], 2)
    Xt = torch.as_tensor(np.random.default_rng(6).random((6, 2)))
    best = float(y.min()) - 0.1
    val = _MES_MFGP(1, obj, best, False)(Xt).detach().numpy()
    with torch.no_grad():
        p = obj.predict(Xt, 1)
        mu, var = p.mean.numpy(), p.variance.numpy()
    noise = float(obj.likelihood.noise)
    z = (best - mu) / np.sqrt(var)
    cdf = np.minimum(st.norm.cdf(z), 1 - np.finfo(np.float32).eps)
    ratio = st.norm.pdf(z) / (1 - cdf)
    vt = var * np.maximum(1 + (z - ratio) * ratio, np.finfo(np.float32).eps) + noise
    want = np.maximum(0.5 * np.log(var + noise) - 0.5 * np.log(vt), 0.0)          # MESMOC_MFGP.py:44-65
    assert np.abs(val - want).max() < 1e-10
    pf = _MES_MFGP(1, con, 0.0, True)(Xt).detach().numpy()
    with torch.no_grad():
        pc = con.predict(Xt, 1)
    assert np.abs(pf - (1 - st.norm.cdf((0.0 - pc.mean.numpy()) / np.sqrt(pc.variance.numpy())))).max() < 1e-10
    acq = MESMOC_MFGP({"o": obj}, {"c": con}, 2, 2, {"o": best}, {"c": 0.0},
                      standard_bounds=torch.tensor([[0.0, 0.0], [1.0, 1.0]], dtype=torch.float64))
    for f in range(2):
        acq.add_blackbox(f, "o", cost_evaluation=1.0 if f == 0 else 10.0)
        acq.add_blackbox(f, "c", cost_evaluation=1.0, is_constraint=True)
    tot = acq.coupled_acq(Xt, 0).detach().numpy()
    assert np.abs(tot - _MES_MFGP(0, obj, best, False)(Xt).detach().numpy() * pf).max() < 1e-12   # feasibility at the top fidelity
    torch.manual_seed(0)
    xn, fsel = acq.get_nextpoint_coupled(maxiter=10)
    assert xn.shape == (2,) and fsel in (0, 1) and bool(((xn >= 0) & (xn <= 1)).all())


def test_multistart_search_scores_every_iterate_once_and_returns_the_best():
    """optimize_acqf_multistart (the stand-in for BoTorch's optimize_acqf at JESMOC_MFDGP.py:137-184): on a concave test
    function it finds the maximiser inside the box, the value it returns is the function's value at the candidate it returns, and
    the function is called once for the raw samples and once per iterate (X_0 ... X_maxiter) -- not twice."""
    from mobocmf_amd.acquisition_functions.JESMOC_MFDGP import optimize_acqf_multistart
    c = torch.tensor([0.3, 0.8], dtype=torch.float64)
    calls = []

    def f(X):
        calls.append(X.shape[0])
        return 1.0 - ((X - c) ** 2).sum(-1)

    bounds = torch.tensor([[0.0, 0.0], [1.0, 1.0]], dtype=torch.float64)
    g = torch.Generator().manual_seed(0)
    x, v = optimize_acqf_multistart(f, bounds, num_restarts=4, raw_samples=50, maxiter=60, lr=0.05, generator=g)
    assert x.shape == (1, 2) and float((x[0] - c).abs().max()) < 2e-2
    assert abs(float(v) - float(f(x)[0])) < 1e-15
    assert calls[0] == 50 and calls[1:-1] == [4] * 61
    # a maximiser outside the box: the search stops at the boundary
    c = torch.tensor([1.4, -0.2], dtype=torch.float64)
    x, v = optimize_acqf_multistart(f, bounds, num_restarts=3, raw_samples=30, maxiter=80, lr=0.05, generator=g)
    assert bool(((x >= 0) & (x <= 1)).all()) and float((x[0] - torch.tensor([1.0, 0.0], dtype=torch.float64)).abs().max()) < 2e-2
