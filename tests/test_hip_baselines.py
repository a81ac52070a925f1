"""SURVEY 8(f) N4 on the GPU path: the exact-GP comparison baselines (mobocmf/models/mfgp.py:24-141,145-184;
mfgp_lin.py:101-189) evaluated on this package's kernels -- Gram (mobocmf_gram_forward), multi-fidelity combination, the
layer's blocked Cholesky + triangular inverse, the triangular MFMA product with column statistics -- against
oracle/exact_gp_oracle.py (a numpy restatement of the reference's formulas with dense inverses: kernel matrix, marginal
likelihood, predictive moments at every fidelity), and beside it against the package's own differentiable torch statement
(which tests/test_baselines_cpu.py pins to the same oracle on the CPU)."""
import copy
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _data(n, d, nf, seed):
    rng = np.random.default_rng(seed)
    x = rng.random((n, d))
    fid = (np.arange(n) % nf).astype(float)
    y = np.sin(3 * x.sum(1)) + 0.3 * fid * np.cos(2 * x[:, 0]) + 0.05 * rng.standard_normal(n)
    return torch.as_tensor(np.concatenate([x, fid[:, None]], 1)), torch.as_tensor(y)[:, None]


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


@pytest.mark.parametrize("cls_name,n,d,nf", [("MFGP", 14, 2, 2), ("MFGP", 200, 5, 3), ("MFGP", 700, 8, 2), ("MFGP_lin", 14, 2, 2),
                                              ("MFGP_lin", 333, 3, 5), ("MFGP_lin", 520, 1, 3)])
def test_exact_gp_baselines_on_the_hip_kernels_match_torch(cls_name, n, d, nf):
    from mobocmf_amd.models.mfgp import MFGP, MFGP_lin
    from mobocmf_amd.models.mfdgp import TL
    cls = MFGP if cls_name == "MFGP" else MFGP_lin
    X, Y = _data(n, d, nf, seed=n + d)
    ref = cls(X, Y, nf, type_lengthscale=TL.MEDIAN if n < 400 else TL.MEDIAN)
    with torch.no_grad():      # off-default hyper-parameters: every factor of the kernels takes part
        ref.covar_module.cov_funct_signal.outputscale = 0.8
        ref.covar_module.cov_funct_noise.outputscale = 0.35
        ref.covar_module.cov_funct_noise.base_kernel.lengthscale = 0.7 * ref.covar_module.cov_funct_signal.base_kernel.lengthscale.reshape(-1)
        ref.likelihood.noise = 0.02
        if cls is MFGP_lin:
            ref.covar_module.rho.copy_(torch.linspace(0.6, 1.4, nf - 1))
    gpu = copy.deepcopy(ref).to(DEV)
    from oracle import exact_gp_oracle as EO
    hyp, Xn, yn, noise = EO.hyp_of(ref), X.numpy(), Y.numpy()[:, 0], float(ref.likelihood.noise)
    # kernel matrix
    with torch.no_grad():
        K_ref = ref.covar_module(ref.x_train, ref.x_train)
        K_hip = gpu._hip_cov(gpu.x_train, gpu.x_train)
        assert _rel(K_hip, K_ref) < 1e-12
        K_or = EO.mf_kernel(Xn, Xn, hyp) if cls is MFGP else EO.mf_kernel_lin(Xn, Xn, hyp, nf)
        assert _rel(K_hip, torch.as_tensor(K_or)) < 1e-12
        # oracle: marginal likelihood and predictive moments from dense numpy inverses (cond(K + noise I) <= ~1e5 here)
        mll_or = EO.marginal_log_likelihood(cls_name, Xn, yn, hyp, noise, nf)
        assert abs(float(gpu.marginal_log_likelihood()) - mll_or) < 1e-8 * max(1.0, abs(mll_or))
        Xo = np.random.default_rng(1).random((37, d))
        for f in range(nf):
            mu_or, var_or = EO.predict(cls_name, Xn, yn, hyp, noise, nf, Xo, f)
            p_hip = gpu.predict(torch.as_tensor(Xo).to(DEV), f)
            assert _rel(p_hip.mean, torch.as_tensor(mu_or)) < 1e-7, f
            assert float((p_hip.variance.cpu() - torch.as_tensor(var_or)).abs().max()) < 1e-7 * float(np.abs(var_or).max() + 1.0), f
        # marginal likelihood: dispatches to the kernels on the GPU under no_grad
        mll_ref = ref.marginal_log_likelihood(hip=False)
        mll_hip = gpu.marginal_log_likelihood()
        assert abs(float(mll_hip) - float(mll_ref)) < 1e-9 * max(1.0, abs(float(mll_ref)))
        # predictive moments at every fidelity, a number of test points that is no multiple of the tile
        Xt = torch.as_tensor(np.random.default_rng(1).random((37, d)))
        for f in range(nf):
            p_ref = ref.predict(Xt, f, hip=False)
            p_hip = gpu.predict(Xt.to(DEV), f)
            assert _rel(p_hip.mean, p_ref.mean) < 1e-8, f
            assert float((p_hip.variance.cpu() - p_ref.variance).abs().max()) < 1e-8 * float(p_ref.variance.abs().max() + 1.0), f
    # with gradients recorded: the torch statement of the kernel matrix, then factorisation, likelihood and their backward on the
    # library (functional.exact_gp_mll) -- same value
    mll_t = gpu.marginal_log_likelihood()
    assert mll_t.requires_grad and abs(float(mll_t) - float(mll_ref)) < 1e-8 * max(1.0, abs(float(mll_ref)))


def test_exact_gp_not_positive_definite_is_reported():
    from mobocmf_amd import functional as F
    K = torch.eye(40, dtype=torch.float64, device=DEV)
    K[7, 7] = -1.0
    st = F.exact_gp_factor(K, torch.ones(40, dtype=torch.float64, device=DEV))
    assert F.check_info(st.info) == 8
    K[7, 7] = 2.0
    st = F.exact_gp_factor(K, torch.ones(40, dtype=torch.float64, device=DEV))
    assert F.check_info(st.info) == 0
    assert abs(float(st.mll) - (-0.5 * 39.5 - 0.5 * math.log(2.0) - 20 * math.log(2 * math.pi))) < 1e-12


@pytest.mark.parametrize("cls_name,n,d,nf", [("MFGP", 60, 2, 2), ("MFGP", 300, 4, 3), ("MFGP_lin", 150, 3, 3), ("MFGP", 520, 6, 2)])
def test_marginal_likelihood_gradient_and_fit_run_on_the_library(cls_name, n, d, nf):
    """The baselines' fit() (mfgp.py:63-64 / mfgp_lin.py: fit_gpytorch_mll on ExactMarginalLogLikelihood): value AND gradient of
    log p(y | X) through mobocmf_exact_gp_factor + one triangular MFMA product (functional.exact_gp_mll) equal the plain torch
    statement's (torch.linalg Cholesky + solves, autograd) for every raw parameter; a short fit follows the same trajectory."""
    from mobocmf_amd.models.mfgp import MFGP, MFGP_lin
    cls = MFGP if cls_name == "MFGP" else MFGP_lin
    X, Y = _data(n, d, nf, seed=3 * n + d)
    base = cls(X, Y, nf)
    with torch.no_grad():
        base.covar_module.cov_funct_signal.outputscale = 0.9
        base.covar_module.cov_funct_noise.outputscale = 0.3
        base.likelihood.noise = 0.03
    grads = {}
    for hip in (True, False):
        m = copy.deepcopy(base).to(DEV)
        v = m.marginal_log_likelihood(hip=hip)
        assert v.requires_grad
        v.backward()
        grads[hip] = (float(v), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    assert abs(grads[True][0] - grads[False][0]) < 1e-9 * max(1.0, abs(grads[False][0]))
    assert grads[True][1].keys() == grads[False][1].keys() and len(grads[True][1]) >= 4
    for k in grads[False][1]:
        a, b = grads[True][1][k], grads[False][1][k]
        assert float((a - b).abs().max()) <= 1e-7 * max(1e-6, float(b.abs().max())), k
    # fit(): the default on the GPU is the library path; the torch statement is forced by patching the dispatcher
    fitted = {}
    for hip in (True, False):
        m = copy.deepcopy(base).to(DEV)
        if not hip:
            orig = m.marginal_log_likelihood
            m.marginal_log_likelihood = lambda hip=None, _o=orig: _o(hip=False)
        m.fit(num_iters=15, lr=0.05)
        with torch.no_grad():
            fitted[hip] = (float(m._hip_factor().mll), torch.cat([p.detach().reshape(-1) for p in m.parameters()]))
    assert abs(fitted[True][0] - fitted[False][0]) < 1e-6 * max(1.0, abs(fitted[False][0]))
    assert float((fitted[True][1] - fitted[False][1]).abs().max()) < 1e-6
    assert fitted[True][0] > grads[True][0]      # the fit improved the likelihood
