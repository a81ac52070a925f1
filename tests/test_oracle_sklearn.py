"""The oracle against THIRD-PARTY code: scikit-learn's Gaussian-process kernels and exact GP regressor (sklearn is in the
image; nothing here was written for this repository).  The reference's arithmetic lives in GPyTorch, which cannot be imported
(parity unpinned by the reference); this pins the oracle's kernel functions and its variational predictive -- in the limit
where it must equal an exact GP -- to an independent implementation of the same textbook objects.

* ``oracle.gram`` kind 0 (alpha * ARD-RBF, mfdgp_hidden_layer.py:43-47) == ConstantKernel * RBF(length_scale=ls);
* ``oracle.gram`` kind 1 (a1 RBF_x1 (nu <f, f'> + af RBF_f) + a2 RBF_x2, mfdgp_hidden_layer.py:68-88,115) assembled from
  sklearn's RBF / DotProduct evaluations on the active columns;
* ``oracle.layer_moments`` with the optimal q(u) at Z = X == GaussianProcessRegressor(optimizer=None).predict
  mean and variance, up to the 1e-6 variational jitter the reference adds to K_mm.
"""
import numpy as np
import pytest
import torch

from oracle import mfdgp_oracle as O

sk = pytest.importorskip("sklearn.gaussian_process")
from sklearn.gaussian_process import GaussianProcessRegressor  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, DotProduct  # noqa: E402

T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)


@pytest.mark.parametrize("d,n1,n2,seed", [(1, 7, 5, 0), (3, 12, 9, 1), (8, 20, 20, 2)])
def test_layer0_kernel_is_sklearns_scaled_ard_rbf(d, n1, n2, seed):
    rng = np.random.default_rng(seed)
    X1, X2 = rng.random((n1, d)), rng.random((n2, d))
    ls, alpha = 0.2 + rng.random(d), 0.5 + rng.random()
    K = O.gram({"ls": T(ls), "alpha": T(alpha)}, T(X1), T(X2)).numpy()
    K_sk = (ConstantKernel(alpha) * RBF(length_scale=ls))(X1, X2)
    assert np.abs(K - K_sk).max() < 1e-13
    Kx = O.gram({"ls": T(ls), "alpha": T(alpha)}, T(X1), T(X2), expanded=True).numpy()      # GPyTorch's own evaluation order
    assert np.abs(Kx - K_sk).max() < 1e-12
    assert np.abs(O.gram_diag({"ls": T(ls), "alpha": T(alpha)}, T(X1)).numpy() - np.diag((ConstantKernel(alpha) * RBF(ls))(X1))).max() < 1e-13


@pytest.mark.parametrize("d,n1,n2,seed", [(1, 6, 4, 3), (2, 10, 11, 4), (5, 15, 8, 5)])
def test_multifidelity_kernel_is_the_composition_of_sklearn_kernels(d, n1, n2, seed):
    rng = np.random.default_rng(seed)
    X1, X2 = rng.standard_normal((n1, d + 1)), rng.standard_normal((n2, d + 1))
    h = {"ls1": 0.5 + rng.random(d), "ls2": 0.2 + rng.random(d), "lsf": 0.7 + rng.random(), "a1": 0.3 + rng.random(),
         "af": 0.3 + rng.random(), "nu": 0.3 + rng.random(), "a2": 0.05 + rng.random()}
    hyp = {k: T(v) for k, v in h.items()}
    K = O.gram(hyp, T(X1), T(X2)).numpy()
    x1, f1, x2, f2 = X1[:, :d], X1[:, d:], X2[:, :d], X2[:, d:]
    K_sk = h["a1"] * RBF(h["ls1"])(x1, x2) * (h["nu"] * DotProduct(sigma_0=0.0)(f1, f2) + h["af"] * RBF(h["lsf"])(f1, f2)) \
        + h["a2"] * RBF(h["ls2"])(x1, x2)
    assert np.abs(K - K_sk).max() < 1e-12 * max(1.0, np.abs(K_sk).max())
    diag = np.array([(h["a1"] * (h["nu"] * f * f + h["af"]) + h["a2"]) for f in f1[:, 0]])
    assert np.abs(O.gram_diag(hyp, T(X1)).numpy() - diag).max() < 1e-13
    assert np.abs(np.diag(O.gram(hyp, T(X1), T(X1)).numpy()) - diag).max() < 1e-12


@pytest.mark.parametrize("d,n,seed", [(1, 9, 0), (2, 14, 1), (4, 25, 2)])
def test_optimal_variational_posterior_predicts_like_sklearns_exact_gp(d, n, seed):
    """Z = X and q(u) = the exact posterior of f(X): the variational layer's predictive (eval branch) is the exact GP's.
    sklearn: GaussianProcessRegressor(kernel, alpha = noise, optimizer = None).  The reference's K_mm carries a 1e-6 jitter
    (gpytorch.settings.variational_cholesky_jitter), so the regressor is given the same jittered prior: kernel + WhiteKernel
    is avoided by folding the jitter into the comparison tolerance (1e-4 absolute on O(1) quantities)."""
    rng = np.random.default_rng(seed)
    X, Xs = rng.random((n, d)), rng.random((11, d))
    y = np.sin(3.0 * X.sum(1)) + 0.1 * rng.standard_normal(n)
    ls, alpha, tau = 0.3 + 0.5 * rng.random(d), 0.8 + rng.random(), 0.05
    hyp = {"ls": T(ls), "alpha": T(alpha)}
    Kt = O.gram(hyp, T(X), T(X)) + O.JITTER * torch.eye(n, dtype=torch.float64)
    Ky = Kt + tau * torch.eye(n, dtype=torch.float64)
    m = Kt @ torch.linalg.solve(Ky, T(y))                       # E[u | y]
    S = Kt - Kt @ torch.linalg.solve(Ky, Kt)                    # cov[u | y]
    L_S = torch.linalg.cholesky(0.5 * (S + S.T) + 1e-12 * torch.eye(n, dtype=torch.float64))
    mean, var, _ = O.layer_moments(hyp, T(Xs), T(X), m, L_S, training=False, shortcut=False)
    gpr = GaussianProcessRegressor(kernel=ConstantKernel(alpha, "fixed") * RBF(ls, "fixed"), alpha=tau, optimizer=None)
    gpr.fit(X, y)
    mu_sk, sd_sk = gpr.predict(Xs, return_std=True)
    assert np.abs(mean.numpy() - mu_sk).max() < 1e-4
    assert np.abs(var.numpy() - sd_sk ** 2).max() < 1e-4
    # and the marginal likelihood the ELBO is bounded by: collapsed bound at Z = X == sklearn's log marginal likelihood
    hyp_state = {"Zx": T(X), "layers": [{"hyp": hyp, "m": m, "L_S": L_S}], "noise": [T(tau)]}
    elbo, _ = O.elbo(hyp_state, T(X), T(y), torch.zeros(n, dtype=torch.float64), shortcut=False)
    lml = gpr.log_marginal_likelihood_value_
    assert abs(float(elbo) - lml) < 2e-3 * max(1.0, abs(lml)), (float(elbo), lml)


@pytest.mark.parametrize("M,d,seed", [(5, 1, 0), (12, 3, 1), (20, 2, 2)])
def test_kl_term_is_torch_distributions_kl(M, d, seed):
    """KL(q(u) || p(u)) of the oracle (kl_mvn_mvn restated) == torch.distributions.kl_divergence of the two
    multivariate normals (PyTorch's own implementation)."""
    rng = np.random.default_rng(seed)
    Z = T(rng.random((M, d)))
    hyp = {"ls": T(0.3 + rng.random(d)), "alpha": T(0.5 + rng.random())}
    m = T(rng.standard_normal(M))
    L_S = torch.tril(T(0.3 * rng.standard_normal((M, M)))) + torch.diag(T(0.5 + rng.random(M)))
    K = O.gram(hyp, Z, Z) + O.JITTER * torch.eye(M, dtype=torch.float64)
    ref = torch.distributions.kl_divergence(torch.distributions.MultivariateNormal(m, scale_tril=L_S),
                                            torch.distributions.MultivariateNormal(torch.zeros(M, dtype=torch.float64), K))
    assert abs(float(O.kl_layer(hyp, Z, m, L_S)) - float(ref)) < 1e-9 * max(1.0, abs(float(ref)))


def test_expected_log_prob_is_the_gauss_hermite_integral():
    """GaussianLikelihood.expected_log_prob restated: E_{f ~ N(mu, var)} log N(y | f, tau), against numpy's
    Gauss-Hermite quadrature of torch.distributions.Normal.log_prob (exact for this quadratic integrand)."""
    rng = np.random.default_rng(0)
    y, mu = T(rng.standard_normal(9)), T(rng.standard_normal(9))
    var, tau = T(0.05 + rng.random(9)), T(0.3)
    t, w = np.polynomial.hermite.hermgauss(20)
    f = mu[:, None] + torch.sqrt(2.0 * var)[:, None] * T(t)[None, :]
    quad = (torch.distributions.Normal(f, torch.sqrt(tau)).log_prob(y[:, None]) * T(w)[None, :]).sum(1) / np.sqrt(np.pi)
    assert float((O.expected_log_prob(y, mu, var, tau) - quad).abs().max()) < 1e-12
