"""CPU oracle for the random-Fourier-feature posterior function sampler  (TEST INFRASTRUCTURE ONLY).

numpy/scipy restatement of the reference's sampler, which lives in methods of ``MFDGPHiddenLayer``
(mobocmf/layers/mfdgp_hidden_layer.py) -- a module that imports gpytorch at the top and therefore cannot be imported here
(PARITY UNPINNED by the reference itself: it holds no fixtures for this either).  Followed line by line:

  * ``phi_rbf``                      :288-292   features sqrt(2 alpha / F) cos(W x^T + b) and their gradient
  * ``posterior_weights_reference``  :294-307   ``_chol2inv`` + ``_rff_sample_posterior_weights``: the F x F route
        A = Phi Phi^T + s2 I,  m = A^-1 Phi y,  extraVar = A^-1 Phi S Phi^T A^-1,
        theta = m + chol(s2 A^-1 + extraVar)^T z,  z ~ N(0, I_F)
  * ``layer0_features`` / ``layer1_features``   :319-321, :384-399   the feature matrices of the two kernels
  * ``posterior_moments``            the mean and covariance of that draw in closed form -- what any other sampler of the
        same distribution (the product uses Matheron's rule in the M-dimensional function space) must reproduce.
Never imported by the product package.
"""
import numpy as np
import scipy.linalg as spla


def phi_rbf(x, W, b, alpha, nFeatures, gradient=False):
    """:288-292.  x (n, d), W (F, d), b (F, 1) -> (F, n); gradient: (F, d) for a single point."""
    if gradient:
        return -np.sqrt(2.0 * alpha / nFeatures) * np.sin(W @ x.T + b) * W
    return np.sqrt(2.0 * alpha / nFeatures) * np.cos(W @ x.T + b)


def chol2inv(chol):
    """:294-295 (upper Cholesky factor in, inverse out)."""
    return spla.cho_solve((chol, False), np.eye(chol.shape[0]))


def posterior_weights_reference(y_data, S, Phi, sigma2=1e-6, rng=None):
    """:296-307 with the N(0, 1) draw taken from ``rng`` (the reference uses numpy's global generator)."""
    rng = np.random.default_rng() if rng is None else rng
    randomness = rng.normal(loc=0.0, scale=1.0, size=Phi.shape[0])
    A = Phi @ Phi.T + sigma2 * np.eye(Phi.shape[0])
    chol_A = spla.cholesky(A)
    A_inv = chol2inv(chol_A)
    m = spla.cho_solve((chol_A, False), Phi @ y_data)
    extraVar = (A_inv @ Phi) @ S @ (Phi.T @ A_inv)
    return m + (randomness @ spla.cholesky(sigma2 * A_inv + extraVar, lower=False)).T


def posterior_moments(y_data, S, Phi, sigma2=1e-6):
    """Mean and covariance of ``posterior_weights_reference``'s output (same expressions, no draw)."""
    A = Phi @ Phi.T + sigma2 * np.eye(Phi.shape[0])
    chol_A = spla.cholesky(A)
    A_inv = chol2inv(chol_A)
    mean = spla.cho_solve((chol_A, False), Phi @ y_data)
    cov = sigma2 * A_inv + (A_inv @ Phi) @ S @ (Phi.T @ A_inv)
    return mean, cov


def layer0_features(x, W, b, alpha):
    """:319-321."""
    return phi_rbf(x, W, b, alpha, W.shape[0])


def layer1_features(x, f, W_x1, W_f, W_x2, b_x1, b_x2, alpha_x1, alpha_f, alpha_x2, nu_lin):
    """:384-399.  x (n, d), f (n,): [sqrt(nu) f phi_x1(x); phi_x1f([x, f]); phi_x2(x)]  (3F, n)."""
    F = W_x1.shape[0]
    xf = np.concatenate([x, f[:, None]], axis=1)
    W_x1f = np.concatenate([W_x1, W_f[:, None]], axis=1)
    Phi_x1 = phi_rbf(x, W_x1, b_x1, alpha_x1, F)
    Phi_x1f = phi_rbf(xf, W_x1f, b_x1, alpha_x1 * alpha_f, F)
    Phi_x2 = phi_rbf(x, W_x2, b_x2, alpha_x2, F)
    return np.concatenate([Phi_x1 * f * np.sqrt(nu_lin), Phi_x1f, Phi_x2])


def matheron_moments(m, S, Phi, sigma2=1e-6):
    """Mean and covariance of  theta0 + Phi G^-1 (u - Phi^T theta0 - e),  theta0 ~ N(0, I_F), u ~ N(m, S),
    e ~ N(0, s2 I_M), G = Phi^T Phi + s2 I_M  (the estimator mobocmf_amd/layers/rff.py draws), in closed form."""
    M = Phi.shape[1]
    G = Phi.T @ Phi + sigma2 * np.eye(M)
    B = Phi @ np.linalg.inv(G)                       # F x M
    P = np.eye(Phi.shape[0]) - B @ Phi.T             # coefficient of theta0
    return B @ m, P @ P.T + B @ (S + sigma2 * np.eye(M)) @ B.T
