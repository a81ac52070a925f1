"""TEST INFRASTRUCTURE -- numpy restatement of the reference's exact-GP comparison baselines.

Only tests/ may import this module; nothing under mobocmf_amd/ does.  Parity unpinned by the reference (no tests or
fixtures there, gpytorch absent): the formulas below follow the reference's source line by line and exact-GP conditioning is
textbook algebra (Rasmussen & Williams eq. 2.23-2.24, 2.30), evaluated with dense numpy inverses -- nothing shared with the
package's torch statement or its HIP kernels.

Conventions (mfgp.py:26): the last column of X is the fidelity, counted from 0.
"""
import math

import numpy as np


def ard_rbf(a, b, ls):
    """RBFKernel with ARD lengthscales on the non-fidelity columns (mfgp.py:156-162): exp(-1/2 |(a - b) / ls|^2)."""
    a, b = np.asarray(a, dtype=np.float64) / ls, np.asarray(b, dtype=np.float64) / ls
    return np.exp(-0.5 * ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1))


def mf_kernel(X1, X2, hyp):
    """MFKernel.forward (mfgp.py:170-184): k_signal + (min(l, l') + 1 - 1) k_noise with fidelities counted from 0.

    hyp = {"alpha_signal", "ls_signal", "alpha_noise", "ls_noise"}."""
    x1, l1 = X1[:, :-1], X1[:, -1]
    x2, l2 = X2[:, :-1], X2[:, -1]
    min_fid = np.minimum(l1[:, None] + 1.0, l2[None, :] + 1.0)
    return hyp["alpha_signal"] * ard_rbf(x1, x2, hyp["ls_signal"]) + \
        (min_fid - 1.0) * hyp["alpha_noise"] * ard_rbf(x1, x2, hyp["ls_noise"])


def mf_kernel_lin(X1, X2, hyp, num_fidelities):
    """MFKernel_lin.forward (mfgp_lin.py:127-189): the signal term scaled by the products of rho up to either fidelity
    (:152-155), the noise term by 1[min >= 2] + sum_{k=3}^{nf-2} rho[k-2]^2 1[min >= k] (fidelities from 1; the loop's upper
    limit ``range(3, num_fidelities - 1)`` is the reference's, :184-185).  hyp additionally holds "rho" (nf - 1 values)."""
    x1, f1 = X1[:, :-1], X1[:, -1].astype(np.int64) + 1
    x2, f2 = X2[:, :-1], X2[:, -1].astype(np.int64) + 1
    rho = np.asarray(hyp["rho"], dtype=np.float64)
    cum = np.concatenate([[1.0], np.cumprod(rho)])
    factor_signal = np.outer(cum[f1 - 1], cum[f2 - 1])
    min_fid = np.minimum(f1[:, None], f2[None, :])
    factor_noise = (min_fid >= 2).astype(np.float64)
    for k in range(3, num_fidelities - 1):
        factor_noise += (min_fid >= k) * rho[k - 2] ** 2
    return factor_signal * hyp["alpha_signal"] * ard_rbf(x1, x2, hyp["ls_signal"]) + \
        factor_noise * hyp["alpha_noise"] * ard_rbf(x1, x2, hyp["ls_noise"])


def _kernel(kind, X1, X2, hyp, num_fidelities):
    return mf_kernel(X1, X2, hyp) if kind == "MFGP" else mf_kernel_lin(X1, X2, hyp, num_fidelities)


def marginal_log_likelihood(kind, X, y, hyp, noise, num_fidelities):
    """ExactGP + GaussianLikelihood with a zero mean (mfgp.py:33-41): log N(y | 0, K + noise I) -- the SUM over the data
    (GPyTorch's ExactMarginalLogLikelihood divides by n; the package reports the sum and so does this)."""
    Kn = _kernel(kind, X, X, hyp, num_fidelities) + noise * np.eye(X.shape[0])
    sign, logdet = np.linalg.slogdet(Kn)
    assert sign > 0
    return -0.5 * y @ np.linalg.solve(Kn, y) - 0.5 * logdet - 0.5 * len(y) * math.log(2.0 * math.pi)


def predict(kind, X, y, hyp, noise, num_fidelities, Xt, fidelity):
    """MFGP.predict (mfgp.py:49-61): the fidelity column is appended to the test inputs, the latent posterior
    (no observation noise added) is returned: mean = K_*n (K + noise I)^-1 y, var = k_** - diag(K_*n (K + noise I)^-1 K_n*)."""
    Xs = np.concatenate([np.asarray(Xt, dtype=np.float64), np.full((len(Xt), 1), float(fidelity))], 1)
    Kn = _kernel(kind, X, X, hyp, num_fidelities) + noise * np.eye(X.shape[0])
    Ks = _kernel(kind, Xs, X, hyp, num_fidelities)
    kss = np.diag(_kernel(kind, Xs, Xs, hyp, num_fidelities))
    return Ks @ np.linalg.solve(Kn, y), kss - np.einsum("ij,ji->i", Ks, np.linalg.solve(Kn, Ks.T))


def hyp_of(model):
    """The constrained hyper-parameter values of a package-side MFGP / MFGP_lin as plain numpy (read through the attribute
    tree the reference exposes: covar_module.cov_funct_{signal,noise}.{outputscale, base_kernel.lengthscale}, .rho)."""
    cm = model.covar_module
    n = lambda t: t.detach().cpu().double().numpy()
    hyp = {"alpha_signal": float(cm.cov_funct_signal.outputscale), "ls_signal": n(cm.cov_funct_signal.base_kernel.lengthscale).ravel(),
           "alpha_noise": float(cm.cov_funct_noise.outputscale), "ls_noise": n(cm.cov_funct_noise.base_kernel.lengthscale).ravel()}
    if hasattr(cm, "rho"):
        hyp["rho"] = n(cm.rho).ravel()
    return hyp
