"""Random-Fourier-feature function samples from the prior / the variational posterior of an MFDGP layer
(reference: mobocmf/layers/mfdgp_hidden_layer.py:288-514, numpy host code there as well).

Weight-space view: f(x) = theta^T phi(x), phi(x) = sqrt(2 a / F) cos(W x + b) with W ~ N(0, 1) / lengthscale,
b ~ U(0, 2 pi).  Layer >= 1 kernel a1 E1(x) (nu f f' + af Ef(f)) + a2 E2(x) has the feature map
[ sqrt(nu) f phi_x1(x) ; phi_{x1 f}([x, f]) ; phi_x2(x) ]  (3F features; the middle block is the RBF on [x, f] with
outputscale a1*af, sharing W_x1 and b_x1 with the first block, as the reference does).
Posterior weights given q(u) = N(m, S) at the inducing inputs:  A = Phi Phi^T + s2 I,
theta ~ N( A^-1 Phi m ,  s2 A^-1 + A^-1 Phi S Phi^T A^-1 ), drawn by Matheron's rule with M x M algebra.

The weights are drawn on the host in float64 (M x M algebra, not part of the ELBO hot path).  The returned callables take a
numpy array (n, d) or (d,), like the reference's, and recurse through the previous layer's sample; large batches (the
Pareto grid of MOOP, 1000 d^2 rows) are evaluated on the model's GPU by the HIP kernel ``mobocmf_rff_eval`` (one launch per
layer, the F x n feature matrix never materialised), single points and gradients on the host.
"""
import math

import numpy as np
import torch


def _phi(x, W, b, alpha):
    """(F, n) feature matrix."""
    F = W.shape[0]
    return math.sqrt(2.0 * alpha / F) * torch.cos(W @ x.T + b)


def _posterior_weights(Phi, m, Ls, sigma2, gen):
    """One draw of theta ~ N(A^-1 Phi m, s2 A^-1 + A^-1 Phi S Phi^T A^-1), A = Phi Phi^T + s2 I, S = Ls Ls^T -- the
    distribution the reference samples with F x F factorisations (mfdgp_hidden_layer.py:296-307) -- by Matheron's
    rule in the M-dimensional function space (M inducing points << F features):

        theta = theta0 + Phi (Phi^T Phi + s2 I)^-1 (u - Phi^T theta0 - e),   theta0 ~ N(0, I_F), u ~ N(m, S), e ~ N(0, s2 I_M)

    Same mean (push-through identity) and covariance (s2 A^-1 from the prior draw, A^-1 Phi S Phi^T A^-1 from u);
    O(F M^2) instead of O(F^3)."""
    nF, M = Phi.shape
    rn = lambda n: torch.randn(n, dtype=Phi.dtype, generator=gen)
    theta0, z, e = rn(nF), rn(M), rn(M) * math.sqrt(sigma2)
    u = m + Ls @ z
    G = Phi.T @ Phi
    eye = torch.eye(M, dtype=Phi.dtype)
    jit = sigma2
    for i in range(6):
        Lg, info = torch.linalg.cholesky_ex(G + jit * eye)
        if int(info) == 0:
            break
        jit = sigma2 + 1e-10 * 10 ** i
    rhs = u - Phi.T @ theta0 - e
    return theta0 + Phi @ torch.cholesky_solve(rhs[:, None], Lg)[:, 0]


GRID_ROWS_ON_DEVICE = 4096      # batches at least this large are evaluated on the sample's device (the Pareto grid)


class _OnDevice:
    """Tensors of a sample, mirrored lazily on the devices they are evaluated on."""

    def __init__(self, **tensors):
        self._t = {torch.device("cpu"): tensors}

    def on(self, dev):
        dev = torch.device(dev)
        if dev not in self._t:
            self._t[dev] = {k: v.to(dev) for k, v in self._t[torch.device("cpu")].items()}
        return self._t[dev]


def _as_callable(feature_fn, theta, prev, device=None, kernel_args=None):
    """f(x, gradient=False): numpy in, numpy out -- (n,) values, or the (d,) gradient for a single point.
    Single points (the SLSQP refinements) stay on the host; grids of >= GRID_ROWS_ON_DEVICE rows run on ``device`` through
    the HIP kernel (``kernel_args``: kind, the feature tensors and the three scale factors)."""
    th = _OnDevice(theta=theta)

    def evaluate(xt):
        f_prev = prev._torch(xt) if prev is not None else None
        return th.on(xt.device)["theta"] @ feature_fn(xt, f_prev)

    def evaluate_device(xd):
        """xd (n, d) on the GPU -> the sample's values there (layer recursion: the previous sample first)."""
        from .. import functional as F
        kind, P, scales = kernel_args
        p = P.on(xd.device)
        f_prev = prev._device(xd) if prev is not None else None
        return F.rff_eval(kind, xd, f_prev, p["W1"], p["b1"].reshape(-1), p.get("Wf"), p.get("W2"),
                          None if "b2" not in p else p["b2"].reshape(-1), th.on(xd.device)["theta"], *scales)

    def wrapper(x, gradient=False):
        xt = torch.as_tensor(np.asarray(x), dtype=torch.float64)
        if xt.dim() == 1:
            xt = xt[None, :]
        if gradient:
            assert xt.shape[0] == 1, "the gradient is defined for a single point (as in the reference)"
            xt = xt.clone().requires_grad_(True)
            (g,) = torch.autograd.grad(evaluate(xt).sum(), xt)
            return g[0].numpy()
        with torch.no_grad():
            if device is not None and kernel_args is not None and xt.shape[0] >= GRID_ROWS_ON_DEVICE:
                return evaluate_device(xt.to(device).contiguous()).cpu().numpy()
            return evaluate(xt).numpy()

    wrapper._torch = evaluate
    wrapper._device = evaluate_device
    return wrapper


def _hypers(layer):
    cm = layer.covar_module
    g = lambda t: t.detach().cpu().double()
    if layer.num_layer == 0:
        return {"ls": g(cm.base_kernel.lengthscale).reshape(-1), "alpha": float(g(cm.outputscale))}
    k1, kf = cm.kernels[0].kernels[0], cm.kernels[0].kernels[1].kernels[1]
    kl, k2 = cm.kernels[0].kernels[1].kernels[0], cm.kernels[1]
    return {"ls1": g(k1.base_kernel.lengthscale).reshape(-1), "a1": float(g(k1.outputscale)),
            "lsf": g(kf.base_kernel.lengthscale).reshape(-1), "af": float(g(kf.outputscale)),
            "ls2": g(k2.base_kernel.lengthscale).reshape(-1), "a2": float(g(k2.outputscale)),
            "nu": float(g(kl.variance))}


def _draw_features(h, d, F, gen, layer0):
    """Returns (feature function for host torch, kernel_args = (kind, tensors, (s0, s1, s2)) for mobocmf_rff_eval)."""
    rn = lambda *s: torch.randn(*s, dtype=torch.float64, generator=gen)
    ru = lambda *s: 2.0 * math.pi * torch.rand(*s, dtype=torch.float64, generator=gen)
    if layer0:
        P = _OnDevice(W1=rn(F, d) / h["ls"], b1=ru(F, 1))

        def feats0(x, f):
            p = P.on(x.device)
            return _phi(x, p["W1"], p["b1"], h["alpha"])

        return feats0, (0, P, (math.sqrt(2.0 * h["alpha"] / F), 0.0, 0.0))
    W1, Wf, W2 = rn(F, d) / h["ls1"], rn(F) / h["lsf"], rn(F, d) / h["ls2"]
    b1, b2 = ru(F, 1), ru(F, 1)
    P = _OnDevice(W1=W1, W2=W2, b1=b1, b2=b2, Wf=Wf, W1f=torch.cat([W1, Wf[:, None]], 1))

    def feats(x, f):
        p = P.on(x.device)
        xf = torch.cat([x, f[:, None]], 1)
        return torch.cat([_phi(x, p["W1"], p["b1"], h["a1"]) * f * math.sqrt(h["nu"]),
                          _phi(xf, p["W1f"], p["b1"], h["a1"] * h["af"]), _phi(x, p["W2"], p["b2"], h["a2"])], 0)

    scales = (math.sqrt(2.0 * h["a1"] * h["nu"] / F), math.sqrt(2.0 * h["a1"] * h["af"] / F), math.sqrt(2.0 * h["a2"] / F))
    return feats, (1, P, scales)


def sample_from_posterior(layer, input_dim, prev_sample=None, nFeatures=500, sigma2=1e-6, generator=None, device=None):
    """One function sample from the layer's variational posterior (reference :309-337 layer 0, :364-444 layer >= 1)."""
    h = _hypers(layer)
    vs = layer.variational_strategy
    Z = vs.inducing_points.detach().cpu().double()
    vd = vs._variational_distribution
    m = vd.variational_mean.detach().cpu().double()
    Ls = torch.tril(vd.chol_variational_covar.detach().cpu().double())
    feats, kargs = _draw_features(h, input_dim, nFeatures, generator, layer.num_layer == 0)
    if layer.num_layer == 0:
        assert prev_sample is None
        Phi = feats(Z, None)
    else:
        assert prev_sample is not None
        Phi = feats(Z[:, :-1], Z[:, -1])          # the f column of Z~ is the previous layer's variational mean
    theta = _posterior_weights(Phi, m, Ls, sigma2, generator)
    return _as_callable(feats, theta, prev_sample, device, kargs)


def sample_from_prior(layer, input_dim, prev_sample=None, nFeatures=500, generator=None, device=None):
    """One function sample from the synthetic-problem prior (reference :339-362, :446-514: fixed test hyper-parameters
    lengthscale 0.25 d (x10 for x1), outputscales 1 / 1 / 0.01, nu 1)."""
    d = input_dim
    if layer.num_layer == 0:
        h = {"ls": torch.full((d,), 0.25 * d, dtype=torch.float64), "alpha": 1.0}
    else:
        h = {"ls1": torch.full((d,), 2.5 * d, dtype=torch.float64), "a1": 1.0, "lsf": torch.ones(1, dtype=torch.float64),
             "af": 1.0, "ls2": torch.full((d,), 0.25 * d, dtype=torch.float64), "a2": 0.01, "nu": 1.0}
    feats, kargs = _draw_features(h, d, nFeatures, generator, layer.num_layer == 0)
    nF = nFeatures if layer.num_layer == 0 else 3 * nFeatures
    theta = torch.randn(nF, dtype=torch.float64, generator=generator)
    return _as_callable(feats, theta, prev_sample, device, kargs)
