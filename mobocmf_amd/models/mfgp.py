"""Exact-GP multi-fidelity baselines -- host mirror of mobocmf/models/mfgp.py (``MFGP`` :24-141, ``MFKernel`` :145-184) and
mobocmf/models/mfgp_lin.py (``MFGP_lin`` :23-97, ``MFKernel_lin`` :101-189).  SURVEY row N4: the COMPARISON baselines of the
reference's experiments.  Two statements of the same model live here: the plain float64 torch one (differentiable: ``fit()``
and the acquisition optimisation use it; it also runs on the CPU), and -- when the model's tensors live on the GPU and no
gradient is asked for -- the evaluation path on this package's kernels ("exact-GP baselines reuse K1 / K3 / K4"):
the two ARD-RBF Gram matrices from mobocmf_gram_forward, the multi-fidelity combination in one element-wise launch, the
layer's blocked Cholesky + triangular inverse, and the layer's triangular MFMA product with its column-statistics epilogue for
the predictive moments (``marginal_log_likelihood()`` / ``predict()`` under ``torch.no_grad()``, ``hip=True`` forces it).

Same surface as the reference classes: constructor ``(x_train, y_train, num_fidelities, type_lengthscale)`` with the
fidelity in the LAST column of ``x_train`` (counted from 0), ``predict(x, fidelity)`` -> distribution with ``.mean`` /
``.variance`` of the latent function, ``likelihood.noise``, ``covar_module.cov_funct_signal / cov_funct_noise`` with
``base_kernel.lengthscale`` and ``outputscale``, ``sample_from_posterior(fidelity, nFeatures)`` (RFF, ``MFGP``),
``get_mean_function_high_fidelity()`` (``MFGP_lin``).  The reference inherits exact inference and the marginal likelihood
from GPyTorch / BoTorch (``ExactGP``, ``fit_gpytorch_mll`` in its experiment scripts); here ``marginal_log_likelihood()``
and ``fit()`` (Adam on the raw parameters) state them directly.
"""
import math

import numpy as np
import scipy.linalg as spla
import torch
from torch import nn

from .. import gp
from ..util.util import compute_dist, triu_indices
from .mfdgp import TL


class gp_NotPSD(RuntimeError):
    pass


def _rbf(x1, x2, lengthscale):
    a, b = x1 / lengthscale, x2 / lengthscale
    d2 = (a * a).sum(1, keepdim=True) - 2.0 * a @ b.T + (b * b).sum(1, keepdim=True).T
    return torch.exp(-0.5 * d2.clamp_min(0.0))


def _scaled_rbf(input_dim, init_lengthscale, outputscale, ls_constraint=None, os_constraint=None):
    k = gp.ScaleKernel(gp.RBFKernel(ard_num_dims=input_dim, active_dims=list(range(input_dim))))
    if ls_constraint is not None:
        k.base_kernel.raw_lengthscale_constraint = ls_constraint
    if os_constraint is not None:
        k.raw_outputscale_constraint = os_constraint
    k.base_kernel.initialize(lengthscale=init_lengthscale)
    k.initialize(outputscale=outputscale)
    return k


class MFKernel(gp.Kernel):
    """k([x, t], [x', t']) = k_signal(x, x') + min(t, t') k_noise(x, x')   (mfgp.py:145-184; fidelities from 0)."""

    def __init__(self, input_dim, init_lengthscale, **kwargs):
        super().__init__()
        self.input_dim = input_dim
        d = input_dim - 1
        self.cov_funct_noise = _scaled_rbf(d, init_lengthscale, 0.1, gp.Interval(1e-3, 1000.0), gp.Interval(1e-3, 100.0))
        self.cov_funct_signal = _scaled_rbf(d, init_lengthscale, 1.0, gp.Interval(1e-3, 1000.0), gp.Interval(1e-3, 100.0))

    def hip_factors(self, x):
        """(levels int32, signal factor or None) of rows x and the noise-factor table for functional.mf_kernel_combine."""
        return x[:, self.input_dim - 1].round().to(torch.int32), None

    def hip_noise_table(self, num_levels, device):
        return torch.arange(num_levels, dtype=torch.float64, device=device)          # min(t, t') itself (mfgp.py:183)

    def forward(self, x1, x2, **params):
        d = self.input_dim - 1
        t1, t2 = x1[:, d:d + 1], x2[:, d:d + 1]
        ks = self.cov_funct_signal.outputscale * _rbf(x1[:, :d], x2[:, :d], self.cov_funct_signal.base_kernel.lengthscale)
        kn = self.cov_funct_noise.outputscale * _rbf(x1[:, :d], x2[:, :d], self.cov_funct_noise.base_kernel.lengthscale)
        return ks + torch.minimum(t1, t2.T) * kn

    __call__ = forward


class MFKernel_lin(gp.Kernel):
    """Linear (auto-regressive) multi-fidelity kernel with learned rho (mfgp_lin.py:101-189), as written there:
    signal factor = prod of rho up to each fidelity (outer product), noise factor = [min fid >= 1] + sum over
    k in range(3, num_fidelities - 1) of [min fid + 1 >= k] rho[k - 2]^2."""

    def __init__(self, input_dim, init_lengthscale, num_fidelities, **kwargs):
        super().__init__()
        self.num_fidelities = num_fidelities
        self.input_dim = input_dim
        d = input_dim - 1
        self.cov_funct_noise = _scaled_rbf(d, init_lengthscale, 0.1)
        self.cov_funct_signal = _scaled_rbf(d, init_lengthscale, 1.0)
        self.rho = nn.Parameter(0.5 * torch.ones(num_fidelities - 1))

    def _cum(self):
        return torch.cat([torch.ones(1, dtype=self.rho.dtype, device=self.rho.device), torch.cumprod(self.rho, 0)], 0)

    def hip_factors(self, x):
        lev = x[:, self.input_dim - 1].round().to(torch.int32)
        return lev, self._cum().detach()[lev.long()]

    def hip_noise_table(self, num_levels, device):
        """Noise factor of min fidelity level t (counted from 0): [t + 1 >= 2] + sum_k [t + 1 >= k] rho[k - 2]^2 (:160-176)."""
        t = torch.arange(num_levels, dtype=torch.float64, device=device) + 1.0
        tab = (t >= 2).to(torch.float64)
        for k in range(3, self.num_fidelities - 1):
            tab = tab + (t >= k).to(torch.float64) * self.rho.detach()[k - 2] ** 2
        return tab

    def forward(self, x1, x2, **params):
        d = self.input_dim - 1
        f1, f2 = x1[:, d:d + 1] + 1, x2[:, d:d + 1] + 1                       # counted from 1, as in the reference
        min_fid = torch.minimum(f1, f2.T)
        cum = torch.cat([torch.ones(1, dtype=self.rho.dtype, device=self.rho.device), torch.cumprod(self.rho, 0)], 0)
        c1, c2 = cum[(f1.long() - 1).reshape(-1)], cum[(f2.long() - 1).reshape(-1)]
        factor_signal = torch.outer(c1, c2)
        factor_noise = (min_fid >= 2).to(x1.dtype)
        for k in range(3, self.num_fidelities - 1):
            factor_noise = factor_noise + (min_fid >= k).to(x1.dtype) * self.rho[k - 2] ** 2
        ks = self.cov_funct_signal.outputscale * _rbf(x1[:, :d], x2[:, :d], self.cov_funct_signal.base_kernel.lengthscale)
        kn = self.cov_funct_noise.outputscale * _rbf(x1[:, :d], x2[:, :d], self.cov_funct_noise.base_kernel.lengthscale)
        return factor_signal * ks + factor_noise * kn

    __call__ = forward


class _ExactMFGP(nn.Module):
    """Zero-mean exact GP on (x_train, y_train) with a multi-fidelity kernel and Gaussian noise."""

    def __init__(self, x_train, y_train, num_fidelities, covar_module):
        super().__init__()
        self.input_dim = x_train.shape[1] - 1
        self.num_fidelities = num_fidelities
        self.register_buffer("x_train", x_train.double())
        self.register_buffer("y_train", y_train.double().reshape(-1, 1))
        self.likelihood = gp.GaussianLikelihood()        # GreaterThan(1e-4) noise constraint, gpytorch's default
        self.likelihood.noise = 1e-1
        self.covar_module = covar_module
        self.double()

    def get_init_lengthscale(self, type_lengthscale, inputs=None):
        """mfgp.py:66-69 (the as-written row-indexing median)."""
        dists = compute_dist(inputs)
        return torch.sqrt(torch.median(dists[triu_indices(inputs.shape[0], 1)]))

    def forward(self, x):
        """Prior N(0, k(x, x)) (mfgp.py:45-48)."""
        return gp.MultivariateNormal(torch.zeros(x.shape[0], dtype=x.dtype, device=x.device),
                                     covariance_matrix=self.covar_module(x, x))

    # ------------------------------------------------------------------ evaluation on the package's kernels
    def _hip_wanted(self, hip, *tensors):
        if hip is None:
            hip = self.x_train.is_cuda and not torch.is_grad_enabled()
        if hip and not all(t.is_cuda for t in (self.x_train,) + tensors):
            raise RuntimeError("the HIP evaluation path needs the model and its inputs on the GPU (there is no CPU fallback)")
        return hip

    def _hip_cov(self, x1, x2, diag=0.0):
        """k(x1, x2) (+ diag on the diagonal) through mobocmf_gram_forward x 2 + mobocmf_mf_kernel_combine."""
        from .. import functional as F
        cm, d = self.covar_module, self.input_dim
        hs = torch.cat([cm.cov_funct_signal.outputscale.reshape(1), cm.cov_funct_signal.base_kernel.lengthscale.reshape(-1)]).detach()
        hn = torch.cat([cm.cov_funct_noise.outputscale.reshape(1), cm.cov_funct_noise.base_kernel.lengthscale.reshape(-1)]).detach()
        a, b = x1[:, :d].contiguous(), x2[:, :d].contiguous()
        Ks, Kn = F.gram(0, a, None, b, None, hs), F.gram(0, a, None, b, None, hn)
        l1, s1 = cm.hip_factors(x1)
        l2, s2 = cm.hip_factors(x2)
        return F.mf_kernel_combine(Ks.contiguous(), Kn.contiguous(), s1, s2, l1, l2, cm.hip_noise_table(self.num_fidelities, x1.device), diag)

    def _hip_factor(self):
        from .. import functional as F
        with torch.no_grad():
            K = self._hip_cov(self.x_train, self.x_train, diag=float(self.likelihood.noise))
            st = F.exact_gp_factor(K, self.y_train)
        pivot = F.check_info(st.info)
        F.raise_if_abandoned(pivot, "exact-GP factorisation")
        if pivot != 0:
            raise gp_NotPSD("exact-GP training covariance not positive definite")
        return st

    def predict_hip(self, x, fidelity):
        """``predict`` on the package's kernels: latent posterior mean / variance at fidelity ``fidelity`` (no gradient)."""
        from .. import functional as F
        with torch.no_grad():
            t = fidelity * torch.ones((x.shape[0], 1), dtype=x.dtype, device=x.device)
            xt = torch.cat([x, t], 1)
            st = self._hip_factor()
            Kts = self._hip_cov(self.x_train, xt)
            cm = self.covar_module
            lev, sf = cm.hip_factors(xt)
            tab = cm.hip_noise_table(self.num_fidelities, x.device)
            kss = (1.0 if sf is None else sf * sf) * cm.cov_funct_signal.outputscale.detach() + \
                tab[lev.long()] * cm.cov_funct_noise.outputscale.detach()
            mean, var = F.exact_gp_predict(st, Kts, kss * torch.ones(x.shape[0], dtype=x.dtype, device=x.device))
        return gp.MultivariateNormal(mean, var)

    def _train_factor(self):
        n = self.x_train.shape[0]
        K = self.covar_module(self.x_train, self.x_train)
        K = K + self.likelihood.noise.reshape(()) * torch.eye(n, dtype=K.dtype, device=K.device)
        return torch.linalg.cholesky(K)

    def marginal_log_likelihood(self, hip=None):
        """log p(y | X) of the exact GP (what gpytorch's ExactMarginalLogLikelihood x n evaluates).  ``hip``: None = the
        package's kernels when the model is on the GPU (with a gradient recorded: the differentiable torch statement of the
        kernel matrix, its factorisation, inverse, likelihood AND their backward on the library -- functional.exact_gp_mll),
        else plain torch; True / False force."""
        if hip is None and self.x_train.is_cuda and torch.is_grad_enabled():
            hip = True
        if self._hip_wanted(hip):
            if torch.is_grad_enabled():
                from .. import functional as F
                n = self.x_train.shape[0]
                K = self.covar_module(self.x_train, self.x_train)
                K = K + self.likelihood.noise.reshape(()) * torch.eye(n, dtype=K.dtype, device=K.device)
                return F.exact_gp_mll(K, self.y_train.reshape(-1))
            return self._hip_factor().mll
        L = self._train_factor()
        alpha = torch.cholesky_solve(self.y_train, L)
        n = self.y_train.shape[0]
        return -0.5 * (self.y_train * alpha).sum() - torch.log(torch.diagonal(L)).sum() - 0.5 * n * math.log(2.0 * math.pi)

    def posterior(self, x):
        """Latent posterior at rows x = [inputs, fidelity] (exact conditioning, the ExactGP eval-mode call)."""
        L = self._train_factor()
        Ks = self.covar_module(x, self.x_train)
        alpha = torch.cholesky_solve(self.y_train, L)
        V = torch.linalg.solve_triangular(L, Ks.T, upper=False)
        cov = self.covar_module(x, x) - V.T @ V
        return gp.MultivariateNormal((Ks @ alpha)[:, 0], covariance_matrix=cov)

    def __call__(self, x):
        return self.forward(x) if self.training else self.posterior(x)

    def predict(self, x, fidelity, hip=None):
        """mfgp.py:50-61: posterior of the latent function at fidelity ``fidelity`` (``hip`` as in marginal_log_likelihood;
        the kernel path returns mean and marginal variances)."""
        if len(x.shape) > 2:
            assert x.shape[1] == 1
            x = x[:, 0, :]
        if self._hip_wanted(hip, x) and not x.requires_grad:
            return self.predict_hip(x, fidelity)
        self.eval()
        t = fidelity * torch.ones((x.shape[0], 1), dtype=x.dtype, device=x.device)
        result = self(torch.cat([x, t], 1))
        self.train()
        return result

    def fit(self, num_iters=200, lr=0.05):
        """Maximise the marginal likelihood over the raw parameters with Adam (the reference's experiments call BoTorch's
        fit_gpytorch_mll; any optimiser of the same objective serves a baseline)."""
        opt = torch.optim.Adam(self.parameters(), lr=lr)
        best, state = None, None
        for _ in range(num_iters):
            opt.zero_grad()
            loss = -self.marginal_log_likelihood()
            if not bool(torch.isfinite(loss)):
                break
            if best is None or float(loss) < best:
                best, state = float(loss), {k: v.detach().clone() for k, v in self.state_dict().items()}
            loss.backward()
            opt.step()
        if state is not None:
            final = -self.marginal_log_likelihood()
            if not bool(torch.isfinite(final)) or float(final) > best:
                self.load_state_dict(state)
        return self


class MFGP(_ExactMFGP):

    def __init__(self, x_train, y_train, num_fidelities, type_lengthscale=TL.MEDIAN):
        nn.Module.__init__(self)
        ls = self.get_init_lengthscale(type_lengthscale, x_train[:, :x_train.shape[1] - 1].double())
        _ExactMFGP.__init__(self, x_train, y_train, num_fidelities, MFKernel(x_train.shape[1], ls))

    # ---- RFF posterior function sample (mfgp.py:71-141), numpy on the host like the reference
    def _phi_rbf(self, x, W, b, alpha, nFeatures, gradient=False):
        if gradient:
            return -np.sqrt(2.0 * alpha / nFeatures) * np.sin(W @ x.T + b) * W
        return np.sqrt(2.0 * alpha / nFeatures) * np.cos(W @ x.T + b)

    def _rff_sample_posterior_weights(self, y_data, Phi, rng):
        noise = float(self.likelihood.noise)
        z = rng.normal(size=Phi.shape[0])
        A = Phi @ Phi.T + np.eye(Phi.shape[0]) * noise
        chol_A = spla.cholesky(A)
        A_inv = spla.cho_solve((chol_A, False), np.eye(A.shape[0]))
        m = spla.cho_solve((chol_A, False), Phi @ y_data)
        return m + (z @ spla.cholesky(noise * A_inv, lower=False)).T

    def sample_from_posterior(self, fidelity, nFeatures=500, rng=None):
        rng = np.random.default_rng() if rng is None else rng
        x_data, y_data = self.x_train.cpu().numpy(), self.y_train.cpu().numpy()
        g = lambda t: t.detach().cpu().numpy()
        ls_s = g(self.covar_module.cov_funct_signal.base_kernel.lengthscale).flatten()
        ls_n = g(self.covar_module.cov_funct_noise.base_kernel.lengthscale).flatten()
        a_s = float(self.covar_module.cov_funct_signal.outputscale)
        a_n = float(self.covar_module.cov_funct_noise.outputscale)
        d, nf = self.input_dim, self.num_fidelities
        W_n, b_n = rng.normal(size=(nFeatures, d)) / ls_n, rng.uniform(0.0, 2 * np.pi, size=(nFeatures, 1))
        W_s, b_s = rng.normal(size=(nFeatures, d)) / ls_s, rng.uniform(0.0, 2 * np.pi, size=(nFeatures, 1))
        xs, fid = x_data[:, :d], x_data[:, d]
        Phi_n = self._phi_rbf(xs, W_n, b_n, a_n, nFeatures)
        Phi_s = self._phi_rbf(xs, W_s, b_s, a_s, nFeatures)
        mask = np.ones((nFeatures * (nf - 1), xs.shape[0]))
        for i in range(xs.shape[0]):
            mask[0:int(nFeatures * (nf - fid[i] - 1)), i] = 0
        Phi = np.concatenate((Phi_s, np.tile(Phi_n, (nf - 1, 1)) * mask), 0)
        theta = self._rff_sample_posterior_weights(y_data[:, 0], Phi, rng)

        def wrapper(x, gradient=False):
            if x.ndim == 1:
                x = x[None, :]
            if gradient:
                assert x.shape[0] == 1
            P_n = self._phi_rbf(x, W_n, b_n, a_n, nFeatures, gradient=gradient)
            P_s = self._phi_rbf(x, W_s, b_s, a_s, nFeatures, gradient=gradient)
            m = np.ones(nFeatures * (nf - 1))
            m[0:(nFeatures * (nf - fidelity - 1))] = 0
            return theta @ np.concatenate((P_s, np.tile(P_n, (nf - 1, 1)) * m[:, None]), 0)

        return wrapper


class MFGP_lin(_ExactMFGP):

    def __init__(self, x_train, y_train, num_fidelities, type_lengthscale=TL.MEDIAN):
        nn.Module.__init__(self)
        ls = self.get_init_lengthscale(type_lengthscale, x_train[:, :x_train.shape[1] - 1].double())
        _ExactMFGP.__init__(self, x_train, y_train, num_fidelities, MFKernel_lin(x_train.shape[1], ls, num_fidelities))

    def get_mean_function_high_fidelity(self):
        """mfgp_lin.py:66-97: numpy callable of the highest fidelity's posterior mean (and its gradient row by row)."""
        def mean_function(x, gradient=False):
            if len(x.shape) != 2:
                x = x.reshape((1, len(x)))
            dev, dt = self.x_train.device, self.x_train.dtype
            xt = torch.as_tensor(np.concatenate([x, (self.num_fidelities - 1) * np.ones((x.shape[0], 1))], 1), dtype=dt,
                                 device=dev)
            self.eval()
            try:
                if not gradient:
                    with torch.no_grad():
                        return self(xt).mean.cpu().numpy()
                grads = np.ones((xt.shape[0], xt.shape[1] - 1))
                for i in range(xt.shape[0]):
                    xi = xt[i:i + 1].clone().requires_grad_(True)
                    grads[i] = torch.autograd.grad(self(xi).mean.sum(), xi)[0][0, :-1].cpu().numpy()
                return grads
            finally:
                self.train()
        return mean_function
