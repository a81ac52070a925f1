"""GPyTorch-free parameter containers with the attribute surface the reference's callers touch
(SURVEY.md section 8(b)): constraints, kernels (``.kernels[i]`` tree, ``.base_kernel.lengthscale``,
``.outputscale``, ``.variance``), Gaussian likelihood (``.noise`` / ``.raw_noise``), the Cholesky
variational distribution and a light diagonal ``MultivariateNormal``.

These classes hold parameters and host logic only.  Gram matrices / Cholesky / moments are evaluated
by the HIP library (mobocmf_amd.functional); the only CPU arithmetic here is the one-time
initialisation of q(u) at construction (mfdgp_hidden_layer.py:127-136), which happens before the
model is moved to the GPU.
"""
import math

import torch
from torch import nn


# ------------------------------------------------------------------------------ constraints
class Positive:
    """softplus transform (gpytorch.constraints.Positive)."""

    def transform(self, raw):
        return torch.nn.functional.softplus(raw)

    def inverse_transform(self, v):
        v = torch.as_tensor(v)
        return v + torch.log(-torch.expm1(-v))


class Interval:
    """lo + (hi - lo) * sigmoid(raw)  (gpytorch.constraints.Interval, used at mfdgp.py:116)."""

    def __init__(self, lower_bound, upper_bound):
        self.lower_bound = float(lower_bound)
        self.upper_bound = float(upper_bound)

    def transform(self, raw):
        return self.lower_bound + (self.upper_bound - self.lower_bound) * torch.sigmoid(raw)

    def inverse_transform(self, v):
        v = torch.as_tensor(v)
        p = (v - self.lower_bound) / (self.upper_bound - self.lower_bound)
        return torch.log(p) - torch.log1p(-p)


class GreaterThan(Interval):
    def __init__(self, lower_bound):
        super().__init__(lower_bound, math.inf)

    def transform(self, raw):
        return torch.nn.functional.softplus(raw) + self.lower_bound

    def inverse_transform(self, v):
        v = torch.as_tensor(v) - self.lower_bound
        return v + torch.log(-torch.expm1(-v))


class _Module(nn.Module):
    def initialize(self, **kwargs):
        """gpytorch.Module.initialize: set (constrained) values by name."""
        for name, val in kwargs.items():
            if not hasattr(self, name):
                raise AttributeError(f"Unknown parameter {name} for {self.__class__.__name__}")
            setattr(self, name, val)
        return self


# ------------------------------------------------------------------------------ kernels
class Kernel(_Module):
    def __add__(self, other):
        return AdditiveKernel(self, other)

    def __mul__(self, other):
        return ProductKernel(self, other)


class RBFKernel(Kernel):
    def __init__(self, ard_num_dims=None, active_dims=None, batch_shape=None, **kw):
        super().__init__()
        n = 1 if ard_num_dims is None else ard_num_dims
        self.ard_num_dims = ard_num_dims
        self.active_dims = None if active_dims is None else tuple(active_dims)
        self.raw_lengthscale = nn.Parameter(torch.zeros(1, n))
        self.raw_lengthscale_constraint = Positive()

    @property
    def lengthscale(self):
        return self.raw_lengthscale_constraint.transform(self.raw_lengthscale)

    @lengthscale.setter
    def lengthscale(self, value):
        v = torch.as_tensor(value, dtype=self.raw_lengthscale.dtype).reshape(-1)
        v = v.expand(self.raw_lengthscale.shape[-1]) if v.numel() == 1 else v
        with torch.no_grad():
            self.raw_lengthscale.copy_(self.raw_lengthscale_constraint.inverse_transform(v).reshape(1, -1))


class LinearKernel(Kernel):
    def __init__(self, active_dims=None, batch_shape=None, **kw):
        super().__init__()
        self.active_dims = None if active_dims is None else tuple(active_dims)
        self.raw_variance = nn.Parameter(torch.zeros(1, 1))
        self.raw_variance_constraint = Positive()

    @property
    def variance(self):
        return self.raw_variance_constraint.transform(self.raw_variance)

    @variance.setter
    def variance(self, value):
        v = torch.as_tensor(value, dtype=self.raw_variance.dtype).reshape(1, 1)
        with torch.no_grad():
            self.raw_variance.copy_(self.raw_variance_constraint.inverse_transform(v))


class ScaleKernel(Kernel):
    def __init__(self, base_kernel, batch_shape=None, **kw):
        super().__init__()
        self.base_kernel = base_kernel
        self.raw_outputscale = nn.Parameter(torch.zeros(()))
        self.raw_outputscale_constraint = Positive()

    @property
    def outputscale(self):
        return self.raw_outputscale_constraint.transform(self.raw_outputscale)

    @outputscale.setter
    def outputscale(self, value):
        v = torch.as_tensor(value, dtype=self.raw_outputscale.dtype).reshape(())
        with torch.no_grad():
            self.raw_outputscale.copy_(self.raw_outputscale_constraint.inverse_transform(v))


class AdditiveKernel(Kernel):
    def __init__(self, *kernels):
        super().__init__()
        self.kernels = nn.ModuleList(kernels)


class ProductKernel(Kernel):
    def __init__(self, *kernels):
        super().__init__()
        self.kernels = nn.ModuleList(kernels)


def _hyper_sources(covar_module, kind):
    """(module, raw parameter name) of every hyper-parameter in the C-ABI order (include/mobocmf_hip.h)."""
    if kind == 0:
        return [(covar_module, "raw_outputscale"), (covar_module.base_kernel, "raw_lengthscale")]
    k_x1 = covar_module.kernels[0].kernels[0]
    k_lin = covar_module.kernels[0].kernels[1].kernels[0]
    k_f = covar_module.kernels[0].kernels[1].kernels[1]
    k_x2 = covar_module.kernels[1]
    return [(k_x1, "raw_outputscale"), (k_f, "raw_outputscale"), (k_lin, "raw_variance"), (k_x2, "raw_outputscale"),
            (k_f.base_kernel, "raw_lengthscale"), (k_x1.base_kernel, "raw_lengthscale"),
            (k_x2.base_kernel, "raw_lengthscale")]


def pack_hypers(covar_module, kind):
    """Constrained hyper-parameters in the C-ABI order (include/mobocmf_hip.h).  Differentiable.
    All of them carry the softplus constraint: on the device one launch transforms and packs every raw tensor
    (mobocmf_softplus_pack; its backward writes every raw gradient in one launch too -- instead of a concatenation, a
    softplus, its backward and one gradient copy per parameter: the per-step glue that dominated the small
    configurations); on the CPU the transform is applied once to the concatenated raw values; any other constraint falls
    back to transforming parameter by parameter."""
    src = _hyper_sources(covar_module, kind)
    params = [getattr(m, n) for m, n in src]
    raws = [p.reshape(-1) for p in params]
    if all(type(getattr(m, n + "_constraint")) is Positive for m, n in src):
        if all(p.is_cuda and p.dtype == torch.float64 and p.is_contiguous() for p in params):
            from . import functional as F      # one launch forward, one backward (instead of cat + softplus and a
            return F.softplus_pack(params)     # softplus backward + one gradient copy per parameter)
        return torch.nn.functional.softplus(torch.cat(raws))
    return torch.cat([getattr(m, n + "_constraint").transform(r) for (m, n), r in zip(src, raws)])


def pack_hypers_many(modules_kinds):
    """``pack_hypers`` for several layers in ONE launch forward and one backward (the layers of a model at the top of a
    training step): returns the list of packed vectors (views of one buffer), or None when the single-launch form does not
    apply (a non-softplus constraint, host tensors, more raw tensors than one launch takes) -- the caller then packs layer
    by layer."""
    srcs = [_hyper_sources(cm, kind) for cm, kind in modules_kinds]
    flat = [(m, n) for src in srcs for m, n in src]
    params = [getattr(m, n) for m, n in flat]
    if len(params) > 16 or not all(type(getattr(m, n + "_constraint")) is Positive for m, n in flat) or \
            not all(p.is_cuda and p.dtype == torch.float64 and p.is_contiguous() for p in params):
        return None
    from . import functional as F
    return F.softplus_pack_segments([[getattr(m, n) for m, n in src] for src in srcs])


def gram_cpu_init(covar_module, kind, X):
    """k(X, X) on the host, ONLY for the one-time initial S of the top layer (mfdgp_hidden_layer.py:131-132)."""
    h = pack_hypers(covar_module, kind).detach().double()
    X = X.double()

    def rbf(a, ls):
        a = a / ls
        return torch.exp(-0.5 * ((a[:, None, :] - a[None, :, :]) ** 2).sum(-1))

    if kind == 0:
        return h[0] * rbf(X, h[1:])
    d = X.shape[1] - 1
    x, f = X[:, :d], X[:, d:]
    a1, af, nu, a2, lsf = h[0], h[1], h[2], h[3], h[4]
    return a1 * rbf(x, h[5:5 + d]) * (nu * (f @ f.T) + af * rbf(f, lsf.reshape(1))) + a2 * rbf(x, h[5 + d:])


# ------------------------------------------------------------------------------ distributions / likelihood
class MultivariateNormal:
    """Mean + marginal variance (+ optional dense covariance): what the callers of the path read."""

    batch_rows = None      # set by MFDGP.forward(rows=...): the distribution covers the first batch_rows rows of the batch

    def __init__(self, mean, variance=None, covariance_matrix=None):
        self.mean = mean
        self._variance = variance
        self._cov = covariance_matrix

    @property
    def loc(self):
        return self.mean

    @property
    def variance(self):
        if self._variance is None:
            return torch.diagonal(self._cov, dim1=-2, dim2=-1)
        return self._variance

    @property
    def stddev(self):
        return self.variance.sqrt()

    @property
    def covariance_matrix(self):
        if self._cov is None:
            return torch.diag_embed(self._variance)
        return self._cov


class GaussianLikelihood(_Module):
    def __init__(self, noise_constraint=None, **kw):
        super().__init__()
        self.raw_noise_constraint = noise_constraint if noise_constraint is not None else GreaterThan(1e-4)
        self.raw_noise = nn.Parameter(torch.zeros(1))

    @property
    def noise(self):
        return self.raw_noise_constraint.transform(self.raw_noise)

    @noise.setter
    def noise(self, value):
        v = torch.as_tensor(value, dtype=self.raw_noise.dtype).reshape(1)
        with torch.no_grad():
            self.raw_noise.copy_(self.raw_noise_constraint.inverse_transform(v))

    def expected_log_prob(self, target, dist):
        """Element-wise E_q[log N(y | f, noise)]; the fused masked sum is functional.elbo_data."""
        mean, var, tau = dist.mean, dist.variance, self.noise
        return -0.5 * (((target - mean) ** 2 + var) / tau + torch.log(tau) + math.log(2 * math.pi))

    def forward(self, dist):
        """Marginal p(y*): adds the noise variance (mfdgp.py:230-233)."""
        return MultivariateNormal(dist.mean, dist.variance + self.noise)


# ------------------------------------------------------------------------------ variational distribution
class CholeskyVariationalDistribution(_Module):
    def __init__(self, num_inducing_points, batch_shape=None, mean_init_std=1e-3, **kw):
        super().__init__()
        self.num_inducing_points = num_inducing_points
        self.mean_init_std = mean_init_std
        self.variational_mean = nn.Parameter(torch.zeros(num_inducing_points))
        self.chol_variational_covar = nn.Parameter(torch.eye(num_inducing_points))

    def initialize_variational_distribution(self, prior_dist):
        with torch.no_grad():
            self.variational_mean.copy_(prior_dist.mean.to(self.variational_mean.dtype))
            # GPyTorch adds randn * mean_init_std (0.0 here) -- still advances the global RNG
            self.variational_mean.add_(torch.randn_like(self.variational_mean), alpha=self.mean_init_std)
            cov = prior_dist.covariance_matrix.double()
            jit = 0.0
            for i in range(4):
                L, info = torch.linalg.cholesky_ex(cov + jit * torch.eye(cov.shape[0], dtype=cov.dtype))
                if int(info) == 0:
                    break
                jit = 1e-8 * (10 ** i)
            else:
                raise RuntimeError("initial variational covariance is not positive definite")
            self.chol_variational_covar.copy_(L.to(self.chol_variational_covar.dtype))

    def forward(self):
        L = torch.tril(self.chol_variational_covar)
        return MultivariateNormal(self.variational_mean, covariance_matrix=L @ L.T)
