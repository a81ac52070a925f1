"""torch.autograd wrappers over the C-ABI (device pointers in, device pointers out).

PyTorch is plumbing here: it owns device memory, streams and the autograd tape; all arithmetic of
the layer runs in libmobocmf_hip.so.  Tensors must be CUDA(HIP) float64; there is no CPU fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import LayerDesc

JITTER = 1e-6        # gpytorch.settings.variational_cholesky_jitter, float64
MIN_VARIANCE = 1e-10  # gpytorch.settings.min_variance, float64

_scratch = {}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _prep(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.MobocmfError("mobocmf_amd: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != torch.float64:
        raise _lib.MobocmfError("mobocmf_amd: the hot path is float64 end to end (got %s)" % t.dtype)
    return t.contiguous()


def scratch_buffer(nbytes, device):
    """Per (device, stream) scratch, grown on demand; dead after each C call."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.05) + 4096, dtype=torch.uint8, device=device)
        _scratch[key] = buf
    return buf


def hyp_len(kind, d):
    return 1 + d if kind == 0 else 5 + 2 * d


def make_desc(kind, d, M, Np, xdiv=1, branch=0, want_dx=False, jitter=JITTER, min_var=MIN_VARIANCE):
    return LayerDesc(kind=kind, d=d, M=M, xdiv=xdiv, Np=Np, branch=branch, want_dx=int(want_dx), jitter=jitter,
                     min_var=min_var)


def workspace_bytes(desc):
    lib = _lib.load()
    a, b = ctypes.c_size_t(), ctypes.c_size_t()
    _lib.check(lib.mobocmf_layer_workspace_bytes(ctypes.byref(desc), ctypes.byref(a), ctypes.byref(b)),
               "mobocmf_layer_workspace_bytes")
    return a.value, b.value


class _LayerFn(torch.autograd.Function):
    """(mean, var, kl) of one variational layer; see include/mobocmf_hip.h."""

    @staticmethod
    def forward(ctx, x, f, Zx, zf, hyp, m, L_S, kind, xdiv, branch, jitter, min_var, want_dx, info_out):
        lib = _lib.require_device()
        x, f, Zx, zf, hyp, m, L_S = (_prep(t) for t in (x, f, Zx, zf, hyp, m, L_S))
        d, M = Zx.shape[1], Zx.shape[0]
        nbase = x.shape[0]
        Np = nbase * xdiv
        if kind == 1 and (f is None or f.numel() != Np or zf is None or zf.numel() != M):
            raise _lib.MobocmfError("layer kind 1 needs f (N') and zf (M)")
        if hyp.numel() != hyp_len(kind, d) or m.numel() != M or tuple(L_S.shape) != (M, M) or x.shape[1] != d:
            raise _lib.MobocmfError("shape mismatch in layer forward")
        desc = make_desc(kind, d, M, Np, xdiv, branch, want_dx, jitter, min_var)
        sb, cb = workspace_bytes(desc)
        dev = x.device
        saved = torch.empty(sb, dtype=torch.uint8, device=dev)
        scratch = scratch_buffer(cb, dev)
        mean = torch.empty(Np, dtype=torch.float64, device=dev)
        var = torch.empty(Np, dtype=torch.float64, device=dev)
        kl = torch.empty((), dtype=torch.float64, device=dev)
        info = info_out if info_out is not None else torch.zeros((), dtype=torch.int32, device=dev)
        rc = lib.mobocmf_layer_forward(ctypes.byref(desc), _ptr(x), _ptr(f), _ptr(Zx), _ptr(zf), _ptr(hyp), _ptr(m),
                                       _ptr(L_S), _ptr(mean), _ptr(var), _ptr(kl), _ptr(info), _ptr(saved), sb,
                                       _ptr(scratch), scratch.numel(), _stream())
        _lib.check(rc, "mobocmf_layer_forward")
        ctx.desc, ctx.sb = desc, sb
        ctx.saved_ws = saved
        ctx.save_for_backward(*[t for t in (x, f, Zx, zf, hyp, m, L_S) if t is not None])
        ctx.has_f = f is not None
        return mean, var, kl

    @staticmethod
    def backward(ctx, g_mean, g_var, g_kl):
        lib = _lib.require_device()
        ts = list(ctx.saved_tensors)
        if ctx.has_f:
            x, f, Zx, zf, hyp, m, L_S = ts
        else:
            x, Zx, hyp, m, L_S = ts
            f = zf = None
        desc = ctx.desc
        dev = x.device
        M, d = Zx.shape
        g_mean, g_var, g_kl = (_prep(t) for t in (g_mean, g_var, g_kl))
        _, cb = workspace_bytes(desc)
        scratch = scratch_buffer(cb, dev)
        new = lambda *s: torch.empty(*s, dtype=torch.float64, device=dev)
        g_f = new(desc.Np) if ctx.has_f else None
        g_zf = new(M) if ctx.has_f else None
        g_hyp, g_m, g_LS = new(hyp.numel()), new(M), new(M, M)
        g_x = new(x.shape[0], d) if desc.want_dx else None
        rc = lib.mobocmf_layer_backward(ctypes.byref(desc), _ptr(x), _ptr(f), _ptr(Zx), _ptr(zf), _ptr(hyp), _ptr(m),
                                        _ptr(L_S), _ptr(g_mean), _ptr(g_var), _ptr(g_kl), _ptr(g_f), _ptr(g_zf),
                                        _ptr(g_hyp), _ptr(g_m), _ptr(g_LS), _ptr(g_x), _ptr(ctx.saved_ws), ctx.sb,
                                        _ptr(scratch), scratch.numel(), _stream())
        _lib.check(rc, "mobocmf_layer_backward")
        return (g_x, g_f, None, g_zf, g_hyp, g_m, g_LS, None, None, None, None, None, None, None)


def layer_forward(x, f, Zx, zf, hyp, m, L_S, kind, xdiv=1, branch=0, jitter=JITTER, min_var=MIN_VARIANCE,
                  want_dx=False, info_out=None):
    """mean (N'), var (N'), kl ()  --  N' = x.shape[0] * xdiv."""
    return _LayerFn.apply(x, f, Zx, zf, hyp, m, L_S, kind, xdiv, branch, jitter, min_var, want_dx, info_out)


def predictive_covariance(x, f, Zx, zf, hyp, m, L_S, kind, xdiv=1, jitter=JITTER):
    """Full eval-branch predictive covariance (N' x N') -- K10, MFMA contraction.  No autograd."""
    lib = _lib.require_device()
    with torch.no_grad():
        x, f, Zx, zf, hyp, m, L_S = (_prep(t) for t in (x, f, Zx, zf, hyp, m, L_S))
        d, M = Zx.shape[1], Zx.shape[0]
        Np = x.shape[0] * xdiv
        desc = make_desc(kind, d, M, Np, xdiv, 1, False, jitter, MIN_VARIANCE)
        sb, cb = workspace_bytes(desc)
        dev = x.device
        saved = torch.empty(sb, dtype=torch.uint8, device=dev)
        scratch = scratch_buffer(cb, dev)
        mean = torch.empty(Np, dtype=torch.float64, device=dev)
        var = torch.empty(Np, dtype=torch.float64, device=dev)
        kl = torch.empty((), dtype=torch.float64, device=dev)
        info = torch.zeros((), dtype=torch.int32, device=dev)
        _lib.check(lib.mobocmf_layer_forward(ctypes.byref(desc), _ptr(x), _ptr(f), _ptr(Zx), _ptr(zf), _ptr(hyp),
                                             _ptr(m), _ptr(L_S), _ptr(mean), _ptr(var), _ptr(kl), _ptr(info),
                                             _ptr(saved), sb, _ptr(scratch), scratch.numel(), _stream()),
                   "mobocmf_layer_forward")
        cov = torch.empty(Np, Np, dtype=torch.float64, device=dev)
        _lib.check(lib.mobocmf_predictive_covariance(ctypes.byref(desc), _ptr(x), _ptr(f), _ptr(hyp), _ptr(cov), Np,
                                                     _ptr(saved), sb, _ptr(scratch), scratch.numel(), _stream()),
                   "mobocmf_predictive_covariance")
    return mean, cov


class _PropagateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, var, eps, div):
        lib = _lib.require_device()
        mean, var, eps = _prep(mean.reshape(-1)), _prep(var.reshape(-1)), _prep(eps.reshape(-1))
        n = eps.numel()
        if mean.numel() * div != n:
            raise _lib.MobocmfError("propagate: eps must have mean.numel()*div entries")
        out = torch.empty(n, dtype=torch.float64, device=mean.device)
        _lib.check(lib.mobocmf_propagate_forward(_ptr(mean), _ptr(var), _ptr(eps), _ptr(out), n, div, _stream()),
                   "mobocmf_propagate_forward")
        ctx.save_for_backward(var, eps)
        ctx.div = div
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.require_device()
        var, eps = ctx.saved_tensors
        g = _prep(g)
        gm, gv = torch.empty_like(var), torch.empty_like(var)
        _lib.check(lib.mobocmf_propagate_backward(_ptr(var), _ptr(eps), _ptr(g), _ptr(gm), _ptr(gv), eps.numel(),
                                                  ctx.div, _stream()), "mobocmf_propagate_backward")
        return gm, gv, None, None


def propagate(mean, var, eps, div=1):
    """f~[n] = mean[n/div] + sqrt(var[n/div]) * eps[n]   (mfdgp_hidden_layer.py:263-274)."""
    return _PropagateFn.apply(mean, var, eps, div)


class _ElboDataFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, var, y, fid, tau, level, div):
        lib = _lib.require_device()
        mean, var, y, fid, tau = (_prep(t.reshape(-1)) for t in (mean, var, y, fid, tau))
        n = mean.numel()
        if y.numel() * div != n or fid.numel() != y.numel():
            raise _lib.MobocmfError("elbo_data: shape mismatch")
        out = torch.empty((), dtype=torch.float64, device=mean.device)
        scratch = scratch_buffer(8192, mean.device)
        _lib.check(lib.mobocmf_elbo_data_forward(_ptr(mean), _ptr(var), _ptr(y), _ptr(fid), _ptr(tau), float(level), n,
                                                 div, _ptr(out), _ptr(scratch), scratch.numel(), _stream()),
                   "mobocmf_elbo_data_forward")
        ctx.save_for_backward(mean, var, y, fid, tau)
        ctx.level, ctx.div = float(level), div
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.require_device()
        mean, var, y, fid, tau = ctx.saved_tensors
        g = _prep(g)
        gm, gv = torch.empty_like(mean), torch.empty_like(var)
        gt = torch.empty_like(tau)
        scratch = scratch_buffer(8192, mean.device)
        _lib.check(lib.mobocmf_elbo_data_backward(_ptr(mean), _ptr(var), _ptr(y), _ptr(fid), _ptr(tau), ctx.level,
                                                  mean.numel(), ctx.div, _ptr(g), _ptr(gm), _ptr(gv), _ptr(gt),
                                                  _ptr(scratch), scratch.numel(), _stream()),
                   "mobocmf_elbo_data_backward")
        return gm, gv, None, None, gt.reshape(ctx.saved_tensors[4].shape), None, None


def elbo_data(mean, var, y, fid, tau, level, div=1):
    """(1/div) sum_{fid==level} E_q[log N(y | f, tau)]   (variational_elbo_mf.py:31-35)."""
    return _ElboDataFn.apply(mean, var, y, fid, tau, level, div)


class _AcqMomentsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu_t, var_t, S):
        lib = _lib.require_device()
        mu_t, var_t = _prep(mu_t.reshape(-1)), _prep(var_t.reshape(-1))
        T = mu_t.numel() // S
        mus = torch.empty(T, dtype=torch.float64, device=mu_t.device)
        vs = torch.empty_like(mus)
        _lib.check(lib.mobocmf_acq_moments_forward(_ptr(mu_t), _ptr(var_t), _ptr(mus), _ptr(vs), T, S, _stream()),
                   "mobocmf_acq_moments_forward")
        ctx.save_for_backward(mu_t)
        ctx.S = S
        return mus, vs

    @staticmethod
    def backward(ctx, g_mus, g_vars):
        lib = _lib.require_device()
        (mu_t,) = ctx.saved_tensors
        g_mus, g_vars = _prep(g_mus), _prep(g_vars)
        gm, gv = torch.empty_like(mu_t), torch.empty_like(mu_t)
        _lib.check(lib.mobocmf_acq_moments_backward(_ptr(mu_t), _ptr(g_mus), _ptr(g_vars), _ptr(gm), _ptr(gv),
                                                    mu_t.numel() // ctx.S, ctx.S, _stream()),
                   "mobocmf_acq_moments_backward")
        return gm, gv, None


def acq_moments(mu_t, var_t, S):
    """mus, vars over the S samples of each test point (mfdgp.py:258-260)."""
    return _AcqMomentsFn.apply(mu_t, var_t, S)


class _JesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vu, vc):
        lib = _lib.require_device()
        vu, vc = _prep(vu.reshape(-1)), _prep(vc.reshape(-1))
        out = torch.empty_like(vu)
        _lib.check(lib.mobocmf_jes_forward(_ptr(vu), _ptr(vc), _ptr(out), vu.numel(), _stream()), "mobocmf_jes_forward")
        ctx.save_for_backward(vu, vc, out)
        return out

    @staticmethod
    def backward(ctx, g):
        vu, vc, out = ctx.saved_tensors
        act = (out > 0).to(g.dtype) * g * 0.5
        return act / vu, -act / vc


def jes(v_uncond, v_cond):
    """0.5 * clamp(log v_uncond - log v_cond, min=0)   (JESMOC_MFDGP.py:52)."""
    return _JesFn.apply(v_uncond, v_cond)


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, mask=None):
    """In-place fused Adam on flat float64 buffers (torch.optim.Adam semantics)."""
    lib = _lib.require_device()
    _lib.check(lib.mobocmf_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(mask),
                                     param.numel(), lr, beta1, beta2, eps, int(step), _stream()), "mobocmf_adam_step")


def check_info(info):
    """Synchronising: raises if the last Cholesky reported a non-positive pivot."""
    lib = _lib.require_device()
    piv = ctypes.c_int32()
    rc = lib.mobocmf_check_info(_ptr(info), ctypes.byref(piv), _stream())
    if rc == _lib.NOT_PD:
        return piv.value
    _lib.check(rc, "mobocmf_check_info")
    return 0


def gemm_f64(A, B, C=None, tri=0, trans_b=False, alpha=1.0, accumulate=False):
    """C (+)= alpha * A @ (B.T if trans_b else B) on the f64 MFMA kernel (tile-aligned shapes only)."""
    lib = _lib.require_device()
    A, B = _prep(A), _prep(B)
    Mr, Kd = A.shape
    Nc = B.shape[0] if trans_b else B.shape[1]
    if C is None:
        C = torch.empty(Mr, Nc, dtype=torch.float64, device=A.device)
    _lib.check(lib.mobocmf_gemm_f64(tri, int(trans_b), Mr, Nc, Kd, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(C),
                                    C.stride(0), alpha, int(accumulate), _stream()), "mobocmf_gemm_f64")
    return C
