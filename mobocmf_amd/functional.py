"""torch.autograd wrappers over the C-ABI (device pointers in, device pointers out).

PyTorch is plumbing here: it owns device memory, streams and the autograd tape; all arithmetic of
the layer runs in libmobocmf_hip.so.  Tensors must be CUDA(HIP) float64; there is no CPU fallback.
"""
import contextlib
import ctypes
import os
import threading

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
    if t.device.index != torch.cuda.current_device():
        # work is enqueued on the CURRENT device's stream: one process per GPU (torch.cuda.set_device first)
        raise _lib.MobocmfError("mobocmf_amd: tensor on %s but the current device is cuda:%d -- call "
                                "torch.cuda.set_device (one process per GPU)" % (t.device, torch.cuda.current_device()))
    return t.contiguous()


_POISON = bool(int(os.environ.get("MOBOCMF_POISON", "0")))   # debugging aid: NaN-fill every workspace before use


def _poison(buf):
    if _POISON and not torch.cuda.is_current_stream_capturing():
        buf[:buf.numel() // 8 * 8].view(torch.float64).fill_(float("nan"))
    return buf


def _empty(*shape, dtype=torch.float64, device=None):
    """Output allocation; NaN-filled under MOBOCMF_POISON so that a kernel leaving part of an output unwritten shows."""
    t = torch.empty(*shape, dtype=dtype, device=device)
    if _POISON and dtype == torch.float64 and not torch.cuda.is_current_stream_capturing():
        t.fill_(float("nan"))
    return t


def _empty_like(t):
    return _empty(t.shape, dtype=t.dtype, device=t.device)


def scratch_buffer(nbytes, device):
    """Per (device, stream) scratch, grown on demand; dead after each C call.  A buffer handed out while the stream is
    being captured is baked into the graph: it is also recorded in ``_capture_pins`` so that the object owning the graph
    can keep it alive (``take_capture_pins``) -- growing the arena later must not free memory a live graph replays on."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.05) + 4096, dtype=torch.uint8, device=device)
        _scratch[key] = buf
    if torch.cuda.is_current_stream_capturing() and not any(b is buf for b in _capture_pins):
        _capture_pins.append(buf)
    return _poison(buf)


_capture_pins = []


def take_capture_pins():
    """The scratch buffers handed out under capture since the last call (the caller -- a graphed step -- holds them for as
    long as its graph lives)."""
    out = list(_capture_pins)
    _capture_pins.clear()
    return out


def release_scratch(stream=None):
    """Drop the arena's entry of ``stream`` (all entries without an argument): the memory returns to the allocator once
    no graph pins it.  Fitters call this when a training phase's streams retire (torch hands stream handles out of a
    small pool, so a stale entry would otherwise pin up to ~1 GB per handle for ever)."""
    if stream is None:
        _scratch.clear()
        _side_streams.clear()
        return
    for key in [k for k in _scratch if k[1] == stream.cuda_stream]:
        del _scratch[key]
    for key in [k for k in _side_streams if k[1] == stream.cuda_stream]:
        del _side_streams[key]


def hyp_len(kind, d):
    return 1 + d if kind == 0 else 5 + 2 * d


PHASE_ALL, PHASE_CHAIN, PHASE_PANEL, PHASE_CHAIN_ONLY, PHASE_PANEL_INPUTS = 0, 1, 2, 3, 4


# ---------------------------------------------------------------------------------------------------------------------
# Kernel-selection knobs (include/mobocmf_hip.h: mobocmf_tuning).  The LIBRARY keeps no state: the values travel with every
# call.  This module keeps the host-side default record (`set_*` below change it: size sweeps, A/B timing, parity tests) and a
# per-thread override (`tuning(...)` context manager).  A descriptor takes a SNAPSHOT when it is built in the forward; the
# backward -- which autograd runs on its own thread -- reuses the forward's snapshot, so a call is sized and launched under one
# set of values whatever happens to the defaults in between.
# ---------------------------------------------------------------------------------------------------------------------
_TUNING_DEFAULT = None
_tuning_tls = threading.local()


def _default_tuning():
    global _TUNING_DEFAULT
    if _TUNING_DEFAULT is None:
        t = _lib.Tuning()
        _lib.check(_lib.load().mobocmf_tuning_init(ctypes.byref(t)), "mobocmf_tuning_init")
        _TUNING_DEFAULT = t
    return _TUNING_DEFAULT


def current_tuning():
    """A snapshot (copy) of the tuning calls issued by this thread use now."""
    t = getattr(_tuning_tls, "override", None)
    return (t if t is not None else _default_tuning()).copy()


def _set_default(**kw):
    t = _default_tuning()
    for k, v in kw.items():
        assert k in _lib.Tuning.KNOBS, k
        setattr(t, k, int(v))


@contextlib.contextmanager
def tuning(**kw):
    """Per-THREAD override of the knobs for the calls issued inside the block (forward calls; a backward uses the snapshot
    its forward took): ``with F.tuning(tile_rows=64, sparse_backward=0): ...``."""
    prev = getattr(_tuning_tls, "override", None)
    t = current_tuning()
    for k, v in kw.items():
        if k not in _lib.Tuning.KNOBS:
            raise TypeError("unknown tuning knob %r" % k)
        setattr(t, k, int(v))
    _tuning_tls.override = t
    try:
        yield t
    finally:
        _tuning_tls.override = prev


_probe_tls = threading.local()


@contextlib.contextmanager
def probe_events(table, layer=None):
    """Diagnostic (bench.py per_kernel_instep_ms): the PANEL calls issued by this thread inside the block -- those of layer
    index ``layer`` of a batched-chain forward, or every one if None -- carry ``table`` (a ctypes array of PROBE_EVENTS
    hipEvent_t handles) in their descriptor; the backward of such a call records into the same table."""
    prev = getattr(_probe_tls, "cur", None)
    _probe_tls.cur = (table, layer)
    try:
        yield
    finally:
        _probe_tls.cur = prev


def _probe_for(layer):
    cur = getattr(_probe_tls, "cur", None)
    if cur is None or (cur[1] is not None and layer is not None and cur[1] != layer):
        return None
    return cur[0]


def make_desc(kind, d, M, Np, xdiv=1, branch=0, want_dx=False, jitter=JITTER, min_var=MIN_VARIANCE, phase=PHASE_ALL,
              tune=None, probe=None):
    """``tune``: a Tuning snapshot to carry (default: the calling thread's current one); ``probe``: a ctypes array of
    PROBE_EVENTS hipEvent_t handles (mobocmf_layer_desc.probe_events) or None."""
    if not 1 <= d <= _lib.MAX_D:
        raise _lib.MobocmfError("mobocmf_amd: a layer takes 1..%d input dimensions (got %d): the Gram kernels keep one "
                                "input row in registers" % (_lib.MAX_D, d))
    if not 1 <= xdiv <= _lib.MAX_XDIV:
        raise _lib.MobocmfError("mobocmf_amd: at most %d samples per input row (num_samples_for_acquisition / "
                                "num_samples_for_training), got %d" % (_lib.MAX_XDIV, xdiv))
    desc = LayerDesc(kind=kind, d=d, M=M, xdiv=xdiv, Np=Np, branch=branch, want_dx=int(want_dx), jitter=jitter,
                     min_var=min_var, phase=phase, reserved=0)
    snap = tune if tune is not None else current_tuning()
    desc.tuning = ctypes.pointer(snap)
    desc._tune = snap              # keeps the snapshot alive as long as the descriptor
    if probe is not None:
        desc.probe_events = ctypes.cast(probe, ctypes.c_void_p)
        desc._probe = probe
    return desc


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
        saved = _poison(torch.empty(sb, dtype=torch.uint8, device=dev))
        scratch = scratch_buffer(cb, dev)
        mean = _empty(Np, device=dev)
        var = _empty(Np, device=dev)
        kl = _empty((), device=dev)
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
        new = lambda *s: _empty(*s, device=dev)
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


# ------------------------------------------------------------------------------------------------------------
# The same layer call in two halves (include/mobocmf_hip.h, MOBOCMF_PHASE_*): the CHAIN half depends on the
# parameters alone, so a model launches it for every layer up front on a side stream and the latency-bound
# M x M chains (one wavefront per Cholesky panel) run under the grid-filling PANEL work of the other layers.
# Autograd runs each half's backward on the stream its forward ran on and orders them through the ``token``.
# ------------------------------------------------------------------------------------------------------------
_side_streams = {}


def side_stream_for(main):
    """The chain stream paired with ``main`` (one per (device, main stream))."""
    key = (main.device.index, main.cuda_stream)
    st = _side_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=main.device)
        _side_streams[key] = st
    return st


class LayerPass:
    """State shared by the two halves of one split layer call."""
    __slots__ = ("kind", "d", "M", "Np", "xdiv", "branch", "jitter", "min_var", "want_dx", "info", "saved", "sb", "cb",
                 "bscratch", "main", "side", "ready", "kl", "frozen", "tune", "probe")

    def desc(self, phase):
        # one tuning snapshot (taken when the pass was created, in the forward) serves both halves, forward and backward
        return make_desc(self.kind, self.d, self.M, self.Np, self.xdiv, self.branch, self.want_dx, self.jitter,
                         self.min_var, phase, tune=self.tune, probe=self.probe)


class _ChainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Zx, zf, hyp, m, L_S, P):
        lib = _lib.require_device()
        Zx, zf, hyp, m, L_S = (_prep(t) for t in (Zx, zf, hyp, m, L_S))
        dev = Zx.device
        desc = P.desc(PHASE_CHAIN)
        P.sb, P.cb = workspace_bytes(desc)
        P.saved = _poison(torch.empty(P.sb, dtype=torch.uint8, device=dev))
        scratch = scratch_buffer(P.cb, dev)
        kl = _empty((), device=dev)
        token = torch.empty(1, dtype=torch.float64, device=dev)      # never read: it only orders the autograd nodes
        rc = lib.mobocmf_layer_forward(ctypes.byref(desc), None, None, _ptr(Zx), _ptr(zf), _ptr(hyp), _ptr(m), _ptr(L_S),
                                       None, None, _ptr(kl), _ptr(P.info), _ptr(P.saved), P.sb, _ptr(scratch),
                                       scratch.numel(), _stream())
        _lib.check(rc, "mobocmf_layer_forward[chain]")
        if P.main is not None:      # allocated here (side stream), consumed on the main stream
            for t in (P.saved, kl, token):
                t.record_stream(P.main)
            for t in (hyp, zf, m, L_S):     # allocated on the main stream, read here
                if t is not None:
                    t.record_stream(P.side)
        ctx.P = P
        ctx.save_for_backward(*[t for t in (Zx, zf, hyp, m, L_S) if t is not None])
        ctx.has_f = zf is not None
        return token, kl

    @staticmethod
    def backward(ctx, g_token, g_kl):
        lib = _lib.require_device()
        P = ctx.P
        ts = list(ctx.saved_tensors)
        if ctx.has_f:
            Zx, zf, hyp, m, L_S = ts
        else:
            Zx, hyp, m, L_S = ts
            zf = None
        dev = Zx.device
        M = P.M
        phase = PHASE_CHAIN
        scratch = P.bscratch
        P.bscratch = None
        if scratch is None:         # the PANEL half took no part in this backward (only the KL was differentiated)
            scratch = _poison(torch.empty(P.cb, dtype=torch.uint8, device=dev))
            phase = PHASE_CHAIN_ONLY
        if g_kl is None:
            g_kl = torch.zeros((), dtype=torch.float64, device=dev)
        g_kl = _prep(g_kl)
        desc = P.desc(phase)
        new = lambda *s: _empty(*s, device=dev)
        g_zf = new(M) if ctx.has_f else None
        g_hyp, g_m, g_LS = new(hyp.numel()), new(M), new(M, M)
        rc = lib.mobocmf_layer_backward(ctypes.byref(desc), None, None, _ptr(Zx), _ptr(zf), _ptr(hyp), _ptr(m),
                                        _ptr(L_S), None, None, _ptr(g_kl), None, _ptr(g_zf), _ptr(g_hyp), _ptr(g_m),
                                        _ptr(g_LS), None, _ptr(P.saved), P.sb, _ptr(scratch), scratch.numel(), _stream())
        _lib.check(rc, "mobocmf_layer_backward[chain]")
        return None, g_zf, g_hyp, g_m, g_LS, None


class _PanelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, Zx, zf, hyp, token, P):
        lib = _lib.require_device()
        x, f, Zx, zf, hyp = (_prep(t) for t in (x, f, Zx, zf, hyp))
        if x.shape[0] * P.xdiv != P.Np or x.shape[1] != P.d or (P.kind == 1 and (f is None or f.numel() != P.Np)):
            raise _lib.MobocmfError("shape mismatch between the two halves of a split layer call")
        dev = x.device
        desc = P.desc(PHASE_PANEL)
        scratch = scratch_buffer(P.cb, dev)
        mean = _empty(P.Np, device=dev)
        var = _empty(P.Np, device=dev)
        rc = lib.mobocmf_layer_forward(ctypes.byref(desc), _ptr(x), _ptr(f), _ptr(Zx), _ptr(zf), _ptr(hyp), None, None,
                                       _ptr(mean), _ptr(var), None, None, _ptr(P.saved), P.sb, _ptr(scratch),
                                       scratch.numel(), _stream())
        _lib.check(rc, "mobocmf_layer_forward[panel]")
        ctx.P = P
        ctx.save_for_backward(*[t for t in (x, f, Zx, zf, hyp) if t is not None])
        ctx.has_f = f is not None
        return mean, var

    @staticmethod
    def backward(ctx, g_mean, g_var):
        lib = _lib.require_device()
        P = ctx.P
        ts = list(ctx.saved_tensors)
        if ctx.has_f:
            x, f, Zx, zf, hyp = ts
        else:
            x, Zx, hyp = ts
            f = zf = None
        dev = x.device
        g_mean, g_var = _prep(g_mean), _prep(g_var)
        if P.frozen:        # constant parameters: nothing is handed to a CHAIN half
            desc = P.desc(PHASE_PANEL_INPUTS)
            scratch = scratch_buffer(P.cb, dev)
        else:
            desc = P.desc(PHASE_PANEL)
            # private scratch: H, Hc, da stay in it until the CHAIN half (side stream) has consumed them
            scratch = _poison(torch.empty(P.cb, dtype=torch.uint8, device=dev))
            if P.side is not None:
                scratch.record_stream(P.side)
        new = lambda *s: _empty(*s, device=dev)
        g_f = new(P.Np) if ctx.has_f else None
        g_zf = new(P.M) if ctx.has_f else None
        g_hyp = new(hyp.numel())
        g_x = new(x.shape[0], P.d) if P.want_dx else None
        rc = lib.mobocmf_layer_backward(ctypes.byref(desc), _ptr(x), _ptr(f), _ptr(Zx), _ptr(zf), _ptr(hyp), None, None,
                                        _ptr(g_mean), _ptr(g_var), None, _ptr(g_f), _ptr(g_zf), _ptr(g_hyp), None, None,
                                        _ptr(g_x), _ptr(P.saved), P.sb, _ptr(scratch), scratch.numel(), _stream())
        _lib.check(rc, "mobocmf_layer_backward[panel]")
        if P.frozen:
            return g_x, g_f, None, None, None, None, None
        P.bscratch = scratch
        # the token gradient is a real tensor here: the CHAIN half may run on another stream, and the autograd engine
        # orders the two streams through the tensors that flow between the nodes
        return g_x, g_f, None, g_zf, g_hyp, torch.zeros(1, dtype=torch.float64, device=dev), None


def layer_chain(Zx, zf, hyp, m, L_S, kind, Np, xdiv=1, branch=0, jitter=JITTER, min_var=MIN_VARIANCE, want_dx=False,
                info_out=None, main=None, side=None):
    """CHAIN half of a layer call for N' = Np rows: returns (LayerPass, token); ``LayerPass.kl`` holds the KL.  Call it
    under ``torch.cuda.stream(side)`` with ``main`` = the stream the PANEL half will run on (or both None: same stream)."""
    P = LayerPass()
    P.kind, P.d, P.M, P.Np, P.xdiv, P.branch = kind, Zx.shape[1], Zx.shape[0], Np, xdiv, branch
    P.jitter, P.min_var, P.want_dx = jitter, min_var, want_dx
    P.info = info_out if info_out is not None else torch.zeros((), dtype=torch.int32, device=Zx.device)
    P.bscratch, P.main, P.side, P.frozen = None, main, side, False
    P.tune, P.probe = current_tuning(), _probe_for(None)
    token, P.kl = _ChainFn.apply(Zx, zf, hyp, m, L_S, P)
    P.ready = None
    if side is not None:
        P.ready = torch.cuda.Event()
        P.ready.record(side)
    return P, token


def layer_panel(P, token, x, f, Zx, zf, hyp):
    """PANEL half: mean (N'), var (N').  Waits (stream-side) for the CHAIN half of the same pass."""
    if P.ready is not None:
        torch.cuda.current_stream(x.device).wait_event(P.ready)
    return _PanelFn.apply(x, f, Zx, zf, hyp, token, P)


# ------------------------------------------------------------------------------------------------------------
# All layers of a model in one batch of CHAIN launches (include/mobocmf_hip.h, mobocmf_layers_chain_*): the chain is a
# serial string of latency-bound M x M kernels, the layers' chains do not depend on each other, so one z-batched
# sequence divides its length by the number of layers.  One autograd node carries every layer's chain; the per-layer
# PANEL nodes hang off its ``token``, so autograd runs the batched chain backward after all PANEL backwards.
# ------------------------------------------------------------------------------------------------------------
class ChainBatch:
    """Workspace + bookkeeping shared by ``layers_chain`` and the per-layer ``layer_panel_batched`` calls of one forward."""
    __slots__ = ("n", "kinds", "ds", "M", "branch", "jitters", "min_var", "blocks", "stride", "block_bytes", "infos",
                 "kls", "had_panel", "token", "panel_g", "tune")

    def chain_desc(self, z):
        return make_desc(self.kinds[z], self.ds[z], self.M, 1, 1, self.branch, False, self.jitters[z], self.min_var,
                         PHASE_CHAIN, tune=self.tune)

    def block(self, z):
        return self.blocks[z * self.stride:z * self.stride + self.block_bytes]


def _table(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


def _desc_table(descs):
    return (ctypes.POINTER(LayerDesc) * len(descs))(*[ctypes.pointer(d) for d in descs])


class _ChainsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, CB, *flat):
        lib = _lib.require_device()
        n = CB.n
        per = [[_prep(t) for t in flat[5 * z:5 * z + 5]] for z in range(n)]      # Zx, zf, hyp, m, L_S per layer
        dev = per[0][0].device
        descs = [CB.chain_desc(z) for z in range(n)]
        bb, st = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(lib.mobocmf_chain_block_bytes(ctypes.byref(descs[0]), ctypes.byref(bb), ctypes.byref(st)),
                   "mobocmf_chain_block_bytes")
        CB.block_bytes = bb.value
        CB.stride = (bb.value + 255) // 256 * 256
        CB.blocks = _poison(torch.empty(n * CB.stride, dtype=torch.uint8, device=dev))
        CB.kls = [_empty((), device=dev) for _ in range(n)]
        CB.had_panel = [0] * n
        CB.panel_g = [None] * n
        ctx.set_materialize_grads(False)
        token = torch.empty(1, dtype=torch.float64, device=dev)      # never read: it only orders the autograd nodes
        P = lambda i, none_ok=False: _table([0 if per[z][i] is None else per[z][i].data_ptr() for z in range(n)])
        rc = lib.mobocmf_layers_chain_forward(n, _desc_table(descs), P(0), P(1), P(2), P(3), P(4),
                                              _table([k.data_ptr() for k in CB.kls]),
                                              _table([t.data_ptr() for t in CB.infos]), _ptr(CB.blocks), CB.stride,
                                              CB.blocks.numel(), _stream())
        _lib.check(rc, "mobocmf_layers_chain_forward")
        ctx.CB = CB
        ctx.descs = descs
        # zf of layer z is the variational mean of layer z - 1 (the model's Z~_z = [Z_x, m_{z-1}]): its gradient is folded into
        # g_m[z - 1] inside the chain backward (had_panel |= 4) instead of by an autograd add of two M-vectors
        same = lambda a, b: a is not None and b is not None and (a is b or (a.data_ptr() == b.data_ptr() and a.shape == b.shape
                                                                            and a.stride() == b.stride()))
        ctx.fold = [z > 0 and CB.kinds[z] == 1 and same(flat[5 * z + 1], flat[5 * (z - 1) + 3]) and
                    bool(ctx.needs_input_grad[1 + 5 * z + 1]) and bool(ctx.needs_input_grad[1 + 5 * (z - 1) + 3])
                    for z in range(n)]
        ctx.has = [[t is not None for t in per[z]] for z in range(n)]
        ctx.save_for_backward(*[t for z in range(n) for t in per[z] if t is not None])
        return (token,) + tuple(CB.kls)

    @staticmethod
    def backward(ctx, g_token, *g_kls):
        lib = _lib.require_device()
        CB = ctx.CB
        n = CB.n
        it = iter(ctx.saved_tensors)
        per = [[next(it) if h else None for h in ctx.has[z]] for z in range(n)]
        dev = per[0][0].device
        gk = [_prep(g) if g is not None else torch.zeros((), dtype=torch.float64, device=dev) for g in g_kls]
        new = lambda *s: _empty(*s, device=dev)
        # a layer whose PANEL half ran left its share of g_hyp / g_zf in CB.panel_g: the chain accumulates into those buffers
        # (had_panel = 2) instead of autograd adding two tensors per layer afterwards
        had = list(CB.had_panel)
        g_zf, g_hyp = [], []
        for z in range(n):
            pg = CB.panel_g[z]
            if pg is not None:
                had[z] = 2
                g_hyp.append(pg[0])
                g_zf.append(pg[1] if per[z][1] is not None else None)
            else:
                g_hyp.append(new(per[z][2].numel()))
                g_zf.append(new(CB.M) if per[z][1] is not None else None)
        g_m = [new(CB.M) for _ in range(n)]
        g_LS = [new(CB.M, CB.M) for _ in range(n)]
        for z in range(n):
            if ctx.fold[z]:
                had[z] |= 4
        T = lambda ts: _table([0 if t is None else t.data_ptr() for t in ts])
        rc = lib.mobocmf_layers_chain_backward(n, _desc_table(ctx.descs), T([p[0] for p in per]), T([p[1] for p in per]),
                                               T([p[2] for p in per]), T(gk), (ctypes.c_int32 * n)(*had),
                                               T(g_zf), T(g_hyp), T(g_m), T(g_LS), _ptr(CB.blocks), CB.stride,
                                               CB.blocks.numel(), _stream())
        _lib.check(rc, "mobocmf_layers_chain_backward")
        out = [None]
        for z in range(n):
            out += [None, None if ctx.fold[z] else g_zf[z], g_hyp[z], g_m[z], g_LS[z]]
        return tuple(out)


def layers_chain(layers_params, kinds, branch, jitters, infos, min_var=MIN_VARIANCE):
    """CHAIN halves of all layers in one batch.  ``layers_params``: per layer (Zx, zf or None, hyp, m, L_S), equal M.
    Returns the ChainBatch (``.kls`` = per-layer KL tensors with gradient, ``.token`` for the PANEL calls)."""
    CB = ChainBatch()
    CB.n = len(layers_params)
    if not 1 <= CB.n <= 4:
        raise _lib.MobocmfError("layers_chain: 1..4 layers per batch")
    CB.kinds, CB.ds = list(kinds), [p[0].shape[1] for p in layers_params]
    CB.M = layers_params[0][0].shape[0]
    if any(p[0].shape[0] != CB.M for p in layers_params):
        raise _lib.MobocmfError("layers_chain: the layers must share the number of inducing points")
    CB.branch, CB.jitters, CB.min_var, CB.infos = branch, list(jitters), min_var, list(infos)
    CB.tune = current_tuning()      # one snapshot for the chains and every PANEL call that hangs off them
    out = _ChainsFn.apply(CB, *[t for p in layers_params for t in p])
    CB.token, CB.kls = out[0], list(out[1:])
    return CB


class _PanelBatchedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, Zx, zf, hyp, token, CB, z, xdiv, want_dx):
        lib = _lib.require_device()
        x, f, Zx, zf, hyp = (_prep(t) for t in (x, f, Zx, zf, hyp))
        Np = x.shape[0] * xdiv
        kind = CB.kinds[z]
        if x.shape[1] != CB.ds[z] or (kind == 1 and (f is None or f.numel() != Np)):
            raise _lib.MobocmfError("shape mismatch between the chain batch and a panel call")
        dev = x.device
        desc = make_desc(kind, CB.ds[z], CB.M, Np, xdiv, CB.branch, want_dx, CB.jitters[z], CB.min_var, PHASE_PANEL,
                         tune=CB.tune, probe=_probe_for(z))
        sb, cb = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(lib.mobocmf_panel_workspace_bytes(ctypes.byref(desc), ctypes.byref(sb), ctypes.byref(cb)),
                   "mobocmf_panel_workspace_bytes")
        saved = _poison(torch.empty(sb.value, dtype=torch.uint8, device=dev))
        scratch = scratch_buffer(cb.value, dev)
        mean, var = _empty(Np, device=dev), _empty(Np, device=dev)
        blk = CB.block(z)
        rc = lib.mobocmf_layer_panel_forward(ctypes.byref(desc), _ptr(x), _ptr(f), _ptr(Zx), _ptr(zf), _ptr(hyp),
                                             _ptr(mean), _ptr(var), _ptr(blk), blk.numel(), _ptr(saved), sb.value,
                                             _ptr(scratch), scratch.numel(), _stream())
        _lib.check(rc, "mobocmf_layer_panel_forward")
        ctx.CB, ctx.z, ctx.desc, ctx.saved_ws, ctx.sb, ctx.cb = CB, z, desc, saved, sb.value, cb.value
        ctx.save_for_backward(*[t for t in (x, f, Zx, zf, hyp) if t is not None])
        ctx.has_f = f is not None
        return mean, var

    @staticmethod
    def backward(ctx, g_mean, g_var):
        lib = _lib.require_device()
        CB, z, desc = ctx.CB, ctx.z, ctx.desc
        ts = list(ctx.saved_tensors)
        if ctx.has_f:
            x, f, Zx, zf, hyp = ts
        else:
            x, Zx, hyp = ts
            f = zf = None
        dev = x.device
        g_mean, g_var = _prep(g_mean), _prep(g_var)
        scratch = scratch_buffer(ctx.cb, dev)      # H / Hc / da go to the chain block, nothing of the scratch is kept
        new = lambda *s: _empty(*s, device=dev)
        g_f = new(desc.Np) if ctx.has_f else None
        g_zf = new(CB.M) if ctx.has_f else None
        g_hyp = new(hyp.numel())
        g_x = new(x.shape[0], desc.d) if desc.want_dx else None
        blk = CB.block(z)
        rc = lib.mobocmf_layer_panel_backward(ctypes.byref(desc), _ptr(x), _ptr(f), _ptr(Zx), _ptr(zf), _ptr(hyp),
                                              _ptr(g_mean), _ptr(g_var), _ptr(g_f), _ptr(g_zf), _ptr(g_hyp), _ptr(g_x),
                                              _ptr(blk), blk.numel(), _ptr(ctx.saved_ws), ctx.sb, _ptr(scratch),
                                              scratch.numel(), _stream())
        _lib.check(rc, "mobocmf_layer_panel_backward")
        CB.had_panel[z] = 1
        # the K_mn share of g_hyp / g_zf is handed to the chain node (which adds its K_mm share into the same buffers), and
        # None for the token: the edge to the chain node orders the backward passes, no gradient has to flow through it
        CB.panel_g[z] = (g_hyp, g_zf)
        return g_x, g_f, None, None, None, None, None, None, None, None


def layer_panel_batched(CB, z, x, f, Zx, zf, hyp, xdiv=1, want_dx=False):
    """PANEL half of layer ``z`` of a chain batch: mean (N'), var (N')."""
    return _PanelBatchedFn.apply(x, f, Zx, zf, hyp, CB.token, CB, z, xdiv, want_dx)


class FrozenChain:
    """CHAIN state of one layer for FIXED parameters (acquisition optimisation): computed once, then copied to the
    front of the ``saved`` buffer of every PANEL call; gradients flow to the layer inputs only."""
    __slots__ = ("kind", "d", "M", "branch", "jitter", "min_var", "state", "kl", "info", "Zx", "zf", "hyp")


def freeze_chain(Zx, zf, hyp, m, L_S, kind, branch=1, jitter=JITTER, min_var=MIN_VARIANCE):
    lib = _lib.require_device()
    with torch.no_grad():
        Zx, zf, hyp, m, L_S = (_prep(None if t is None else t.detach()) for t in (Zx, zf, hyp, m, L_S))
        fc = FrozenChain()
        fc.kind, fc.d, fc.M, fc.branch, fc.jitter, fc.min_var = kind, Zx.shape[1], Zx.shape[0], branch, jitter, min_var
        fc.Zx, fc.zf, fc.hyp = Zx, zf, hyp
        dev = Zx.device
        desc = make_desc(kind, fc.d, fc.M, 1, 1, branch, False, jitter, min_var, PHASE_CHAIN)
        nb = ctypes.c_size_t()
        _lib.check(lib.mobocmf_layer_chain_state_bytes(ctypes.byref(desc), ctypes.byref(nb)), "chain_state_bytes")
        sb, cb = workspace_bytes(desc)
        saved = _poison(torch.empty(sb, dtype=torch.uint8, device=dev))
        scratch = scratch_buffer(cb, dev)
        fc.kl = _empty((), device=dev)
        fc.info = torch.zeros((), dtype=torch.int32, device=dev)
        rc = lib.mobocmf_layer_forward(ctypes.byref(desc), None, None, _ptr(Zx), _ptr(zf), _ptr(hyp), _ptr(m), _ptr(L_S),
                                       None, None, _ptr(fc.kl), _ptr(fc.info), _ptr(saved), sb, _ptr(scratch),
                                       scratch.numel(), _stream())
        _lib.check(rc, "mobocmf_layer_forward[chain]")
        fc.state = saved[:nb.value].clone()
    return fc


def layer_panel_frozen(fc, x, f, xdiv=1, want_dx=False):
    """mean (N'), var (N') against a FrozenChain; differentiable w.r.t. f (and x if want_dx) only."""
    P = LayerPass()
    P.kind, P.d, P.M, P.Np, P.xdiv, P.branch = fc.kind, fc.d, fc.M, x.shape[0] * xdiv, xdiv, fc.branch
    P.jitter, P.min_var, P.want_dx, P.info = fc.jitter, fc.min_var, want_dx, fc.info
    P.bscratch = P.main = P.side = P.ready = None
    P.kl, P.frozen = fc.kl, True
    P.tune, P.probe = current_tuning(), _probe_for(None)
    P.sb, P.cb = workspace_bytes(P.desc(PHASE_PANEL))
    P.saved = _poison(torch.empty(P.sb, dtype=torch.uint8, device=x.device))
    P.saved[:fc.state.numel()].copy_(fc.state)
    return _PanelFn.apply(x, f, fc.Zx, fc.zf, fc.hyp, None, P)


def layer_forward(x, f, Zx, zf, hyp, m, L_S, kind, xdiv=1, branch=0, jitter=JITTER, min_var=MIN_VARIANCE,
                  want_dx=False, info_out=None):
    """mean (N'), var (N'), kl ()  --  N' = x.shape[0] * xdiv."""
    return _LayerFn.apply(x, f, Zx, zf, hyp, m, L_S, kind, xdiv, branch, jitter, min_var, want_dx, info_out)


def predictive_covariance(x, f, Zx, zf, hyp, m, L_S, kind, xdiv=1, jitter=JITTER, chain=None):
    """Full eval-branch predictive covariance (N' x N') -- K10: symmetric rank-M updates on the MFMA, lower tiles only,
    any N'.  Returns (mean, cov).  ``chain``: a FrozenChain of the same layer (then m, L_S are not needed and the M x M
    chain is not recomputed).  No autograd."""
    lib = _lib.require_device()
    with torch.no_grad():
        x, f, Zx, zf, hyp, m, L_S = (_prep(t) for t in (x, f, Zx, zf, hyp, m, L_S))
        d, M = Zx.shape[1], Zx.shape[0]
        Np = x.shape[0] * xdiv
        dev = x.device
        if chain is None:
            chain = freeze_chain(Zx, zf, hyp, m, L_S, kind, branch=1, jitter=jitter)
        desc = make_desc(kind, d, M, Np, xdiv, 1, False, jitter, MIN_VARIANCE)
        mean, _ = layer_panel_frozen(chain, x, f, xdiv=xdiv)
        cb = ctypes.c_size_t()
        _lib.check(lib.mobocmf_predictive_covariance_workspace_bytes(ctypes.byref(desc), ctypes.byref(cb)),
                   "mobocmf_predictive_covariance_workspace_bytes")
        scratch = scratch_buffer(cb.value, dev)
        cov = _empty(Np, Np, device=dev)
        _lib.check(lib.mobocmf_predictive_covariance(ctypes.byref(desc), _ptr(x), _ptr(f), _ptr(Zx), _ptr(zf), _ptr(hyp),
                                                     _ptr(cov), Np, _ptr(chain.state), chain.state.numel(), _ptr(scratch),
                                                     scratch.numel(), _stream()), "mobocmf_predictive_covariance")
    return mean, cov


def _propagate_backward(ctx, g, add_mean=None, add_var=None):
    """g_mean / g_var of the previous layer's moments: zeros beyond the propagated prefix (one launch either way).
    add_mean / add_var: the gradients the pass-through copies of the moments received (``through=True``), added in the same
    launch."""
    lib = _lib.require_device()
    var, eps = ctx.saved_tensors
    if g is None:        # nothing came back through the propagated samples
        if add_mean is None and add_var is None:
            return None, None
        z = lambda t: _prep(t) if t is not None else torch.zeros_like(var)
        return z(add_mean), z(add_var)
    g = _prep(g)
    add_mean = None if add_mean is None else _prep(add_mean.reshape(-1))
    add_var = None if add_var is None else _prep(add_var.reshape(-1))
    gm, gv = _empty_like(var), _empty_like(var)
    _lib.check(lib.mobocmf_propagate_backward_prefix(_ptr(var), _ptr(eps), _ptr(g), _ptr(gm), _ptr(gv), eps.numel(),
                                                     ctx.div, var.numel(), _ptr(add_mean), _ptr(add_var), _stream()),
               "mobocmf_propagate_backward_prefix")
    return gm, gv


class _PropagateFn(torch.autograd.Function):
    """f~ = mean + sqrt(var) eps.  through=True additionally returns pass-through aliases of (mean, var): a caller that
    also scores the SAME moments (the ELBO's data term of the previous layer's own fidelity) uses those, so the moments have
    one consumer node and their two gradient contributions are summed inside the propagate-backward launch."""

    @staticmethod
    def forward(ctx, mean, var, eps, div, through):
        lib = _lib.require_device()
        shape = mean.shape
        mean, var, eps = _prep(mean.reshape(-1)), _prep(var.reshape(-1)), _prep(eps.reshape(-1))
        n = eps.numel()
        if mean.numel() * div < n or n % div or var.numel() != mean.numel():
            raise _lib.MobocmfError("propagate: eps must have k*div entries, k <= mean.numel() (a prefix of the rows)")
        out = _empty(n, device=mean.device)
        _lib.check(lib.mobocmf_propagate_forward(_ptr(mean), _ptr(var), _ptr(eps), _ptr(out), n, div, _stream()),
                   "mobocmf_propagate_forward")
        ctx.save_for_backward(var, eps)
        ctx.div, ctx.shape = div, shape
        ctx.set_materialize_grads(False)
        if through:
            return out, mean.view(shape), var.view(shape)
        return out

    @staticmethod
    def backward(ctx, g, g_mean_t=None, g_var_t=None):
        gm, gv = _propagate_backward(ctx, g, g_mean_t, g_var_t)
        r = lambda t: None if t is None else t.view(ctx.shape)
        return r(gm), r(gv), None, None, None


class _PropagateRngFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, var, rng_state, n_out, div, through):
        lib = _lib.require_device()
        shape = mean.shape
        mean, var = _prep(mean.reshape(-1)), _prep(var.reshape(-1))
        if mean.numel() * div < n_out or n_out % div or var.numel() != mean.numel() or rng_state.dtype != torch.int64 or \
                rng_state.numel() != 3 or not rng_state.is_cuda:
            raise _lib.MobocmfError("propagate_rng: n_out must be k*div, k <= mean.numel(), and rng_state three int64 on the GPU")
        out, eps = _empty(n_out, device=mean.device), _empty(n_out, device=mean.device)
        _lib.check(lib.mobocmf_propagate_rng_forward(_ptr(mean), _ptr(var), _ptr(rng_state), _ptr(out), _ptr(eps), n_out, div,
                                                     _stream()), "mobocmf_propagate_rng_forward")
        ctx.save_for_backward(var, eps)
        ctx.div, ctx.shape = div, shape
        ctx.mark_non_differentiable(eps)
        ctx.set_materialize_grads(False)      # no zero-filled "gradient of eps" (a fill launch per backward)
        if through:
            return out, eps, mean.view(shape), var.view(shape)
        return out, eps

    @staticmethod
    def backward(ctx, g, _g_eps, g_mean_t=None, g_var_t=None):
        gm, gv = _propagate_backward(ctx, g, g_mean_t, g_var_t)
        r = lambda t: None if t is None else t.view(ctx.shape)
        return r(gm), r(gv), None, None, None, None


def propagate_rng(mean, var, rng_state, n_out, div=1):
    """f~[n] = mean[n/div] + sqrt(var[n/div]) * eps[n] with eps ~ N(0, 1) drawn inside the launch (Philox4x32-10 keyed by
    rng_state = int64 [seed, calls, ticket] on the device; every call advances ``calls``).  Returns (f~, eps)."""
    return _PropagateRngFn.apply(mean, var, rng_state, int(n_out), int(div), False)


def propagate_rng_through(mean, var, rng_state, n_out, div=1):
    """``propagate_rng`` + pass-through aliases: returns (f~, eps, mean', var'); score mean' / var' instead of mean / var."""
    return _PropagateRngFn.apply(mean, var, rng_state, int(n_out), int(div), True)


def propagate(mean, var, eps, div=1):
    """f~[n] = mean[n/div] + sqrt(var[n/div]) * eps[n]   (mfdgp_hidden_layer.py:263-274).  ``eps`` may cover only a prefix of
    the rows of (mean, var): the rest is not propagated and gets zero gradient."""
    return _PropagateFn.apply(mean, var, eps, div, False)


def propagate_through(mean, var, eps, div=1):
    """``propagate`` + pass-through aliases: returns (f~, mean', var'); score mean' / var' instead of mean / var."""
    return _PropagateFn.apply(mean, var, eps, div, True)


class _ElboDataFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, var, y, fid, tau, level, div, lo=0.0, hi=0.0):
        lib = _lib.require_device()
        mean, var, y, fid, tau = (_prep(t.reshape(-1)) for t in (mean, var, y, fid, tau))
        n = mean.numel()
        if y.numel() * div != n or fid.numel() != y.numel():
            raise _lib.MobocmfError("elbo_data: shape mismatch")
        out = _empty((), device=mean.device)
        scratch = scratch_buffer(8192, mean.device)
        _lib.check(lib.mobocmf_elbo_data_interval_forward(_ptr(mean), _ptr(var), _ptr(y), _ptr(fid), _ptr(tau), float(lo),
                                                          float(hi), float(level), n, div, _ptr(out), _ptr(scratch),
                                                          scratch.numel(), _stream()),
                   "mobocmf_elbo_data_forward")
        ctx.save_for_backward(mean, var, y, fid, tau)
        ctx.level, ctx.div, ctx.lo, ctx.hi = float(level), div, float(lo), float(hi)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.require_device()
        mean, var, y, fid, tau = ctx.saved_tensors
        g = _prep(g)
        gm, gv = _empty_like(mean), _empty_like(var)
        gt = _empty_like(tau)
        scratch = scratch_buffer(8192, mean.device)
        _lib.check(lib.mobocmf_elbo_data_interval_backward(_ptr(mean), _ptr(var), _ptr(y), _ptr(fid), _ptr(tau), ctx.lo,
                                                           ctx.hi, ctx.level, mean.numel(), ctx.div, _ptr(g), _ptr(gm),
                                                           _ptr(gv), _ptr(gt), _ptr(scratch), scratch.numel(), _stream()),
                   "mobocmf_elbo_data_backward")
        return gm, gv, None, None, gt.reshape(ctx.saved_tensors[4].shape), None, None, None, None


def elbo_data(mean, var, y, fid, tau, level, div=1, interval=None):
    """(1/div) sum_{fid==level} E_q[log N(y | f, tau)]   (variational_elbo_mf.py:31-35).
    ``interval=(lo, hi)``: ``tau`` is the RAW noise parameter of an Interval constraint; the sigmoid transform and its
    chain rule run inside the kernels (six element-wise launches less per fidelity and step)."""
    if interval is not None:
        return _ElboDataFn.apply(mean, var, y, fid, tau, level, div, float(interval[0]), float(interval[1]))
    return _ElboDataFn.apply(mean, var, y, fid, tau, level, div)


class _ShortcutVarFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, L_S, min_var):
        lib = _lib.require_device()
        L_S = _prep(L_S)
        M = L_S.shape[0]
        var = _empty(M, device=L_S.device)
        _lib.check(lib.mobocmf_shortcut_var_forward(_ptr(L_S), M, float(min_var), _ptr(var), _stream()),
                   "mobocmf_shortcut_var_forward")
        ctx.save_for_backward(L_S, var)
        ctx.min_var = float(min_var)
        return var

    @staticmethod
    def backward(ctx, g):
        lib = _lib.require_device()
        L_S, var = ctx.saved_tensors
        g = _prep(g)
        gL = _empty_like(L_S)
        _lib.check(lib.mobocmf_shortcut_var_backward(_ptr(L_S), _ptr(var), _ptr(g), L_S.shape[0], ctx.min_var, _ptr(gL),
                                                     _stream()), "mobocmf_shortcut_var_backward")
        return gL, None


def shortcut_var(L_S, min_var=MIN_VARIANCE):
    """diag(L_S L_S^T) floored at min_var: the marginal variances of q(u) (GPyTorch's equal-inputs shortcut)."""
    return _ShortcutVarFn.apply(L_S, min_var)


class _ElboCombineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, n_data, *terms):
        lib = _lib.require_device()
        terms = [_prep(t.reshape(())) for t in terms]
        data, kls = terms[:n_data], terms[n_data:]
        out = _empty(2, device=terms[0].device)
        A = (ctypes.c_void_p * max(len(data), 1))(*[t.data_ptr() for t in data])
        B = (ctypes.c_void_p * max(len(kls), 1))(*[t.data_ptr() for t in kls])
        _lib.check(lib.mobocmf_elbo_combine_forward(len(data), A, len(kls), B, float(scale), _ptr(out), _stream()),
                   "mobocmf_elbo_combine_forward")
        ctx.scale, ctx.n_data, ctx.n = float(scale), n_data, len(terms)
        ctx.keep = terms               # the table holds raw pointers: keep the scalars alive until the launch is enqueued
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_elbo, g_skl):
        lib = _lib.require_device()
        ref = g_elbo if g_elbo is not None else g_skl
        g = _empty(2, device=ref.device)
        ge = None if g_elbo is None else _prep(g_elbo)
        gs = None if g_skl is None else _prep(g_skl)
        _lib.check(lib.mobocmf_elbo_combine_backward(_ptr(ge), _ptr(gs), ctx.scale, _ptr(g), _stream()),
                   "mobocmf_elbo_combine_backward")
        return (None, None) + (g[0],) * ctx.n_data + (g[1],) * (ctx.n - ctx.n_data)


def elbo_combine(data_terms, kls, scale):
    """(sum data - scale * sum kl, scale * sum kl) in one launch (and one in backward)."""
    if len(data_terms) > 8 or len(kls) > 8:
        d = sum(data_terms) if data_terms else 0.0
        k = sum(kls) * scale if kls else 0.0
        return d - k, k
    return _ElboCombineFn.apply(scale, len(data_terms), *data_terms, *kls)


class _ElboFusedFn(torch.autograd.Function):
    """(elbo, scaled_kl, -elbo) of variational_elbo_mf.py:24-51 in one launch, one more in backward (mobocmf_elbo_forward)."""

    @staticmethod
    def forward(ctx, scale, B, specs, y, fid, n_kl, *tensors):
        # specs[l] = None | (div, lo, hi, rows): the layer holds the first ``rows`` base rows of the batch
        lib = _lib.require_device()
        L = len(specs)
        y, fid = _prep(y.reshape(-1)), _prep(fid.reshape(-1))
        dev = y.device
        means, vars_, raws, it = [None] * L, [None] * L, [None] * L, iter(tensors)
        for l, sp in enumerate(specs):
            if sp is not None:
                means[l], vars_[l], raws[l] = _prep(next(it).reshape(-1)), _prep(next(it).reshape(-1)), _prep(next(it).reshape(-1))
                if means[l].numel() != sp[3] * sp[0] or vars_[l].numel() != sp[3] * sp[0] or not 0 <= sp[3] <= B:
                    raise _lib.MobocmfError("elbo: layer %d holds %d rows, expected %d" % (l, means[l].numel(), sp[3] * sp[0]))
        kls = [_prep(next(it).reshape(())) for _ in range(n_kl)]
        if y.numel() != B or fid.numel() != B:
            raise _lib.MobocmfError("elbo: target / fidelities shape mismatch")
        out = _empty(3, device=dev)
        scratch = scratch_buffer(8 * 512 * 8, dev)
        T = lambda ts: _table([0 if t is None else t.data_ptr() for t in ts])
        div = (ctypes.c_int32 * L)(*[1 if sp is None else sp[0] for sp in specs])
        lo = (ctypes.c_double * L)(*[0.0 if sp is None else sp[1] for sp in specs])
        hi = (ctypes.c_double * L)(*[0.0 if sp is None else sp[2] for sp in specs])
        rows = (ctypes.c_int64 * L)(*[B if sp is None else sp[3] for sp in specs])
        _lib.check(lib.mobocmf_elbo_forward(L, T(means), T(vars_), div, T(raws), lo, hi, _ptr(y), _ptr(fid), B, rows, n_kl,
                                            T(kls) if n_kl else None, float(scale), _ptr(out), _ptr(scratch),
                                            scratch.numel(), _stream()), "mobocmf_elbo_forward")
        ctx.set_materialize_grads(False)
        ctx.meta = (float(scale), B, specs, n_kl, [None if t is None else t.shape for t in tensors])
        ctx.save_for_backward(y, fid, *[t for l in range(L) if specs[l] is not None for t in (means[l], vars_[l], raws[l])])
        elbo, skl, neg = out[0], out[1], out[2]
        ctx.mark_non_differentiable(neg)
        return elbo, skl, neg

    @staticmethod
    def backward(ctx, g_elbo, g_skl, _g_neg):
        lib = _lib.require_device()
        scale, B, specs, n_kl, shapes = ctx.meta
        L = len(specs)
        y, fid = ctx.saved_tensors[:2]
        it = iter(ctx.saved_tensors[2:])
        dev = y.device
        means, vars_, raws = [None] * L, [None] * L, [None] * L
        gm, gv, gr = [None] * L, [None] * L, [None] * L
        for l, sp in enumerate(specs):
            if sp is not None:
                means[l], vars_[l], raws[l] = next(it), next(it), next(it)
                gm[l], gv[l], gr[l] = _empty_like(means[l]), _empty_like(vars_[l]), _empty_like(raws[l])
        gkl = _empty(1, device=dev)
        ge = None if g_elbo is None else _prep(g_elbo)
        gs = None if g_skl is None else _prep(g_skl)
        scratch = scratch_buffer(8 * 512 * 8, dev)
        T = lambda ts: _table([0 if t is None else t.data_ptr() for t in ts])
        div = (ctypes.c_int32 * L)(*[1 if sp is None else sp[0] for sp in specs])
        lo = (ctypes.c_double * L)(*[0.0 if sp is None else sp[1] for sp in specs])
        hi = (ctypes.c_double * L)(*[0.0 if sp is None else sp[2] for sp in specs])
        rows = (ctypes.c_int64 * L)(*[B if sp is None else sp[3] for sp in specs])
        _lib.check(lib.mobocmf_elbo_backward(L, T(means), T(vars_), div, T(raws), lo, hi, _ptr(y), _ptr(fid), B, rows, scale,
                                             _ptr(ge), _ptr(gs), T(gm), T(gv), T(gr), _ptr(gkl), _ptr(scratch),
                                             scratch.numel(), _stream()), "mobocmf_elbo_backward")
        grads, k = [], 0
        for l, sp in enumerate(specs):
            if sp is not None:
                grads += [gm[l].reshape(shapes[k]), gv[l].reshape(shapes[k + 1]), gr[l].reshape(shapes[k + 2])]
                k += 3
        grads += [gkl[0]] * n_kl
        return (None,) * 6 + tuple(grads)


ELBO_MAX_LAYERS = 8


def elbo_fused(layers, y, fid, kls, scale):
    """``layers``: per fidelity level None or (mean, var, raw_noise, div, lo, hi[, rows]) -- hi <= lo: ``raw_noise`` is the
    noise itself; ``rows`` (default: all of y): the layer holds the first ``rows`` rows of the batch only (rows * div
    entries).  Returns (elbo, scaled_kl, -elbo); the last one carries no gradient (the loss value a training step reports)."""
    B = int(y.numel())
    specs = [None if lay is None else (int(lay[3]), float(lay[4]), float(lay[5]), int(lay[6]) if len(lay) > 6 else B)
             for lay in layers]
    tensors = [t for lay in layers if lay is not None for t in lay[:3]]
    return _ElboFusedFn.apply(float(scale), int(y.numel()), specs, y, fid, len(kls), *tensors, *kls)


# ---------------------------------------------------------------------------------------------------------------------
# Conditioned training: the theta / omega factor losses and the glue around them, one launch each way
# (include/mobocmf_hip.h: mobocmf_cond_factors_forward, _scale_segments, _gather_segments, _scalar_combine).
# ---------------------------------------------------------------------------------------------------------------------
def _ptr_table(ts):
    return (ctypes.c_void_p * max(len(ts), 1))(*[0 if t is None else t.data_ptr() for t in ts])


class _SplitRowsFn(torch.autograd.Function):
    """(mean, var) -> their consecutive row ranges [0, c0), [c0, c0 + c1), ... as views (no launch); the backward writes the
    ranges' gradients -- zeros where a range got none -- into one buffer in ONE launch (autograd's slice backward: a zero fill,
    a copy and an add per range and tensor)."""

    @staticmethod
    def forward(ctx, mean, var, *counts):
        ctx.shapes = (mean.shape, var.shape)
        mean, var = _prep(mean.reshape(-1)), _prep(var.reshape(-1))
        if sum(counts) != mean.numel() or var.numel() != mean.numel():
            raise _lib.MobocmfError("split_rows: the counts must add up to the number of rows")
        ctx.counts = tuple(int(c) for c in counts)
        ctx.set_materialize_grads(False)
        outs, off = [], 0
        for t in (mean, var):
            off = 0
            for c in ctx.counts:
                outs.append(t[off:off + c])
                off += c
        ctx.dev = mean.device
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        lib = _lib.require_device()
        k, n = len(ctx.counts), sum(ctx.counts)
        if all(g is None for g in gs):
            return (None, None) + (None,) * k
        gs = [None if g is None else _prep(g.reshape(-1)) for g in gs]
        out = _empty(2 * n, device=ctx.dev)
        sizes = (ctypes.c_int64 * (2 * k))(*(list(ctx.counts) * 2))
        _lib.check(lib.mobocmf_gather_segments(2 * k, _ptr_table(gs), sizes, _ptr(out), _stream()), "mobocmf_gather_segments")
        return (out[:n].view(ctx.shapes[0]), out[n:].view(ctx.shapes[1])) + (None,) * k


def split_rows(mean, var, counts):
    """Consecutive row ranges of (mean, var): returns ([mean ranges], [var ranges])."""
    outs = _SplitRowsFn.apply(mean, var, *counts)
    k = len(counts)
    return list(outs[:k]), list(outs[k:])


class _CondFactorsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, front, thr, *rows):
        lib = _lib.require_device()
        n_obj, n_con, P, T, coef_c, coef_1mc = meta
        rows = [_prep(r.reshape(-1)) for r in rows]
        if len(rows) != 2 * (n_obj + n_con) or any(r.numel() != T for r in rows):
            raise _lib.MobocmfError("cond_factors: one mean and one variance row of T entries per objective / constraint")
        dev = rows[0].device
        fm, fv = rows[:n_obj], rows[n_obj:2 * n_obj]
        cm, cv = rows[2 * n_obj:2 * n_obj + n_con], rows[2 * n_obj + n_con:]
        grads = _empty(len(rows), T, device=dev)
        g = [grads[i] for i in range(len(rows))]
        loss = _empty((), device=dev)
        front = None if front is None else _prep(front)
        thr = None if thr is None else _prep(thr)
        rc = lib.mobocmf_cond_factors_forward(n_obj, n_con, P, T, _ptr_table(fm), _ptr_table(fv), _ptr_table(cm), _ptr_table(cv),
                                              _ptr(front), _ptr(thr), coef_c, coef_1mc, _ptr(loss),
                                              _ptr_table(g[:n_obj]), _ptr_table(g[n_obj:2 * n_obj]),
                                              _ptr_table(g[2 * n_obj:2 * n_obj + n_con]), _ptr_table(g[2 * n_obj + n_con:]),
                                              _stream())
        _lib.check(rc, "mobocmf_cond_factors_forward")
        ctx.save_for_backward(grads)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.require_device()
        (grads,) = ctx.saved_tensors
        n, T = grads.shape
        out = _empty(n, T, device=grads.device)
        one = (ctypes.c_int64 * 1)(n * T)
        _lib.check(lib.mobocmf_scale_segments(1, _ptr_table([grads]), _ptr_table([out]), one, None, _ptr(_prep(g)), _stream()),
                   "mobocmf_scale_segments")
        return (None, None, None) + tuple(out[i] for i in range(n))


def cond_factors(fs_mean, fs_var, cs_mean, cs_var, front, thr, coef_c, coef_1mc):
    """sum_{p, t} [coef_c c + coef_1mc (1 - c)], c(p, t) = prod_k Phi((cs_mean_k[t] - thr_k) / sd) prod_j Phi((front[p, j] -
    fs_mean_j[t]) / sd): the omega factors (blackbox_mfdgp_fitter.py:235-243; coef_c = log eps) and, with no objective rows and
    P = 1, the theta factors (:227-233; coef_c = log(1 - eps)).  Lists of row tensors (T entries each); one launch forward
    (which also forms the gradients), one backward."""
    n_obj, n_con = len(fs_mean), len(cs_mean)
    T = (fs_mean[0] if n_obj else cs_mean[0]).numel()
    P = front.shape[0] if (front is not None and n_obj) else 1
    meta = (n_obj, n_con, int(P), int(T), float(coef_c), float(coef_1mc))
    return _CondFactorsFn.apply(meta, front if n_obj else None, thr if n_con else None,
                                *(list(fs_mean) + list(fs_var) + list(cs_mean) + list(cs_var)))


class _ScalarCombineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coefs, *terms):
        lib = _lib.require_device()
        terms = [_prep(t.reshape(1)) for t in terms]
        n = len(terms)
        out = _empty((), device=terms[0].device)
        ctx.coefs, ctx.dev = tuple(float(c) for c in coefs), terms[0].device
        _lib.check(lib.mobocmf_scalar_combine(n, _ptr_table(terms), (ctypes.c_double * n)(*ctx.coefs), _ptr(out), _stream()),
                   "mobocmf_scalar_combine")
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.require_device()
        n = len(ctx.coefs)
        one = _ones_scalar(ctx.dev)
        out = _empty(n, device=ctx.dev)
        sizes = (ctypes.c_int64 * n)(*([1] * n))
        _lib.check(lib.mobocmf_scale_segments(n, _ptr_table([one] * n), _ptr_table([out[i:i + 1] for i in range(n)]), sizes,
                                              (ctypes.c_double * n)(*ctx.coefs), _ptr(_prep(g)), _stream()),
                   "mobocmf_scale_segments")
        return (None,) + tuple(out[i] for i in range(n))


_ones = {}


def _ones_scalar(dev):
    t = _ones.get(dev)
    if t is None:
        t = torch.ones(1, dtype=torch.float64, device=dev)
        _ones[dev] = t
    return t


SEG_MAX = 32      # segments one mobocmf_scalar_combine / _scale_segments / _gather_segments launch takes


def scalar_combine(terms, coefs):
    """sum_i coefs[i] * terms[i] for 0-dim device tensors: one launch (and one for all the terms' gradients); more than
    SEG_MAX terms (a conditioned loss over 16+ black-boxes on one rank) are combined in chunks, then the chunk sums."""
    terms, coefs = list(terms), list(coefs)
    while len(terms) > SEG_MAX:
        part = [_ScalarCombineFn.apply(tuple(coefs[i:i + SEG_MAX]), *terms[i:i + SEG_MAX]) for i in range(0, len(terms), SEG_MAX)]
        terms, coefs = part, [1.0] * len(part)
    return _ScalarCombineFn.apply(tuple(coefs), *terms)


class _AcqMomentsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu_t, var_t, S):
        lib = _lib.require_device()
        mu_t, var_t = _prep(mu_t.reshape(-1)), _prep(var_t.reshape(-1))
        T = mu_t.numel() // S
        mus = _empty(T, device=mu_t.device)
        vs = _empty_like(mus)
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
        gm, gv = _empty_like(mu_t), _empty_like(mu_t)
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
        out = _empty_like(vu)
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


class FusedAdam:
    """torch.optim.Adam (defaults: no weight decay, no amsgrad) over all parameters of a model in ONE launch of
    mobocmf_adam_multi, with the step count on the device: capturable, and the update a captured graph replays.
    Duck-types the little of torch.optim.Optimizer the fitter uses (zero_grad / step / state)."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for group in params for p in group["params"]] if params and isinstance(params[0], dict) \
            else list(params)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        dev = self.params[0].device
        self.steps_done = torch.zeros((), dtype=torch.int64, device=dev)
        self.state = {"__step__": {"step": self.steps_done}}
        for i, p in enumerate(self.params):
            if p.dtype != torch.float64 or not p.is_cuda:
                raise _lib.MobocmfError("FusedAdam: float64 GPU parameters only")
            self.state[i] = {"exp_avg": torch.zeros_like(p, memory_format=torch.contiguous_format),
                             "exp_avg_sq": torch.zeros_like(p, memory_format=torch.contiguous_format)}

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self):
        lib = _lib.require_device()
        idx = [i for i, p in enumerate(self.params) if p.grad is not None]
        if not idx:
            return
        n = len(idx)
        arr = lambda vals: (ctypes.c_void_p * n)(*vals)
        grads = [self.params[i].grad if self.params[i].grad.is_contiguous() else self.params[i].grad.contiguous() for i in idx]
        for i in idx:
            if not self.params[i].is_contiguous():
                raise _lib.MobocmfError("FusedAdam: parameters must be contiguous")
        P = arr([self.params[i].data_ptr() for i in idx])
        G = arr([g.data_ptr() for g in grads])
        M = arr([self.state[i]["exp_avg"].data_ptr() for i in idx])
        V = arr([self.state[i]["exp_avg_sq"].data_ptr() for i in idx])
        N = (ctypes.c_int64 * n)(*[self.params[i].numel() for i in idx])
        _lib.check(lib.mobocmf_adam_multi(n, P, G, M, V, N, self.lr, self.betas[0], self.betas[1], self.eps,
                                          _ptr(self.steps_done), _stream()), "mobocmf_adam_multi")


def check_info(info):
    """Synchronising: raises if the last Cholesky reported a non-positive pivot."""
    lib = _lib.require_device()
    piv = ctypes.c_int32()
    rc = lib.mobocmf_check_info(_ptr(info), ctypes.byref(piv), _stream())
    if rc == _lib.NOT_PD:
        return piv.value
    _lib.check(rc, "mobocmf_check_info")
    return 0


class InLaunchWaitAbandoned(FloatingPointError):
    """A one-launch form (cooperative step, one-launch Cholesky) gave up a bounded in-launch wait: its workgroups were not resident
    together (too many such launches in flight on the device, or a partition with few CUs).  The call's results are invalid."""


def raise_if_abandoned(pivot, what="Cholesky"):
    """``pivot`` as returned by check_info: -1 is not a failed pivot but an abandoned in-launch wait (include/mobocmf_hip.h,
    mobocmf_check_info)."""
    if pivot < 0:
        raise InLaunchWaitAbandoned("%s: the one-launch form abandoned a bounded in-launch wait (its workgroups were not resident "
                                    "together); repeat the call with F.tuning(potrf_cols=4) or less concurrent work" % what)


def gemm_colstat_rows(Mr, Nc, Kd, tri=0, tune=None):
    """Partial rows the column-statistics epilogue writes for a product of that shape (two per row block of the tile
    height the tuning gives: mobocmf_gemm_colstat_rows)."""
    rows = ctypes.c_int32()
    snap = tune if tune is not None else current_tuning()
    _lib.check(_lib.load().mobocmf_gemm_colstat_rows(tri, Mr, Nc, Kd, ctypes.byref(snap), ctypes.byref(rows)),
               "mobocmf_gemm_colstat_rows")
    return rows.value


# ---- host-side defaults of the knobs (process-wide in THIS module; the library itself holds nothing -- see `tuning` above).
def set_mid_gemm_max(n=1024):
    """Largest dimension of a plain product that runs on the mid-size (one launch, no k-slicing) kernel; 0 = off."""
    if not 0 <= int(n) <= 4096:
        raise _lib.MobocmfError("set_mid_gemm_max: 0..4096")
    _set_default(mid_gemm_max=n)


def set_mid_gemm_waves(n=32):
    """Form of the mid-size product kernel: 32 (32 x 64 tiles, 64-k stages; default), 8 or 4 (64 x 64 tiles on that many
    wavefronts)."""
    if int(n) not in (4, 8, 32):
        raise _lib.MobocmfError("set_mid_gemm_waves: 4 | 8 | 32")
    _set_default(mid_gemm_waves=n)


def set_syrk_workgroups(n=0):
    """Workgroups a k-sliced weighted syrk may occupy (0 = by shape).  Sizes follow the tuning a call carries."""
    if int(n) != 0 and not 16 <= int(n) <= 4096:
        raise _lib.MobocmfError("set_syrk_workgroups: 0 | 16..4096")
    _set_default(syrk_workgroups=n)


def set_sparse_backward(on=True):
    """Skip the 128-column blocks of a layer backward whose upstream gradients are all exactly zero (default on).  Off = the
    dense backward: A/B timing and the parity tests.  Read when the FORWARD of a layer call is issued."""
    _set_default(sparse_backward=1 if on else 0)


_block_act_tls = threading.local()


def set_block_activity(act=None):
    """Tests / tools: int32 CUDA tensor (one word per 128 columns) that the standalone gemm_f64_epilogue / syrk_weighted calls
    of THIS thread pass as their col_activity / k_activity argument until it is cleared with None."""
    if act is not None:
        assert act.dtype == torch.int32 and act.is_cuda and act.is_contiguous()
    _block_act_tls.act = act


def set_tile_rows(rows=0, pair_mode=0):
    """Tile height of the M x N' panel products: 0 automatic, 64 or 128; row-block pairing 0 automatic, 1 never, 2 always."""
    if int(rows) not in (0, 64, 128) or not 0 <= int(pair_mode) <= 2:
        raise _lib.MobocmfError("set_tile_rows: rows 0 | 64 | 128, pair_mode 0..2")
    _set_default(tile_rows=rows, pair_mode=pair_mode)


def set_potrf_cols(cols=0):
    """The blocked Cholesky: 0 (default) all 64-column steps in one launch where it applies; 4 or 1: a launch pair per 64
    columns with that many columns per hand-over of the panel kernel."""
    if int(cols) not in (0, 1, 4):
        raise _lib.MobocmfError("set_potrf_cols: 0 | 1 | 4")
    _set_default(potrf_cols=cols)


def set_tuning(small_gemm_max=0, small_panel_max=0):
    """Kernel-selection thresholds (size sweeps); 0 leaves a threshold unchanged."""
    if int(small_gemm_max) > 512 or int(small_panel_max) > 512:
        raise _lib.MobocmfError("set_tuning: the small kernels serve dimensions <= 512")
    if small_gemm_max > 0:
        _set_default(small_gemm_max=small_gemm_max)
    if small_panel_max > 0:
        _set_default(small_panel_max=small_panel_max)


def gemm_f64_epilogue(A, B, C, tri, epi, alpha=1.0, stream_out=False, colsq_part=None, coldot_part=None, avec=None,
                      bscale=None, gmu=None, cgv=None, Aaux=None, rowdot_part=None):
    """The GEMM with the epilogue the layer launches it with (mobocmf_gemm_f64_epilogue): tests and per-variant timing."""
    lib = _lib.require_device()
    A, B = _prep(A), _prep(B)
    Mr, Kd = A.shape
    Nc = B.shape[1]
    snap = current_tuning()
    if epi == 1:
        need = gemm_colstat_rows(Mr, Nc, Kd, tri, tune=snap) * Nc
        for part in (colsq_part, coldot_part):
            if part is not None and part.numel() < need:
                raise _lib.MobocmfError("gemm_f64_epilogue: the partial-statistics buffers need gemm_colstat_rows() = %d rows"
                                        % (need // Nc))
    _lib.check(lib.mobocmf_gemm_f64_epilogue(tri, epi, Mr, Nc, Kd, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(C),
                                             C.stride(0), alpha, int(stream_out), _ptr(colsq_part), _ptr(coldot_part),
                                             _ptr(avec), _ptr(bscale), _ptr(gmu), _ptr(cgv), _ptr(Aaux), _ptr(rowdot_part),
                                             _ptr(getattr(_block_act_tls, "act", None)), ctypes.byref(snap), _stream()),
               "mobocmf_gemm_f64_epilogue")
    return C


def gram(kind, x1, f1, x2, f2, hyp):
    """Dense k([x1, f1], [x2, f2]) of a layer's kernel (n1 x n2), no autograd: the evaluated prior of
    MFDGPHiddenLayer.forward."""
    lib = _lib.require_device()
    with torch.no_grad():
        x1, f1, x2, f2, hyp = (_prep(None if t is None else t.detach()) for t in (x1, f1, x2, f2, hyp))
        n1, n2, d = x1.shape[0], x2.shape[0], x1.shape[1]
        if x2.shape[1] != d or hyp.numel() != hyp_len(kind, d) or not 1 <= d <= _lib.MAX_D:
            raise _lib.MobocmfError("gram: shape mismatch")
        K = _empty((n1 + 31) // 32 * 32, n2, device=x1.device)
        _lib.check(lib.mobocmf_gram_forward(kind, d, _ptr(x1), _ptr(f1), n1, _ptr(x2), _ptr(f2), n2, _ptr(hyp), _ptr(K),
                                            K.stride(0), _stream()), "mobocmf_gram_forward")
    return K[:n1]


def rff_eval(kind, x, fprev, W1, b1, Wf, W2, b2, theta, s0, s1=0.0, s2=0.0):
    """One RFF function sample of a layer at the rows of ``x`` (n, d) on the GPU (mobocmf_rff_eval); ``fprev`` (n,) = the
    previous layer's sample at the same rows (kind 1).  No autograd."""
    lib = _lib.require_device()
    with torch.no_grad():
        x, fprev, W1, b1, Wf, W2, b2, theta = (_prep(None if t is None else t.detach().reshape(t.shape))
                                               for t in (x, fprev, W1, b1, Wf, W2, b2, theta))
        n, d = x.shape
        Fn = W1.shape[0]
        if W1.shape[1] != d or b1.numel() != Fn or theta.numel() != (Fn if kind == 0 else 3 * Fn):
            raise _lib.MobocmfError("rff_eval: shape mismatch")
        if kind == 1 and (fprev.numel() != n or Wf.numel() != Fn or tuple(W2.shape) != (Fn, d) or b2.numel() != Fn):
            raise _lib.MobocmfError("rff_eval: shape mismatch (layer >= 1 operands)")
        out = _empty(n, device=x.device)
        _lib.check(lib.mobocmf_rff_eval(kind, d, Fn, n, _ptr(x), _ptr(fprev), _ptr(W1), _ptr(b1), _ptr(Wf), _ptr(W2),
                                        _ptr(b2), _ptr(theta), float(s0), float(s1), float(s2), _ptr(out), _stream()),
                   "mobocmf_rff_eval")
    return out


# ------------------------------------------------------------------------------------------------------------
# Exact-GP comparison baselines on the layer's kernels: Gram (mobocmf_gram_forward), the multi-fidelity
# combination, the blocked Cholesky + triangular inverse of the chain, the triangular MFMA product with column statistics.
# Evaluation only (no autograd): fitting the baselines' hyper-parameters differentiates the plain-torch statement.
# ------------------------------------------------------------------------------------------------------------
def mf_kernel_combine(Ks, Kn, s1, s2, l1, l2, ntab, diag=0.0):
    """K[i][j] = s1[i] s2[j] Ks[i][j] + ntab[min(l1[i], l2[j])] Kn[i][j] (+ diag on the diagonal); l1 / l2 int32 levels."""
    lib = _lib.require_device()
    with torch.no_grad():
        Ks, Kn, s1, s2, ntab = (_prep(t) for t in (Ks, Kn, s1, s2, ntab))
        n1, n2 = Ks.shape
        if tuple(Kn.shape) != (n1, n2) or l1.numel() != n1 or l2.numel() != n2 or l1.dtype != torch.int32 or l2.dtype != torch.int32:
            raise _lib.MobocmfError("mf_kernel_combine: shape / dtype mismatch")
        if int(max(l1.max(), l2.max())) >= ntab.numel() or int(min(l1.min(), l2.min())) < 0:
            raise _lib.MobocmfError("mf_kernel_combine: fidelity level outside the noise-factor table")
        l1, l2 = l1.contiguous(), l2.contiguous()
        out = _empty(n1, n2, device=Ks.device)
        _lib.check(lib.mobocmf_mf_kernel_combine(n1, n2, _ptr(Ks), _ptr(Kn), Ks.stride(0), _ptr(s1), _ptr(s2), _ptr(l1), _ptr(l2),
                                                 _ptr(ntab), float(diag), _ptr(out), n2, n1, n2, _stream()),
                   "mobocmf_mf_kernel_combine")
    return out


class ExactGPState:
    """L, L^-1 and a = L^-1 y of an exact GP on n training points (mobocmf_exact_gp_factor)."""
    __slots__ = ("n", "state", "mll", "info")


def exact_gp_factor(K, y):
    """Factorises the n x n training covariance (noise on its diagonal); returns the state with ``.mll`` = log N(y | 0, K)
    and ``.info`` (0 or the failed pivot, device word)."""
    lib = _lib.require_device()
    with torch.no_grad():
        K, y = _prep(K), _prep(y.reshape(-1))
        n = K.shape[0]
        if K.shape[1] != n or y.numel() != n:
            raise _lib.MobocmfError("exact_gp_factor: K must be n x n and y of length n")
        sb, cb = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(lib.mobocmf_exact_gp_workspace_bytes(n, 1, ctypes.byref(sb), ctypes.byref(cb)), "mobocmf_exact_gp_workspace_bytes")
        st = ExactGPState()
        st.n = n
        st.state = _poison(torch.empty(sb.value, dtype=torch.uint8, device=K.device))
        st.mll = _empty((), device=K.device)
        st.info = torch.zeros((), dtype=torch.int32, device=K.device)
        scratch = scratch_buffer(cb.value, K.device)
        _lib.check(lib.mobocmf_exact_gp_factor(n, _ptr(K), K.stride(0), _ptr(y), _ptr(st.mll), _ptr(st.info), _ptr(st.state),
                                               sb.value, _ptr(scratch), scratch.numel(), ctypes.byref(current_tuning()), _stream()),
                   "mobocmf_exact_gp_factor")
    return st


def _exact_state_views(st):
    """(L, L^-1, a = L^-1 y) of an ExactGPState as padded views (npad x npad, npad)."""
    npad = (st.n + 127) // 128 * 128
    buf = st.state.view(torch.float64)
    step = (npad * npad * 8 + 255) // 256 * 256 // 8
    vstep = (npad * 8 + 255) // 256 * 256 // 8
    L = buf[:npad * npad].view(npad, npad)
    Li = buf[step:step + npad * npad].view(npad, npad)
    a = buf[2 * step:2 * step + npad]
    del vstep
    return L, Li, a, npad


class _ExactGPMLL(torch.autograd.Function):
    """log N(y | 0, K) with its gradient, both on the library's kernels: forward = mobocmf_exact_gp_factor (blocked Cholesky,
    triangular inverse, a = L^-1 y, the likelihood), backward = d/dK = (alpha alpha^T - K^-1) / 2 with alpha = L^-T a and
    K^-1 = L^-T L^-1 as ONE triangular product on the f64 MFMA kernel, d/dy = -alpha.  What gpytorch's
    ExactMarginalLogLikelihood (x n) differentiates through torch.linalg (mfgp.py:63-64 of the reference fits through it)."""

    @staticmethod
    def forward(ctx, K, y):
        st = exact_gp_factor(K, y)
        ctx.st = st
        ctx.y_shape = y.shape
        return st.mll.clone()

    @staticmethod
    def backward(ctx, g):
        st = ctx.st
        n = st.n
        with torch.no_grad():
            _, Li, a, npad = _exact_state_views(st)
            LiT = Li.t().contiguous()
            Kinv = gemm_f64(LiT, LiT, trans_b=True, tri=2)          # L^-T (L^-T)^T, A upper triangular
            alpha = LiT @ a
            gK = (0.5 * g) * (torch.outer(alpha[:n], alpha[:n]) - Kinv[:n, :n])
            gy = (-g) * alpha[:n]
        return gK, gy.reshape(ctx.y_shape)


def exact_gp_mll(K, y):
    """Differentiable log N(y | 0, K) on the library's kernels (K: n x n incl. the noise on its diagonal).  The caller checks
    ``exact_gp_mll.last_info`` style failures through the returned value: a failed pivot makes the value NaN."""
    return _ExactGPMLL.apply(K, y)


def exact_gp_predict(st, Kts, kss):
    """Posterior mean and variance at the nt columns of Kts [n x nt] (prior variances kss)."""
    lib = _lib.require_device()
    with torch.no_grad():
        Kts, kss = _prep(Kts), _prep(kss.reshape(-1))
        n, nt = Kts.shape
        if n != st.n or kss.numel() != nt:
            raise _lib.MobocmfError("exact_gp_predict: shape mismatch")
        sb, cb = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(lib.mobocmf_exact_gp_workspace_bytes(n, nt, ctypes.byref(sb), ctypes.byref(cb)), "mobocmf_exact_gp_workspace_bytes")
        scratch = scratch_buffer(cb.value, Kts.device)
        mean, var = _empty(nt, device=Kts.device), _empty(nt, device=Kts.device)
        _lib.check(lib.mobocmf_exact_gp_predict(n, nt, _ptr(Kts), Kts.stride(0), _ptr(kss), _ptr(mean), _ptr(var), _ptr(st.state),
                                                st.state.numel(), _ptr(scratch), scratch.numel(), ctypes.byref(current_tuning()),
                                                _stream()), "mobocmf_exact_gp_predict")
    return mean, var


class _SoftplusPackFn(torch.autograd.Function):
    """softplus of several raw parameter tensors, concatenated: one launch forward, one backward (mobocmf_softplus_pack)."""

    @staticmethod
    def forward(ctx, *raws):
        lib = _lib.require_device()
        flat = [_prep(r.detach().reshape(-1)) for r in raws]
        n = len(flat)
        sizes = (ctypes.c_int32 * n)(*[f.numel() for f in flat])
        ptrs = (ctypes.c_void_p * n)(*[f.data_ptr() for f in flat])
        out = _empty(sum(f.numel() for f in flat), device=flat[0].device)
        _lib.check(lib.mobocmf_softplus_pack(n, ptrs, sizes, _ptr(out), _stream()), "mobocmf_softplus_pack")
        ctx.save_for_backward(*raws)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.require_device()
        raws = ctx.saved_tensors
        flat = [_prep(r.detach().reshape(-1)) for r in raws]
        n = len(flat)
        grads = [_empty_like(r) for r in raws]
        sizes = (ctypes.c_int32 * n)(*[f.numel() for f in flat])
        ptrs = (ctypes.c_void_p * n)(*[f.data_ptr() for f in flat])
        gptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in grads])
        _lib.check(lib.mobocmf_softplus_pack_backward(n, ptrs, sizes, _ptr(_prep(g)), gptrs, _stream()),
                   "mobocmf_softplus_pack_backward")
        return tuple(grads)


class _SoftplusPackSegFn(torch.autograd.Function):
    """``softplus_pack`` of the raw tensors of SEVERAL consumers (the layers of a model) in one launch: the packed vector is
    handed out as one tensor per consumer (``seg`` = raw tensors per consumer); the backward collects the consumers' gradients
    through a per-tensor pointer table -- one launch, no concatenation of gradients, no slice-backward launches."""

    @staticmethod
    def forward(ctx, seg, *raws):
        lib = _lib.require_device()
        flat = [_prep(r.detach().reshape(-1)) for r in raws]
        n = len(flat)
        sizes = (ctypes.c_int32 * n)(*[f.numel() for f in flat])
        ptrs = (ctypes.c_void_p * n)(*[f.data_ptr() for f in flat])
        out = _empty(sum(f.numel() for f in flat), device=flat[0].device)
        _lib.check(lib.mobocmf_softplus_pack(n, ptrs, sizes, _ptr(out), _stream()), "mobocmf_softplus_pack")
        ctx.save_for_backward(*raws)
        ctx.seg = tuple(seg)
        ctx.set_materialize_grads(False)
        outs, off, k = [], 0, 0
        for cnt in seg:
            ln = sum(f.numel() for f in flat[k:k + cnt])
            outs.append(out[off:off + ln])
            off, k = off + ln, k + cnt
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        lib = _lib.require_device()
        raws = ctx.saved_tensors
        flat = [_prep(r.detach().reshape(-1)) for r in raws]
        n = len(flat)
        grads = [_empty_like(r) for r in raws]
        sizes = (ctypes.c_int32 * n)(*[f.numel() for f in flat])
        ptrs = (ctypes.c_void_p * n)(*[f.data_ptr() for f in flat])
        gptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in grads])
        src, k, keep = [], 0, []
        for cnt, g in zip(ctx.seg, gs):
            g = None if g is None else _prep(g)
            keep.append(g)
            off = 0
            for f in flat[k:k + cnt]:
                src.append(0 if g is None else g.data_ptr() + 8 * off)
                off += f.numel()
            k += cnt
        _lib.check(lib.mobocmf_softplus_pack_backward_v(n, ptrs, sizes, (ctypes.c_void_p * n)(*src), gptrs, _stream()),
                   "mobocmf_softplus_pack_backward_v")
        return (None,) + tuple(grads)


def softplus_pack_segments(raw_groups):
    """One launch for several groups of raw tensors (<= 16 tensors in all): returns one packed, differentiable vector per group."""
    seg = [len(g) for g in raw_groups]
    return list(_SoftplusPackSegFn.apply(seg, *[r for g in raw_groups for r in g]))


def softplus_pack(raws):
    """``softplus(cat(raws))`` for float64 device tensors (<= 16 of them) in one launch, differentiable."""
    return _SoftplusPackFn.apply(*raws)


def syrk_weighted(A, w, H=None):
    """H = A diag(w) A^T (full symmetric, float64) on the k-sliced MFMA kernel: the weighted syrk of the layer backward
    (mobocmf_syrk_weighted_f64).  A: [Mr, Kd] with Mr, Kd multiples of 128."""
    lib = _lib.require_device()
    A, w = _prep(A), _prep(w)
    Mr, Kd = A.shape
    nb = _lib._SZ()
    snap = current_tuning()      # the size query and the launch: the same values
    _lib.check(lib.mobocmf_syrk_workspace_bytes(Mr, Kd, ctypes.byref(snap), ctypes.byref(nb)), "mobocmf_syrk_workspace_bytes")
    ws = scratch_buffer(nb.value, A.device)
    if H is None:
        H = _empty(Mr, Mr, device=A.device)
    _lib.check(lib.mobocmf_syrk_weighted_f64(Mr, Kd, _ptr(A), A.stride(0), _ptr(w), _ptr(H), _ptr(ws), nb.value,
                                             _ptr(getattr(_block_act_tls, "act", None)), ctypes.byref(snap), _stream()),
               "mobocmf_syrk_weighted_f64")
    return H


def gemm_f64(A, B, C=None, tri=0, trans_b=False, alpha=1.0, accumulate=False):
    """C (+)= alpha * A @ (B.T if trans_b else B) on the f64 MFMA kernel (tile-aligned shapes only)."""
    lib = _lib.require_device()
    A, B = _prep(A), _prep(B)
    Mr, Kd = A.shape
    Nc = B.shape[0] if trans_b else B.shape[1]
    if C is None:
        C = _empty(Mr, Nc, device=A.device)
    _lib.check(lib.mobocmf_gemm_f64(tri, int(trans_b), Mr, Nc, Kd, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(C),
                                    C.stride(0), alpha, int(accumulate), ctypes.byref(current_tuning()), _stream()),
               "mobocmf_gemm_f64")
    return C
