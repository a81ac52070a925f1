"""ctypes binding of the C-ABI in include/mobocmf_hip.h (libmobocmf_hip.so, built by csrc/build.sh).

The product path has NO CPU fallback: if the shared library is missing or the device is not a
gfx950, every compute entry point raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOBOCMF_HIP_LIB") or os.path.join(_HERE, "csrc", "libmobocmf_hip.so")   # override: A/B builds

OK, BAD_ARG, WORKSPACE_TOO_SMALL, HIP_ERROR, NOT_PD, BAD_ARCH = range(6)
_ERR = {1: "MOBOCMF_BAD_ARG", 2: "MOBOCMF_WORKSPACE_TOO_SMALL", 3: "MOBOCMF_HIP_ERROR", 4: "MOBOCMF_NOT_PD",
        5: "MOBOCMF_BAD_ARCH"}


class Tuning(ctypes.Structure):
    """mobocmf_tuning of include/mobocmf_hip.h: the kernel-selection knobs that travel with a call (NULL = defaults)."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("small_gemm_max", ctypes.c_int32), ("small_panel_max", ctypes.c_int32),
                ("tile_rows", ctypes.c_int32), ("pair_mode", ctypes.c_int32), ("mid_gemm_max", ctypes.c_int32),
                ("mid_gemm_waves", ctypes.c_int32), ("syrk_workgroups", ctypes.c_int32), ("sparse_backward", ctypes.c_int32),
                ("potrf_cols", ctypes.c_int32)]
    KNOBS = tuple(n for n, _ in _fields_[1:])

    def copy(self):
        t = Tuning()
        ctypes.memmove(ctypes.byref(t), ctypes.byref(self), ctypes.sizeof(Tuning))
        return t


PROBE_EVENTS = 11      # MOBOCMF_PROBE_EVENTS


class LayerDesc(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("d", ctypes.c_int32), ("M", ctypes.c_int32), ("xdiv", ctypes.c_int32),
                ("Np", ctypes.c_int64), ("branch", ctypes.c_int32), ("want_dx", ctypes.c_int32),
                ("jitter", ctypes.c_double), ("min_var", ctypes.c_double), ("phase", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("tuning", ctypes.POINTER(Tuning)), ("probe_events", ctypes.c_void_p)]


TINY_MAX_LAYERS, TINY_MAX_M, TINY_MAX_D = 3, 32, 8      # MOBOCMF_TINY_MAX_* of include/mobocmf_hip.h
COOP_MAX_M = 128                                         # MOBOCMF_COOP_MAX_M
STEP_CHAIN_VALID = 16                                    # MOBOCMF_STEP_CHAIN_VALID


class TinyModel(ctypes.Structure):
    """mobocmf_tiny_model: one small surrogate of mobocmf_tiny_elbo_step (the kernel reads the array from DEVICE memory)."""
    _L = TINY_MAX_LAYERS
    _fields_ = [("L", ctypes.c_int32), ("M", ctypes.c_int32), ("d", ctypes.c_int32), ("S", ctypes.c_int32),
                ("N", ctypes.c_int32), ("rows", ctypes.c_int32 * _L), ("trainable", ctypes.c_uint32 * _L),
                ("branch", ctypes.c_int32),
                ("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("fid", ctypes.c_void_p), ("Zx", ctypes.c_void_p),
                ("raw", (ctypes.c_void_p * 7) * _L), ("m", ctypes.c_void_p * _L), ("L_S", ctypes.c_void_p * _L),
                ("raw_noise", ctypes.c_void_p * _L), ("noise_lo", ctypes.c_double * _L), ("noise_hi", ctypes.c_double * _L),
                ("rng", ctypes.c_void_p * _L), ("eps", ctypes.c_void_p * _L),
                ("adam_m", ctypes.c_void_p), ("adam_v", ctypes.c_void_p), ("steps_done", ctypes.c_void_p),
                ("work", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("out", ctypes.c_void_p), ("info", ctypes.c_void_p),
                ("kl_scale", ctypes.c_double), ("jitter", ctypes.c_double),
                ("row_weight", ctypes.c_void_p), ("seed_gmean", ctypes.c_void_p), ("seed_gvar", ctypes.c_void_p),
                ("seed_scale", ctypes.c_double), ("top_mean", ctypes.c_void_p), ("top_var", ctypes.c_void_p),
                ("xrng", ctypes.c_void_p), ("rand_row0", ctypes.c_int32), ("rand_rows", ctypes.c_int32),
                ("coupling", ctypes.c_void_p), ("role", ctypes.c_int32), ("role_index", ctypes.c_int32)]


class TinyCoupling(ctypes.Structure):
    """mobocmf_tiny_coupling: the theta / omega factors over the models of one mode-4 launch (device-resident)."""
    _fields_ = [("n_obj", ctypes.c_int32), ("n_con", ctypes.c_int32), ("P", ctypes.c_int32), ("T", ctypes.c_int32),
                ("obj_model", ctypes.c_int32 * 8), ("con_model", ctypes.c_int32 * 8),
                ("front", ctypes.c_void_p), ("thresholds", ctypes.c_void_p),
                ("log_eps", ctypes.c_double), ("log_1m_eps", ctypes.c_double), ("losses", ctypes.c_void_p),
                ("barrier", ctypes.c_void_p), ("status", ctypes.c_void_p), ("n_models", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class MobocmfError(RuntimeError):
    pass


_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_I32 = ctypes.c_int32
_D = ctypes.c_double
_SZ = ctypes.c_size_t

# name -> argtypes   (every symbol include/mobocmf_hip.h declares)
SYMBOLS = {
    "mobocmf_version": [],
    "mobocmf_device_arch_ok": [],
    "mobocmf_layer_workspace_bytes": [ctypes.POINTER(LayerDesc), ctypes.POINTER(_SZ), ctypes.POINTER(_SZ)],
    "mobocmf_layer_chain_state_bytes": [ctypes.POINTER(LayerDesc), ctypes.POINTER(_SZ)],
    "mobocmf_layer_forward": [ctypes.POINTER(LayerDesc)] + [_P] * 11 + [_P, _SZ, _P, _SZ, _P],
    "mobocmf_layer_backward": [ctypes.POINTER(LayerDesc)] + [_P] * 16 + [_P, _SZ, _P, _SZ, _P],
    "mobocmf_chain_block_bytes": [ctypes.POINTER(LayerDesc), ctypes.POINTER(_SZ), ctypes.POINTER(_SZ)],
    "mobocmf_panel_workspace_bytes": [ctypes.POINTER(LayerDesc), ctypes.POINTER(_SZ), ctypes.POINTER(_SZ)],
    "mobocmf_layers_chain_forward": [_I32] + [_P] * 8 + [_P, _SZ, _SZ, _P],
    "mobocmf_layers_chain_backward": [_I32] + [_P] * 10 + [_P, _SZ, _SZ, _P],
    "mobocmf_layer_panel_forward": [ctypes.POINTER(LayerDesc)] + [_P] * 7 + [_P, _SZ, _P, _SZ, _P, _SZ, _P],
    "mobocmf_layer_panel_backward": [ctypes.POINTER(LayerDesc)] + [_P] * 11 + [_P, _SZ, _P, _SZ, _P, _SZ, _P],
    "mobocmf_predictive_covariance_workspace_bytes": [ctypes.POINTER(LayerDesc), ctypes.POINTER(_SZ)],
    "mobocmf_predictive_covariance": [ctypes.POINTER(LayerDesc), _P, _P, _P, _P, _P, _P, _I64, _P, _SZ, _P, _SZ, _P],
    "mobocmf_propagate_forward": [_P, _P, _P, _P, _I64, _I32, _P],
    "mobocmf_propagate_rng_forward": [_P, _P, _P, _P, _P, _I64, _I32, _P],
    "mobocmf_propagate_backward": [_P, _P, _P, _P, _P, _I64, _I32, _P],
    "mobocmf_propagate_backward_prefix": [_P, _P, _P, _P, _P, _I64, _I32, _I64, _P, _P, _P],
    "mobocmf_elbo_data_forward": [_P, _P, _P, _P, _P, _D, _I64, _I32, _P, _P, _SZ, _P],
    "mobocmf_elbo_data_backward": [_P, _P, _P, _P, _P, _D, _I64, _I32, _P, _P, _P, _P, _P, _SZ, _P],
    "mobocmf_elbo_data_interval_forward": [_P, _P, _P, _P, _P, _D, _D, _D, _I64, _I32, _P, _P, _SZ, _P],
    "mobocmf_elbo_data_interval_backward": [_P, _P, _P, _P, _P, _D, _D, _D, _I64, _I32, _P, _P, _P, _P, _P, _SZ, _P],
    "mobocmf_elbo_forward": [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I32, _P, _D, _P, _P, _SZ, _P],
    "mobocmf_elbo_backward": [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _D, _P, _P, _P, _P, _P, _P, _P, _SZ, _P],
    "mobocmf_acq_moments_forward": [_P, _P, _P, _P, _I64, _I32, _P],
    "mobocmf_acq_moments_backward": [_P, _P, _P, _P, _P, _I64, _I32, _P],
    "mobocmf_jes_forward": [_P, _P, _P, _I64, _P],
    "mobocmf_shortcut_var_forward": [_P, _I32, _D, _P, _P],
    "mobocmf_shortcut_var_backward": [_P, _P, _P, _I32, _D, _P, _P],
    "mobocmf_elbo_combine_forward": [_I32, _P, _I32, _P, _D, _P, _P],
    "mobocmf_elbo_combine_backward": [_P, _P, _D, _P, _P],
    "mobocmf_adam_multi": [_I32, _P, _P, _P, _P, _P, _D, _D, _D, _D, _P, _P],
    "mobocmf_adam_step": [_P, _P, _P, _P, _P, _I64, _D, _D, _D, _D, _I64, _P],
    "mobocmf_tuning_init": [ctypes.POINTER(Tuning)],
    "mobocmf_gemm_f64": [_I32, _I32, _I32, _I64, _I64, _P, _I64, _P, _I64, _P, _I64, _D, _I32, ctypes.POINTER(Tuning), _P],
    "mobocmf_gemm_f64_epilogue": [_I32, _I32, _I32, _I64, _I64, _P, _I64, _P, _I64, _P, _I64, _D, _I32] + [_P] * 8 +
                                 [_P, ctypes.POINTER(Tuning), _P],
    "mobocmf_mf_kernel_combine": [_I64, _I64, _P, _P, _I64, _P, _P, _P, _P, _P, _D, _P, _I64, _I64, _I64, _P],
    "mobocmf_exact_gp_workspace_bytes": [_I32, _I64, ctypes.POINTER(_SZ), ctypes.POINTER(_SZ)],
    "mobocmf_exact_gp_factor": [_I32, _P, _I64, _P, _P, _P, _P, _SZ, _P, _SZ, ctypes.POINTER(Tuning), _P],
    "mobocmf_exact_gp_predict": [_I32, _I64, _P, _I64, _P, _P, _P, _P, _SZ, _P, _SZ, ctypes.POINTER(Tuning), _P],
    "mobocmf_gemm_colstat_rows": [_I32, _I32, _I64, _I64, ctypes.POINTER(Tuning), ctypes.POINTER(_I32)],
    "mobocmf_syrk_workspace_bytes": [_I32, _I64, ctypes.POINTER(Tuning), ctypes.POINTER(_SZ)],
    "mobocmf_syrk_weighted_f64": [_I32, _I64, _P, _I64, _P, _P, _P, _I64, _P, ctypes.POINTER(Tuning), _P],
    "mobocmf_softplus_pack": [_I32, _P, _P, _P, _P],
    "mobocmf_softplus_pack_backward": [_I32, _P, _P, _P, _P, _P],
    "mobocmf_softplus_pack_backward_v": [_I32, _P, _P, _P, _P, _P],
    "mobocmf_cond_factors_forward": [_I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _D, _D, _P, _P, _P, _P, _P, _P],
    "mobocmf_scale_segments": [_I32, _P, _P, _P, _P, _P, _P],
    "mobocmf_gather_segments": [_I32, _P, _P, _P, _P],
    "mobocmf_scalar_combine": [_I32, _P, _P, _P, _P],
    "mobocmf_tiny_flat_len": [ctypes.POINTER(TinyModel), ctypes.POINTER(_I64)],
    "mobocmf_tiny_work_bytes": [ctypes.POINTER(TinyModel), ctypes.POINTER(_SZ)],
    "mobocmf_tiny_elbo_step": [_P, _P, _I32, _D, _D, _D, _D, _I32, _P],
    "mobocmf_coop_work_bytes": [ctypes.POINTER(TinyModel), ctypes.POINTER(_SZ)],
    "mobocmf_coop_elbo_step": [_P, _P, _I32, _I32, _P, _D, _D, _D, _D, _I32, ctypes.POINTER(_I32), _P],
    "mobocmf_rff_eval": [_I32, _I32, _I32, _I64] + [_P] * 8 + [_D, _D, _D, _P, _P],
    "mobocmf_gram_forward": [_I32, _I32, _P, _P, _I64, _P, _P, _I64, _P, _P, _I64, _P],
    "mobocmf_check_info": [_P, ctypes.POINTER(_I32), _P],
}
MAX_D, MAX_XDIV = 32, 48        # MOBOCMF_MAX_D / MOBOCMF_MAX_XDIV of include/mobocmf_hip.h

_lib = None


def load():
    """Loads the shared library (no GPU needed for loading / symbol checks)."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  -- first: the HIP runtime torch ships must be the one this library binds to
        if not os.path.exists(LIB_PATH):
            raise MobocmfError(f"HIP extension missing: {LIB_PATH} (run mobocmf_amd/csrc/build.sh or "
                               "__graft_entry__.build()); there is no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        _lib = lib
    return _lib


def check(rc, what):
    if rc != OK:
        raise MobocmfError(f"{what} failed: {_ERR.get(rc, rc)}")


_arch_checked = False


def require_device():
    """Fail loudly unless a gfx950 device is current."""
    global _arch_checked
    lib = load()
    if not _arch_checked:
        import torch
        if not torch.cuda.is_available():
            raise MobocmfError("mobocmf_amd needs an MI355X (gfx950) device: no GPU visible, and there is no CPU fallback")
        if not lib.mobocmf_device_arch_ok():
            raise MobocmfError("mobocmf_amd kernels are built for gfx950 only")
        _arch_checked = True
    return lib
