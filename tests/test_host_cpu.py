"""CPU-only checks: the C-ABI library loads and exports every symbol of include/mobocmf_hip.h, the host mirror of
the reference surface behaves (construction, init heuristics, parameter tree, freezing, deepcopy/dill), and the
product path fails loudly without a GPU (no CPU fallback)."""
import copy
import io
import os
import re

import numpy as np
import pytest
import torch

from mobocmf_amd.util import synthetic
from oracle import mfdgp_oracle as O
from tests.helpers import to_t

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mobocmf_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mobocmf_hip.h")).read()
    declared = set(re.findall(r"^int (mobocmf_\w+)\(", hdr, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mobocmf_version() >= 100           # host-only call, no GPU needed


def test_workspace_query_and_bad_args():
    from mobocmf_amd import functional as F
    saved, scratch = F.workspace_bytes(F.make_desc(1, 8, 512, 65536, 8))
    # kept for backward: A, C (two M x N' panels) + the M x M chain state; K_mn itself is forward scratch
    assert 2 * 512 * 65536 * 8 < saved < 2.2 * 512 * 65536 * 8 and scratch > 2 * 512 * 65536 * 8
    from mobocmf_amd import _lib
    with pytest.raises(_lib.MobocmfError):
        F.workspace_bytes(F.make_desc(1, 40, 512, 65536, 8))      # d > 32
    with pytest.raises(_lib.MobocmfError):
        F.workspace_bytes(F.make_desc(0, 2, 8, 13, 2))            # Np % xdiv != 0


def _forrester_model(**kw):
    from mobocmf_amd.models import MFDGP
    x, y, fid = synthetic.forrester_problem(0)
    return MFDGP(to_t(x), to_t(y)[:, None], to_t(fid)[:, None], 2, **kw), (x, y, fid)


def test_tuning_travels_with_the_call_and_the_library_keeps_no_knobs():
    """no global mutable state.  Every knob is a field of mobocmf_tuning, passed per call (descriptor pointer /
    argument); the size queries follow the record they are given; a malformed record is refused; no mobocmf_set_* entry
    point exists any more."""
    import ctypes
    import subprocess
    from mobocmf_amd import _lib
    from mobocmf_amd import functional as F
    lib = _lib.load()
    exported = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "mobocmf_set_" not in exported
    t = _lib.Tuning()
    assert lib.mobocmf_tuning_init(ctypes.byref(t)) == _lib.OK and lib.mobocmf_tuning_init(None) == _lib.BAD_ARG
    assert t.struct_size == ctypes.sizeof(_lib.Tuning)
    assert [getattr(t, k) for k in _lib.Tuning.KNOBS] == [384, 512, 0, 0, 1024, 32, 0, 1, 0]
    nb_def, nb_64, nb_4096 = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.mobocmf_syrk_workspace_bytes(512, 65536, None, ctypes.byref(nb_def)) == _lib.OK
    t64, t4096 = t.copy(), t.copy()
    t64.syrk_workgroups, t4096.syrk_workgroups = 64, 4096
    assert lib.mobocmf_syrk_workspace_bytes(512, 65536, ctypes.byref(t64), ctypes.byref(nb_64)) == _lib.OK
    assert lib.mobocmf_syrk_workspace_bytes(512, 65536, ctypes.byref(t4096), ctypes.byref(nb_4096)) == _lib.OK
    assert nb_64.value < nb_def.value < nb_4096.value          # fewer workgroups = fewer slabs
    # the layer workspaces follow the descriptor's record, and two descriptors do not influence each other
    with F.tuning(syrk_workgroups=64):
        d_small = F.make_desc(1, 8, 512, 65536, xdiv=8)
    d_def = F.make_desc(1, 8, 512, 65536, xdiv=8)
    assert F.workspace_bytes(d_small)[1] < F.workspace_bytes(d_def)[1]
    assert F.workspace_bytes(d_def) == F.workspace_bytes(F.make_desc(1, 8, 512, 65536, xdiv=8))
    rows = ctypes.c_int32()
    for tr, want in ((64, 16), (128, 8)):
        tt = t.copy()
        tt.tile_rows = tr
        assert lib.mobocmf_gemm_colstat_rows(1, 512, 65536, 512, ctypes.byref(tt), ctypes.byref(rows)) == _lib.OK
        assert rows.value == want
    # malformed records
    for field, bad in (("struct_size", 8), ("tile_rows", 96), ("pair_mode", 3), ("mid_gemm_waves", 16), ("syrk_workgroups", 5),
                       ("potrf_cols", 3), ("small_gemm_max", 513), ("sparse_backward", 2)):
        tb = t.copy()
        setattr(tb, field, bad)
        assert lib.mobocmf_syrk_workspace_bytes(512, 65536, ctypes.byref(tb), ctypes.byref(nb_def)) == _lib.BAD_ARG, field
        d_bad = F.make_desc(0, 2, 8, 12, tune=tb)
        a, b = ctypes.c_size_t(), ctypes.c_size_t()
        assert lib.mobocmf_layer_workspace_bytes(ctypes.byref(d_bad), ctypes.byref(a), ctypes.byref(b)) == _lib.BAD_ARG, field
    # the host-side default record is this module's, per-thread overrides do not leak
    import threading
    seen = {}

    def worker():
        with F.tuning(tile_rows=64):
            seen["inner"] = F.current_tuning().tile_rows
    with F.tuning(tile_rows=128):
        th = threading.Thread(target=worker)
        th.start()
        th.join()
        seen["outer"] = F.current_tuning().tile_rows
    assert seen == {"inner": 64, "outer": 128} and F.current_tuning().tile_rows == 0


def test_model_surface_and_init_heuristics_match_reference_rules():
    model, (x, y, fid) = _forrester_model()
    model.double()
    assert model.num_hidden_layers == 2 and model.name_hidden_layer == "hidden_layer_"
    l0, l1 = model.hidden_layer_0, model.hidden_layer_1
    xt, yt, ft = to_t(x), to_t(y), to_t(fid)
    # lengthscale: the as-written MEDIAN rule, float32-rounded like the reference (B.4)
    ls_lo = float(O.median_lengthscale(xt[ft == 0]))
    ls_hi = float(O.median_lengthscale(xt[ft == 1]))
    assert float(l0.covar_module.base_kernel.lengthscale) == pytest.approx(ls_lo, rel=1e-6)
    k1 = l1.covar_module.kernels[0].kernels[0]
    k2 = l1.covar_module.kernels[1]
    kf = l1.covar_module.kernels[0].kernels[1].kernels[1]
    klin = l1.covar_module.kernels[0].kernels[1].kernels[0]
    assert float(k1.base_kernel.lengthscale) == pytest.approx(10 * ls_hi, rel=1e-6)
    assert float(k2.base_kernel.lengthscale) == pytest.approx(ls_hi, rel=1e-6)
    assert float(kf.base_kernel.lengthscale) == pytest.approx(1.0, rel=1e-6)
    assert float(k1.outputscale) == pytest.approx(1.0, rel=1e-6) and float(k2.outputscale) == pytest.approx(0.01, rel=1e-5)
    assert float(klin.variance) == pytest.approx(1.0, rel=1e-6)
    # inducing inputs = all training inputs; q(u) mean = nearest same-fidelity target (mfdgp.py:290-317)
    assert torch.equal(l0.variational_strategy.inducing_points, xt)
    m0 = O.nearest_same_fidelity_values(xt, yt, ft, xt, 0)
    vd0 = l0.variational_strategy._variational_distribution
    assert torch.allclose(vd0.variational_mean, m0, rtol=1e-6)
    assert torch.allclose(torch.diagonal(vd0.chol_variational_covar), torch.full((16,), 1e-4, dtype=torch.float64), rtol=1e-5)
    # layer 1 inducing inputs recomputed from layer 0's variational mean, differentiable
    Z1 = l1.variational_strategy.inducing_points
    assert Z1.shape == (16, 2) and torch.equal(Z1[:, 1], vd0.variational_mean)
    assert Z1.requires_grad
    # likelihood bounds and initial noise (mfdgp.py:113-121)
    lik1 = model.hidden_layer_likelihood_1
    y_high_std = np.std(y[fid == 1])
    assert float(lik1.noise) == pytest.approx(1e-2 * y_high_std, rel=1e-5)
    assert lik1.raw_noise_constraint.upper_bound == pytest.approx(0.1 * y_high_std)
    assert float(model.hidden_layer_likelihood_0.noise) == pytest.approx(1e-6, rel=1e-3)
    assert l1.samples.shape == (25, 1)
    x0, x1 = torch.tensor([[0.0], [0.9], [0.4]], dtype=torch.float64), torch.tensor([[1.0], [0.5], [0.1]], dtype=torch.float64)
    assert model.clip_inducing_values(x0, x1, torch.tensor([10.0, 20.0, 30.0])).tolist() == [30.0, 10.0, 20.0]


def test_fix_variational_hypers_toggles_what_the_reference_toggles():
    model, _ = _forrester_model()
    model.fix_variational_hypers(True)
    for i in range(2):
        assert not getattr(model, f"hidden_layer_likelihood_{i}").raw_noise.requires_grad
        layer = getattr(model, f"hidden_layer_{i}")
        assert not layer.variational_strategy._variational_distribution.chol_variational_covar.requires_grad
        assert layer.variational_strategy._variational_distribution.variational_mean.requires_grad
        assert all(p.requires_grad for p in layer.covar_module.parameters())
    model.fix_variational_hypers(False)
    model.fix_variational_hypers_cond(True)
    assert not any(p.requires_grad for p in model.hidden_layer_1.covar_module.parameters())
    # parameters() de-duplicates the previous layer registered under layer 1's strategy
    names = [n for n, _ in model.named_parameters()]
    assert len(names) == len(set(names)) == 15


def test_deepcopy_and_dill_roundtrip_on_cpu():
    import dill
    model, _ = _forrester_model()
    m2 = copy.deepcopy(model)
    assert m2.hidden_layer_1.variational_strategy.previous_layer is m2.hidden_layer_0
    buf = io.BytesIO()
    dill.dump(model, buf)
    m3 = dill.loads(buf.getvalue())
    for (n, p), (_, q) in zip(model.named_parameters(), m3.named_parameters()):
        assert torch.equal(p, q), n


def test_only_highest_fidelity_ablation_and_extensions():
    model, _ = _forrester_model(use_only_highest_fidelity=True)
    cm = model.hidden_layer_1.covar_module
    assert float(cm.kernels[0].kernels[0].outputscale) == 0.0 and float(cm.kernels[1].outputscale) == pytest.approx(1.0)
    assert not cm.kernels[0].kernels[0].raw_outputscale.requires_grad
    assert model.hidden_layer_1.variational_strategy.inducing_points.shape[0] == 4     # high-fidelity points only
    prob = synthetic.make_problem(d=3, L=3, M=10, N=16, S=2, seed=7)
    m3 = synthetic.model_from_problem(prob, device="cpu")
    assert m3.num_hidden_layers == 3 and m3.hidden_layer_2.variational_strategy.Zx.shape == (10, 3)


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mobocmf_amd import _lib
    model, (x, _, _) = _forrester_model()
    model.double()
    with pytest.raises(_lib.MobocmfError):
        model(to_t(x))


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mobocmf_amd")):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in re.sub(r"#.*", "", src).replace("cpu-oracle", ""), fn


def test_synthetic_problems_are_deterministic():
    a = synthetic.make_problem(d=8, L=2, M=16, N=64, S=2, output=1, seed=3)
    b = synthetic.make_problem(d=8, L=2, M=16, N=64, S=2, output=1, seed=3)
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["eps"][1], b["eps"][1])
    assert (a["fid"][:16] == 1).all() and (a["fid"][16:] == 0).all()


def test_rff_posterior_function_samples():
    """weight-space samples f ~ q(f): at the inducing inputs their mean is the variational mean and
    their spread is diag(S) (+ RFF / jitter error); a sample's gradient matches finite differences; layer >= 1
    recurses through the previous layer's sample."""
    prob = synthetic.make_problem(d=2, L=2, M=8, N=12, S=2, seed=3)
    for lay in prob["layers"]:
        lay["hyp"] = {k: (v * 0 + 0.6 if k.startswith("ls") and k != "lsf" else v) for k, v in lay["hyp"].items()}
    model = synthetic.model_from_problem(prob, device="cpu")
    Z = prob["Zx"]
    g = torch.Generator().manual_seed(0)
    vals = np.stack([model.sample_function_from_each_layer(nFeatures=400, generator=g)[0](Z) for _ in range(300)])
    m0 = prob["layers"][0]["m"]
    S0 = np.tril(prob["layers"][0]["L_S"]) @ np.tril(prob["layers"][0]["L_S"]).T
    assert np.abs(vals.mean(0) - m0).max() < 4 * np.sqrt(np.diag(S0).max() / 300) + 0.02
    assert np.abs(vals.std(0) - np.sqrt(np.diag(S0))).max() < 0.03
    fs = model.sample_function_from_each_layer(nFeatures=200, generator=g)
    x = np.array([0.3, 0.6])
    for f in fs:
        assert f(x).shape == (1,) and f(np.stack([x, x + 0.1])).shape == (2,)
        gr = f(x, gradient=True)
        fd = np.array([(f(x + 1e-6 * e)[0] - f(x - 1e-6 * e)[0]) / 2e-6 for e in np.eye(2)])
        assert np.abs(gr - fd).max() < 1e-5 * max(1.0, np.abs(fd).max())
    pr = model.sample_function_from_prior_each_layer(nFeatures=100, generator=g)
    assert len(pr) == 2 and np.isfinite(pr[1](np.random.default_rng(0).random((5, 2)))).all()


def test_bench_accounting_of_dead_rows_and_active_blocks():
    """bench.py's bookkeeping of the work the step does not do: columns per layer with the dead rows pruned,
    executed GEMM flops, and the share of 128-column blocks a layer backward keeps -- on the synthetic fidelity layout and on
    an interleaved one."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg = dict(d=8, L=2, M=512, N=8192, S=8)
    fid = np.zeros(8192)
    fid[:2048] = 1.0                                   # synthetic layout: the first quarter is the top fidelity
    assert bench.panel_columns(cfg) == [8192, 65536]
    assert bench.panel_columns(cfg, [8192, 2048]) == [8192, 16384]
    assert bench.executed_gemm_flops(cfg) == 5.0 * 512 ** 2 * (8192 + 65536)
    assert bench.executed_gemm_flops(cfg, [8192, 2048]) == 5.0 * 512 ** 2 * (8192 + 16384)
    # reference layout, sorted rows: the top layer's backward keeps a quarter of its blocks, layer 0 all of them
    assert bench.backward_active_fractions(cfg, fid, None, True) == [1.0, 0.25]
    assert bench.backward_active_fractions(cfg, fid, None, False) == [1.0, 1.0]
    assert bench.executed_gemm_flops(cfg, None, [1.0, 0.25]) == 512 ** 2 * (5.0 * 8192 + (2.0 + 0.75) * 65536)
    # pruned: every remaining block is active
    assert bench.backward_active_fractions(cfg, fid, [8192, 2048], True) == [1.0, 1.0]
    # interleaved fidelities: every block of 16 base rows x 8 samples holds a top-fidelity row -- nothing to skip
    fid2 = np.zeros(8192)
    fid2[::4] = 1.0
    assert bench.backward_active_fractions(cfg, fid2, None, True) == [1.0, 1.0]
    # three fidelities: the middle layer is reached by its own rows and by the rows the top layer sends gradient back to
    cfg3 = dict(d=4, L=3, M=128, N=1024, S=4)
    fid3 = np.zeros(1024)
    fid3[:256], fid3[256:512] = 2.0, 1.0
    assert bench.backward_active_fractions(cfg3, fid3, None, True) == [1.0, 0.5, 0.25]
    assert bench.panel_columns(cfg3, [1024, 512, 256]) == [1024, 2048, 1024]


def test_tiny_step_descriptor_layout_and_argument_checks():
    """mobocmf_tiny_model as ctypes sees it == as the library sees it (the size queries read fields on both sides of the
    pointer block), the flat-vector / workspace sizes follow the documented layout, and a malformed descriptor is refused on
    the HOST, before any launch (no GPU needed)."""
    import ctypes
    from mobocmf_amd import _lib
    lib = _lib.load()
    T = _lib.TinyModel()
    T.L, T.M, T.d, T.S, T.N = 3, 20, 5, 2, 57
    T.rows[0], T.rows[1], T.rows[2] = 57, 30, 11
    flat, wb = ctypes.c_int64(), ctypes.c_size_t()
    assert lib.mobocmf_tiny_flat_len(ctypes.byref(T), ctypes.byref(flat)) == _lib.OK
    H = [1 + 5, 5 + 10, 5 + 10]
    assert flat.value == sum(h + 20 + 400 for h in H) + 3
    assert lib.mobocmf_tiny_work_bytes(ctypes.byref(T), ctypes.byref(wb)) == _lib.OK
    cols = [57, 60, 22]
    pool = sum(c * (2 * 20 + 11) for c in cols) + 3 * max(20, 8) * max(cols)
    assert wb.value == 8 * (((flat.value + 1) // 2) * 2 + pool + 256 * 17)
    for field, bad in (("L", 0), ("L", 4), ("M", 0), ("d", 0), ("S", 0)):
        B = _lib.TinyModel.from_buffer_copy(T)
        setattr(B, field, bad)
        assert lib.mobocmf_tiny_work_bytes(ctypes.byref(B), ctypes.byref(wb)) == _lib.BAD_ARG, field
    assert lib.mobocmf_tiny_flat_len(None, ctypes.byref(flat)) == _lib.BAD_ARG
    # the step itself: every check below fails before a launch could happen
    arr = (_lib.TinyModel * 1)(T)
    host = ctypes.cast(arr, ctypes.c_void_p)
    step = lambda n, mode, h=host: lib.mobocmf_tiny_elbo_step(h, host, n, 1e-3, 0.9, 0.999, 1e-8, mode, None)
    assert step(1, 1) == _lib.BAD_ARG              # no pointers set
    assert step(0, 1) == _lib.BAD_ARG and step(1, 3) == _lib.BAD_ARG and step(1, 1, None) == _lib.BAD_ARG
    arr[0].M = 33                                  # beyond MOBOCMF_TINY_MAX_M
    assert step(1, 1) == _lib.BAD_ARG


def test_coop_step_argument_checks_are_made_on_the_host():
    """mobocmf_coop_elbo_step / mobocmf_coop_work_bytes refuse malformed calls on the HOST, before any HIP call (no GPU needed):
    no models, no sync words, more workgroups per surrogate than the interface allows, a mode the kernel does not have, a
    descriptor without its pointers, M beyond MOBOCMF_COOP_MAX_M.  The workspace grows with M^2 (the layer's matrices) and with
    the panel columns; the flat layout is the one-workgroup kernel's."""
    import ctypes
    from mobocmf_amd import _lib
    lib = _lib.load()
    T = _lib.TinyModel()
    T.L, T.M, T.d, T.S, T.N = 2, 100, 2, 4, 300
    T.rows[0], T.rows[1] = 300, 75
    wb, wb2 = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.mobocmf_coop_work_bytes(ctypes.byref(T), ctypes.byref(wb)) == _lib.OK
    Mp = 112
    assert wb.value > 8 * 2 * 15 * Mp * Mp        # 7 + 2 * 8 matrices per layer ... at least the 15 that always exist
    B = _lib.TinyModel.from_buffer_copy(T)
    B.rows[1] = 150
    assert lib.mobocmf_coop_work_bytes(ctypes.byref(B), ctypes.byref(wb2)) == _lib.OK and wb2.value > wb.value
    for field, bad in (("L", 0), ("L", 4), ("M", 0), ("M", _lib.COOP_MAX_M + 1), ("d", 0), ("S", 0)):
        B = _lib.TinyModel.from_buffer_copy(T)
        setattr(B, field, bad)
        assert lib.mobocmf_coop_work_bytes(ctypes.byref(B), ctypes.byref(wb2)) == _lib.BAD_ARG, field
    arr = (_lib.TinyModel * 1)(T)
    host = ctypes.cast(arr, ctypes.c_void_p)
    sync = (ctypes.c_int64 * 32)()
    used = ctypes.c_int32(-7)
    step = lambda n=1, wgs=0, mode=1, h=host, sw=sync: lib.mobocmf_coop_elbo_step(
        h, host, n, wgs, ctypes.cast(sw, ctypes.c_void_p) if sw is not None else None, 1e-3, 0.9, 0.999, 1e-8, mode,
        ctypes.byref(used), None)
    assert step() == _lib.BAD_ARG                  # a descriptor without pointers
    assert step(n=0) == _lib.BAD_ARG and step(wgs=65) == _lib.BAD_ARG and step(wgs=-1) == _lib.BAD_ARG
    assert step(mode=5) == _lib.BAD_ARG and step(mode=-1) == _lib.BAD_ARG
    # MOBOCMF_STEP_CHAIN_VALID goes with the forward-only and the input-gradient mode only
    for m in (0, 1, 4):
        assert step(mode=m | _lib.STEP_CHAIN_VALID) == _lib.BAD_ARG, m
    assert step(h=None) == _lib.BAD_ARG and step(sw=None) == _lib.BAD_ARG
    assert used.value == -7                        # nothing was chosen, nothing launched


def test_an_abandoned_in_launch_wait_is_not_a_failed_pivot():
    """mobocmf_check_info's pivot -1 (include/mobocmf_hip.h): a one-launch form gave up a bounded wait.  The host mirror raises
    InLaunchWaitAbandoned (a FloatingPointError) for it and keeps NotPSDError / the jitter ladder for real pivots."""
    from mobocmf_amd import functional as F
    with pytest.raises(F.InLaunchWaitAbandoned, match="potrf_cols=4"):
        F.raise_if_abandoned(-1, "layer 0")
    assert issubclass(F.InLaunchWaitAbandoned, FloatingPointError)
    F.raise_if_abandoned(0)
    F.raise_if_abandoned(17)
