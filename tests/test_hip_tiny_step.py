"""mobocmf_tiny_elbo_step -- the whole ELBO step of small surrogates in ONE launch -- against the oracle
(mfdgp.py:174-196 + variational_elbo_mf.py:24-51 + blackbox_mfdgp_fitter.py:161-171 restated) and against the layer path."""
import numpy as np
import pytest
import torch

from mobocmf_amd.util import synthetic
from oracle import mfdgp_oracle as O
from tests.test_hip_model import _model_param_for, _raw_from_model, rel

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = [dict(d=1, L=2, M=16, N=16, S=1, seed=0), dict(d=1, L=2, M=16, N=16, S=4, seed=1),
         dict(d=2, L=2, M=12, N=40, S=3, seed=2), dict(d=5, L=3, M=20, N=57, S=2, seed=3),
         dict(d=8, L=3, M=32, N=64, S=4, seed=4), dict(d=3, L=1, M=9, N=30, S=1, seed=5),
         dict(d=2, L=2, M=32, N=128, S=8, seed=6), dict(d=4, L=3, M=7, N=23, S=1, seed=7)]
IDS = ["d%d_L%d_M%d_N%d_S%d" % (c["d"], c["L"], c["M"], c["N"], c["S"]) for c in CASES]


def _problem(cfg, shuffle=True):
    prob = synthetic.make_problem(**cfg)
    N, S = cfg["N"], cfg["S"]
    perm = np.random.default_rng(11).permutation(N) if shuffle else np.arange(N)
    tc = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    x, y, fid = tc(prob["x"])[perm], tc(prob["y"])[perm], tc(prob["fid"])[perm]
    eps = [None] + [tc(e).reshape(N, S)[perm].reshape(-1) for e in prob["eps"][1:]]
    return prob, x, y, fid, eps


def _tiny(models, xs, ys, fids, epss, lr=1e-2, want_grad=False):
    from mobocmf_amd.util.tiny_step import TinyELBOStep
    return TinyELBOStep(models, [x.shape[0] for x in xs], [x.to(DEV) for x in xs], [y.to(DEV) for y in ys],
                        [f.to(DEV) for f in fids], lr=lr,
                        fixed_eps=[None if e is None else [None if v is None else v.to(DEV) for v in e] for e in epss],
                        want_grad=want_grad, force=True)


@pytest.mark.parametrize("cfg", CASES, ids=IDS)
def test_tiny_step_gradients_match_oracle(cfg):
    """ELBO, scaled KL and every raw-parameter gradient of one launch (no update) vs the oracle's autograd through its dense
    evaluation of every layer at every row, on a shuffled batch.  Gradient gate 1e-5 as for the layer path
    (test_hip_pruned_oracle.py): at d = 1 cond(K_mm + 1e-6 I) ~ 1e9 and either implementation carries ~cond * eps."""
    prob, x, y, fid, eps = _problem(cfg)
    L, S = cfg["L"], cfg["S"]
    model = synthetic.model_from_problem(prob, num_samples_for_training=S, device=DEV)
    raw = _raw_from_model(model, L)
    e_o, skl_o = O.elbo(O.state_from_raw(raw), x, y, fid, eps=eps, S=S)
    (-e_o).backward()
    step = _tiny([model], [x], [y], [fid], [eps], want_grad=True)
    grads = step.gradients()[0]
    step.check()
    out = step.losses[0].cpu()
    assert rel(out[0], e_o) < 1e-9 and rel(out[1], skl_o) < 1e-9 and rel(out[2], -e_o) < 1e-9
    for l in range(L):
        for key, tt in raw["layers"][l].items():
            p = _model_param_for(model, l, key)
            gref = tt.grad if key != "L_S" else torch.tril(tt.grad)
            assert rel(grads[p].reshape(gref.shape), gref) < 1e-5, (l, key, rel(grads[p].reshape(gref.shape), gref))
        lk = getattr(model, f"hidden_layer_likelihood_{l}")
        assert rel(grads[lk.raw_noise].reshape(()), raw["raw_noise"][l].grad) < 1e-5


@pytest.mark.parametrize("cfg", CASES[:5], ids=IDS[:5])
def test_tiny_step_trajectory_matches_oracle(cfg):
    """Three fused steps (forward + ELBO + backward + Adam in one launch each) vs oracle.elbo_step with torch.optim.Adam:
    the loss of every step and every parameter afterwards."""
    prob, x, y, fid, eps = _problem(cfg)
    L, S = cfg["L"], cfg["S"]
    model = synthetic.model_from_problem(prob, num_samples_for_training=S, device=DEV)
    raw = _raw_from_model(model, L)
    opt = torch.optim.Adam(O.flatten_raw(raw), lr=1e-2)
    step = _tiny([model], [x], [y], [fid], [eps], lr=1e-2)
    for k in range(3):
        lo, klo = O.elbo_step(raw, opt, x, y, fid, eps, S, ref_equiv=False)
        step.step()
        step.check()
        assert rel(step.loss[0], lo) < 1e-7, (k, rel(step.loss[0], lo))
        assert rel(step.kl[0], klo) < 1e-7, k
    assert int(step.steps_done[0]) == 3
    for l in range(L):
        for key, tt in raw["layers"][l].items():
            p = _model_param_for(model, l, key)
            assert rel(p.reshape(tt.shape), tt.detach()) < 1e-6, (l, key, rel(p.reshape(tt.shape), tt.detach()))
        lk = getattr(model, f"hidden_layer_likelihood_{l}")
        assert rel(lk.raw_noise.reshape(()), raw["raw_noise"][l].detach()) < 1e-6


def test_tiny_step_group_equals_single_models_and_respects_frozen_parameters():
    """Three surrogates in one launch == each alone; parameters with requires_grad = False (fix_variational_hypers(True):
    noise and L_S, mfdgp.py:208-212) are left untouched."""
    cfgs = [CASES[1], dict(CASES[1], seed=9), CASES[2]]
    built = []
    for cfg in cfgs:
        prob, x, y, fid, eps = _problem(cfg)
        a = synthetic.model_from_problem(prob, num_samples_for_training=cfg["S"], device=DEV)
        b = synthetic.model_from_problem(prob, num_samples_for_training=cfg["S"], device=DEV)
        for mdl in (a, b):
            mdl.fix_variational_hypers(True)
        built.append((a, b, x, y, fid, eps))
    group = _tiny([t[0] for t in built], [t[2] for t in built], [t[3] for t in built], [t[4] for t in built],
                  [t[5] for t in built])
    singles = [_tiny([t[1]], [t[2]], [t[3]], [t[4]], [t[5]]) for t in built]
    before = [t[0].hidden_layer_0.variational_strategy._variational_distribution.chol_variational_covar.clone() for t in built]
    for _ in range(4):
        group.step()
        for s in singles:
            s.step()
    group.check()
    for i, (a, b, *_r) in enumerate(built):
        singles[i].check()
        # (not bit-equal: the launch geometry follows the WIDEST model of the group -- 512 threads instead of 256 -- and
        # with it the order of the workgroup reductions)
        assert rel(group.losses[i], singles[i].losses[0]) < 1e-9
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert rel(pa, pb) < 1e-9
        assert torch.equal(a.hidden_layer_0.variational_strategy._variational_distribution.chol_variational_covar, before[i])
        assert not torch.equal(a.hidden_layer_0.variational_strategy._variational_distribution.variational_mean,
                               torch.zeros_like(a.hidden_layer_0.variational_strategy._variational_distribution.variational_mean))


def test_tiny_step_draws_the_same_eps_as_the_layer_path():
    """Without explicit eps both paths draw from the layers' Philox streams (seed, call counter, row): the same model state
    gives the same trajectory through GraphedELBOStep (layer entry points, HIP graph) and through the one-launch step."""
    from mobocmf_amd.mlls import VariationalELBOMF
    from mobocmf_amd.util.graphed_step import GraphedELBOStep
    cfg = dict(d=1, L=2, M=16, N=16, S=4, seed=3)
    prob, x, y, fid, _ = _problem(cfg)
    torch.manual_seed(5)
    a = synthetic.model_from_problem(prob, num_samples_for_training=4, device=DEV)
    torch.manual_seed(5)
    b = synthetic.model_from_problem(prob, num_samples_for_training=4, device=DEV)
    torch.manual_seed(77)
    ga = GraphedELBOStep(a, VariationalELBOMF(a, 16, 2), x.to(DEV), y[:, None].to(DEV), fid[:, None].to(DEV), lr=3e-3)
    for la, lb in zip(a._layers(), b._layers()):      # b's streams: a's seeds, counters at zero
        lb._rng(torch.device(DEV, torch.cuda.current_device())).copy_(la._rng(la._rng_state.device))
    tb = _tiny([b], [x], [y], [fid], [None], lr=3e-3)
    for k in range(5):
        loss, kl = ga.step()
        tb.step()
        ga.stream.synchronize()
        tb.check()
        assert rel(tb.loss[0], loss) < 1e-9, (k, rel(tb.loss[0], loss))
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert rel(pb, pa) < 1e-7


def test_tiny_step_reports_a_failed_cholesky():
    from mobocmf_amd.layers.mfdgp_hidden_layer import NotPSDError
    cfg = CASES[0]
    prob, x, y, fid, eps = _problem(cfg)
    model = synthetic.model_from_problem(prob, num_samples_for_training=1, device=DEV)
    with torch.no_grad():      # a negative output scale cannot come out of softplus; a NaN lengthscale makes K_mm NaN
        model.hidden_layer_1.covar_module.kernels[1].base_kernel.raw_lengthscale.fill_(float("nan"))
    step = _tiny([model], [x], [y], [fid], [eps])
    step.step()
    with pytest.raises((NotPSDError, FloatingPointError)):
        step.check()


def test_eligibility_rules():
    from mobocmf_amd.util.tiny_step import eligible
    cfg = CASES[0]
    prob, x, y, fid, eps = _problem(cfg)
    model = synthetic.model_from_problem(prob, num_samples_for_training=1, device=DEV)
    assert eligible(model, x.to(DEV), fid.to(DEV))
    assert not eligible(model, x, fid)                                  # host batch
    assert not eligible(model, x.to(DEV), torch.zeros_like(fid).to(DEV))     # no row at the top fidelity
    big = synthetic.make_problem(d=2, L=2, M=48, N=64, S=1, seed=0)
    assert not eligible(synthetic.model_from_problem(big, num_samples_for_training=1, device=DEV),
                        torch.as_tensor(big["x"]).to(DEV), torch.as_tensor(big["fid"]).to(DEV))
    # accepted by the kernel, but one workgroup would be slower than the layer path's grid-filling launches: the speed rule
    wide = synthetic.make_problem(d=2, L=2, M=32, N=256, S=4, seed=0)
    mw = synthetic.model_from_problem(wide, num_samples_for_training=4, device=DEV)
    xw, fw = torch.as_tensor(wide["x"]).to(DEV), torch.as_tensor(wide["fid"]).to(DEV)
    assert eligible(mw, xw, fw, speed_rule=False) and not eligible(mw, xw, fw)


# ------------------------------------------------------------------ conditioned training in 3 + n_con launches
def test_tiny_conditioned_iteration_matches_oracle():
    """TinyConditionedStep -- forward-only launch, theta / omega factor launches on the top layers' moments, step launch with
    the factor gradients entering as seeds -- vs the oracle's joint loss (blackbox_mfdgp_fitter.py:270-343 restated): the
    loss and d loss / d (m, L_S) of every layer of every surrogate, explicit x~ and eps."""
    from mobocmf_amd.util.tiny_step import TinyConditionedStep
    from tests.helpers import oracle_state, to_t
    from tests.test_hip_conditioned import _fitter
    n_obj, n_con, N, P, T, d = 2, 1, 12, 5, 10, 2
    fitter, probs = _fitter(n_obj, n_con, N)
    g = torch.Generator().manual_seed(0)
    pareto_set = torch.rand(P, d, dtype=torch.float64, generator=g)
    pareto_front = torch.randn(P, n_obj, dtype=torch.float64, generator=g) * 0.5
    x_tilde = torch.rand(T, d, dtype=torch.float64, generator=g)
    fitter.set_pareto_solution(pareto_set, pareto_front)
    eps_all, objs, cons = {}, [], []
    for idx, (tag, i, h) in enumerate(fitter._handlers()):
        e = torch.randn(N + P + T, dtype=torch.float64, generator=g)
        eps_all[(tag, i)] = [None, e.to(DEV)]
        st = oracle_state(probs[idx], requires_grad=True)
        rec = {"state": st, "x": to_t(probs[idx]["x"]), "y": to_t(probs[idx]["y"]), "fid": to_t(probs[idx]["fid"]),
               "eps_batch": [None, e[:N]], "eps_pareto": [None, e[N:N + P]], "eps_tilde": [None, e[N + P:]]}
        (objs if tag == "OBJ" else cons).append(rec)
        h.mfdgp.fix_variational_hypers_cond(True)
    loss_o = O.conditioned_loss(objs, cons, pareto_set, pareto_front, x_tilde, fitter.thresholds_cons, fitter.eps)
    loss_o.backward()
    step = TinyConditionedStep(fitter, lr=1e-3, fixed_x_tilde=x_tilde.to(DEV), fixed_eps=eps_all, want_grad=True)
    grads = step.gradients()
    step.check()
    assert rel(step.loss, loss_o) < 1e-8, rel(step.loss, loss_o)
    for k, (rec, (tag, i, h)) in enumerate(zip(objs + cons, fitter._handlers())):
        for l in range(2):
            vd = getattr(h.mfdgp, f"hidden_layer_{l}").variational_strategy._variational_distribution
            assert rel(grads[k][vd.variational_mean], rec["state"]["layers"][l]["m"].grad) < 1e-6, (tag, i, l)
            assert rel(grads[k][vd.chol_variational_covar], torch.tril(rec["state"]["layers"][l]["L_S"].grad)) < 1e-6, (tag, i, l)
    # the step itself: only m and L_S move (fix_variational_hypers_cond), by Adam's first step = lr * sign(gradient)
    before = [[p.detach().clone() for p in h.mfdgp.parameters()] for _, _, h in fitter._handlers()]
    step.step()
    step.check()
    for k, (_, _, h) in enumerate(fitter._handlers()):
        for p, p0 in zip(h.mfdgp.parameters(), before[k]):
            if p.requires_grad:
                gk = grads[k][p]
                moved = (p.detach() - p0)
                assert torch.allclose(moved[gk != 0], -1e-3 * torch.sign(gk[gk != 0]), rtol=1e-5, atol=0)
            else:
                assert torch.equal(p.detach(), p0)


def test_tiny_conditioned_step_draws_fresh_shared_x_tilde():
    """Without fixed x~ the forward-only launch draws it (U(0,1), Philox keyed by the step's seed and iteration counter):
    inside (0, 1), the same points for every surrogate, new ones at every iteration; the joint loss stays finite and the
    fitter's conditioned training runs through this step."""
    from mobocmf_amd.util.tiny_step import TinyConditionedStep
    from tests.test_hip_conditioned import _fitter
    fitter, _ = _fitter(2, 1, 12)
    g = torch.Generator().manual_seed(1)
    fitter.set_pareto_solution(torch.rand(5, 2, dtype=torch.float64, generator=g),
                               torch.randn(5, 2, dtype=torch.float64, generator=g) * 0.3)
    for _, _, h in fitter._handlers():
        h.mfdgp.fix_variational_hypers_cond(True)
    step = TinyConditionedStep(fitter, lr=1e-3)
    seen = []
    for _ in range(3):
        step.step()
        step.check()
        xt = [xr[step.P:step.P + step.T].clone() for xr in step.x_rows]
        assert all(torch.equal(xt[0], v) for v in xt[1:])
        assert bool(((xt[0] > 0) & (xt[0] < 1)).all())
        assert all(not torch.equal(xt[0], s) for s in seen)
        seen.append(xt[0])
        assert bool(torch.isfinite(step.loss))
    assert int(step.xrng[1]) == 3
    fitter.lr_2 = 5e-3
    l0 = float(step.loss)
    fitter.train_conditioned_mfdgps(num_iters=150)
    xt = torch.rand(10, 2, dtype=torch.float64, generator=g).to(DEV)
    torch.manual_seed(0)
    assert float(fitter.conditioned_loss(xt)) < l0


# ------------------------------------------------------------------ acquisition: moments + dX in two launches
def _two_fitters(seed=0):
    """An 'unconditioned' and a 'conditioned' fitter over the same three small problems (different variational parameters)."""
    from tests.test_hip_conditioned import _fitter
    fu, _ = _fitter(2, 1, 12)
    fc, _ = _fitter(2, 1, 12)
    g = torch.Generator().manual_seed(seed)
    for _, _, h in fc._handlers():
        for layer in h.mfdgp._layers():
            vd = layer.variational_strategy._variational_distribution
            with torch.no_grad():
                vd.variational_mean.add_(0.05 * torch.randn(vd.variational_mean.shape, dtype=torch.float64, generator=g).to(DEV))
                vd.chol_variational_covar.mul_(0.7)
    return fu, fc


@pytest.mark.parametrize("fidelity", [0, 1])
def test_tiny_predict_group_matches_predict_for_acquisition_and_its_input_gradient(fidelity):
    """TinyPredictGroup (forward-only launch for all models; mode-3 launch for d/dX) vs MFDGP.predict_for_acquisition through
    the layer entry points (itself pinned to the oracle in test_hip_model.py): moments to 1e-9, d(sum of weighted moments)/dX
    to 1e-7, for T = 7 points, S = 25 fixed samples."""
    from mobocmf_amd.util.tiny_step import TinyPredictGroup
    fu, fc = _two_fitters()
    models = [h.mfdgp for _, _, h in fu._handlers()] + [h.mfdgp for _, _, h in fc._handlers()]
    g = torch.Generator().manual_seed(3)
    X = torch.rand(7, 2, dtype=torch.float64, generator=g).to(DEV)
    wm = torch.randn(len(models), 7, dtype=torch.float64, generator=g).to(DEV)
    wv = torch.randn(len(models), 7, dtype=torch.float64, generator=g).to(DEV)
    Xa = X.clone().requires_grad_(True)
    ref_m, ref_v = [], []
    for m in models:
        m.eval()
        mu, v = m.predict_for_acquisition(Xa, fidelity)
        m.train()
        ref_m.append(mu), ref_v.append(v)
    ref_m, ref_v = torch.stack(ref_m), torch.stack(ref_v)
    ((ref_m * wm).sum() + (ref_v * wv).sum()).backward()
    grp = TinyPredictGroup(models, fidelity, 7, 2)
    Xb = X.clone().requires_grad_(True)
    mus, v = grp.acquisition_moments(Xb)
    ((mus * wm).sum() + (v * wv).sum()).backward()
    assert rel(mus, ref_m) < 1e-9 and rel(v, ref_v) < 1e-8, (rel(mus, ref_m), rel(v, ref_v))
    assert rel(Xb.grad, Xa.grad) < 1e-7, rel(Xb.grad, Xa.grad)


def test_coupled_jes_through_the_one_launch_kernel_equals_the_layer_path():
    """JESMOC_MFDGP.coupled_acq (sum over black-boxes of 0.5 clamp(log v_uncond - log v_cond, 0), JESMOC_MFDGP.py:38-52,
    125-135) and its gradient w.r.t. the candidates, both fidelities: one-launch kernel vs layer entry points."""
    from mobocmf_amd.acquisition_functions.JESMOC_MFDGP import JESMOC_MFDGP
    fu, fc = _two_fitters(1)
    fc.pareto_set = torch.zeros(1, 2, dtype=torch.float64, device=DEV)
    fc.pareto_front = torch.zeros(1, 2, dtype=torch.float64, device=DEV)
    acq = JESMOC_MFDGP.__new__(JESMOC_MFDGP)
    acq.blackbox_mfdgp_fitter_uncond, acq.blackbox_mfdgp_fitter_cond = fu, fc
    acq.num_fidelities, acq.eval_highest_fidelity = 2, False
    acq.standard_bounds = torch.tensor([[0.0, 0.0], [1.0, 1.0]], dtype=torch.float64, device=DEV)
    acq.objectives, acq.constraints, acq.costs_blackboxes = {0: {}, 1: {}}, {0: {}, 1: {}}, {0: {"total": 0.0}, 1: {"total": 0.0}}
    for f in (0, 1):
        for name, is_con in (("bb0", False), ("bb1", False), ("bb2", True)):
            acq.add_blackbox(f, name, is_constraint=is_con)
    X = torch.rand(5, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(9)).to(DEV)
    for f in (0, 1):
        res = {}
        for tiny in (True, False):
            acq.use_tiny_step = tiny
            Xg = X.clone().requires_grad_(True)
            v = acq.coupled_acq(Xg, fidelity=f)
            v.sum().backward()
            res[tiny] = (v.detach(), Xg.grad)
        assert acq._tiny_groups[(f, 5, 2)] is not None
        assert float(res[False][0].abs().max()) > 0
        assert rel(res[True][0], res[False][0]) < 1e-8, (f, rel(res[True][0], res[False][0]))
        assert rel(res[True][1], res[False][1]) < 1e-6, (f, rel(res[True][1], res[False][1]))


def test_fitter_falls_back_to_the_layer_path_after_a_failed_cholesky(monkeypatch):
    """A Cholesky that fails inside the one-launch step cannot be retried there (no host in the loop): the fitter rolls the
    group back to the last verified epoch and the layer path -- per-step jitter ladder, as the reference -- finishes the
    phase with the optimiser's moments and step count carried over."""
    import warnings
    from mobocmf_amd.layers.mfdgp_hidden_layer import NotPSDError
    from mobocmf_amd.util import blackbox_mfdgp_fitter as BF
    from mobocmf_amd.util import tiny_step as TS
    from mobocmf_amd.util.graphed_step import GraphedELBOStep
    monkeypatch.setattr(BF, "ITER_PRINT", 5)
    x, y, fid = synthetic.forrester_problem(0)
    tt = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    fitter = BF.BlackBoxMFDGPFitter(2, 16, num_epochs_1=12, num_epochs_2=0, device=DEV)
    fitter.verbose = False
    fitter.initialize_mfdgp(tt(x), tt(y)[:, None], tt(fid)[:, None], "obj1")
    calls = {"n": 0}
    real_check = TS.TinyELBOStep.check

    def failing_check(self):
        calls["n"] += 1
        if calls["n"] == 2:      # the verdict at epoch 5 (epoch 0 passed)
            raise NotPSDError("injected")
        return real_check(self)

    monkeypatch.setattr(TS.TinyELBOStep, "check", failing_check)
    seen = {}
    real_init = GraphedELBOStep.__init__

    def spy_init(self, *a, **k):
        real_init(self, *a, **k)
        seen["step"] = self

    monkeypatch.setattr(GraphedELBOStep, "__init__", spy_init)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        fitter._train_mfdgp_graphed(True, 12, 3e-3)
    assert any("rolling back" in str(m.message) for m in w)
    g = seen["step"]                                   # the layer path took over ...
    assert int(g.optimizer.steps_done) == 12           # ... 1 verified epoch from the one-launch step + 11 of its own
    st = g.optimizer.state[0]
    assert float(st["exp_avg_sq"].abs().sum()) > 0
    for p in fitter.get_model("obj1").parameters():
        assert bool(torch.isfinite(p).all())


def _fuzz_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        L = int(rng.integers(1, 4))
        d = int(rng.integers(1, 9))
        M = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 24, 31, 32]))
        N = int(rng.integers(max(M, 4), 90))
        S = int(rng.choice([1, 1, 2, 3, 5, 8]))
        if N * S > 400:
            continue
        out.append(dict(d=d, L=L, M=M, N=N, S=S, seed=int(rng.integers(1 << 30))))
    return out


FUZZ = _fuzz_cases(24, seed=77)


@pytest.mark.parametrize("cfg", FUZZ, ids=["d%d_L%d_M%d_N%d_S%d" % (c["d"], c["L"], c["M"], c["N"], c["S"]) for c in FUZZ])
def test_fuzz_tiny_step_against_the_layer_path(cfg):
    """Random shapes inside the kernel's limits (M = 1 ... 32 incl. non-powers of two, d = 1 ... 8, 1-3 layers, S = 1 ... 8,
    ragged row counts): ELBO, scaled KL and every raw-parameter gradient of the one-launch step vs the layer entry points
    (GraphedELBOStep's pruned forward / backward, themselves pinned to the oracle in test_hip_pruned_oracle.py)."""
    from mobocmf_amd.mlls import VariationalELBOMF
    prob, x, y, fid, eps = _problem(cfg)
    L, S, N = cfg["L"], cfg["S"], cfg["N"]
    if int((fid >= L - 1).sum()) < 1:
        pytest.skip("no row at the top fidelity")
    ma = synthetic.model_from_problem(prob, num_samples_for_training=S, device=DEV)
    mb = synthetic.model_from_problem(prob, num_samples_for_training=S, device=DEV)
    ma.set_check_pd(False)
    fidv = fid.reshape(-1)
    rows = [int((fidv >= l).sum()) for l in range(L)]
    order = torch.argsort(fidv, descending=True, stable=True)
    xo, yo, fo = x[order].to(DEV), y[order][:, None].to(DEV), fid[order][:, None].to(DEV)
    eo = [None if e is None else e.reshape(N, S)[order][:rows[l]].reshape(-1).contiguous().to(DEV) for l, e in enumerate(eps)]
    e_ref, skl_ref = VariationalELBOMF(ma, N, L)(ma(xo, eps=eo, rows=rows), yo.T, fo)
    (-e_ref).backward()
    step = _tiny([mb], [x], [y], [fid], [eps], want_grad=True)
    grads = step.gradients()[0]
    step.check()
    out = step.losses[0]
    assert rel(out[0], e_ref) < 1e-9 and rel(out[1], skl_ref) < 1e-9, (rel(out[0], e_ref), rel(out[1], skl_ref))
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        if pa.grad is None:
            continue
        ga, gb = pa.grad, grads[pb]
        if pa.dim() == 2 and pa.shape[0] == pa.shape[1] and pa.shape[0] == cfg["M"]:
            ga = torch.tril(ga)
        scale = float(ga.abs().max())
        assert scale == 0.0 or float((gb - ga).abs().max()) / scale < 1e-5, (tuple(pa.shape), float((gb - ga).abs().max()) / scale)


def test_one_launch_conditioned_iteration_equals_the_three_launch_form():
    """Mode 4 -- forward, grid barrier, the theta / omega factor gradients formed by every workgroup for its own model,
    backward, Adam: ONE cooperative launch -- vs forward-only launch + mobocmf_cond_factors_forward launches + step launch
    (itself pinned to the oracle above): the same losses, factor terms and parameters over three iterations, explicit x~, eps."""
    from mobocmf_amd.util.tiny_step import TinyConditionedStep
    from tests.test_hip_conditioned import _fitter
    g = torch.Generator().manual_seed(4)
    P, T, N = 9, 10, 12
    ps = torch.rand(P, 2, dtype=torch.float64, generator=g)
    pf = torch.randn(P, 2, dtype=torch.float64, generator=g) * 0.4
    xt = torch.rand(T, 2, dtype=torch.float64, generator=g).to(DEV)
    runs = {}
    for one in (True, False):
        fitter, _ = _fitter(2, 1, N)
        fitter.set_pareto_solution(ps, pf)
        ge = torch.Generator().manual_seed(5)
        eps_all = {}
        for tag, i, h in fitter._handlers():
            h.mfdgp.fix_variational_hypers_cond(True)
            eps_all[(tag, i)] = [None, torch.randn(N + P + T, dtype=torch.float64, generator=ge).to(DEV)]
        step = TinyConditionedStep(fitter, lr=2e-3, fixed_x_tilde=xt, fixed_eps=eps_all)
        step.one_launch, step.use_graph = one, False
        hist = []
        for _ in range(3):
            step.step()
            step.check()
            hist.append((step.losses.clone(), step.factor_losses.clone()))
        assert step.one_launch is one      # the cooperative launch was accepted
        runs[one] = (hist, [p.detach().clone() for _, _, h in fitter._handlers() for p in h.mfdgp.parameters()])
    for (la, fa), (lb, fb) in zip(runs[True][0], runs[False][0]):
        assert rel(la, lb) < 1e-11 and rel(fa, fb) < 1e-11, (rel(la, lb), rel(fa, fb))
    for pa, pb in zip(runs[True][1], runs[False][1]):
        assert rel(pa, pb) < 1e-10


def test_in_launch_barrier_gives_up_instead_of_hanging():
    """The one-launch conditioned iteration waits for its models' workgroups at an arrival counter.  A wait that cannot end
    (here: the counter is knocked out of step, so the last workgroup waits for arrivals that never come) is abandoned after
    ~0.2-0.5 s: the model is flagged, its loss poisoned, ``check()`` raises -- the device is not hung, and the next launches
    run (the fitter then rolls back and continues on the layer path)."""
    import time
    from mobocmf_amd.layers.mfdgp_hidden_layer import NotPSDError
    from mobocmf_amd.util.tiny_step import TinyConditionedStep
    from tests.test_hip_conditioned import _fitter
    fitter, _ = _fitter(2, 1, 12)
    g = torch.Generator().manual_seed(2)
    fitter.set_pareto_solution(torch.rand(5, 2, dtype=torch.float64, generator=g), torch.randn(5, 2, dtype=torch.float64, generator=g) * 0.3)
    for _, _, h in fitter._handlers():
        h.mfdgp.fix_variational_hypers_cond(True)
    step = TinyConditionedStep(fitter, lr=1e-3)
    step.use_graph = False
    step.step()
    step.check()
    step.snapshot()
    step._barrier.fill_(1)               # 3 workgroups: arrivals 2, 3, 4 -- the third waits for a 6 that never comes
    t0 = time.perf_counter()
    step.step()
    with pytest.raises((NotPSDError, FloatingPointError)):
        step.check()
    assert time.perf_counter() - t0 < 5.0
    step.restore()
    step._barrier.zero_()
    step.infos.zero_()
    torch.cuda.synchronize()
    step.step()
    step.check()                          # the device is fine


def test_barrier_timeout_at_a_non_check_iteration_is_sticky_and_leaves_the_model_untouched():
    """The fitter checks every 1000 iterations only (blackbox_mfdgp_fitter.ITER_PRINT).  A workgroup that gives up at the
    in-launch barrier in iteration k must (i) not update its model from its peers' unpublished moments -- parameters, Adam
    state, step count and random streams stay as they were -- and (ii) still be reported by a check() many iterations later,
    although the launches in between rewrite ``info`` and the losses: the coupling's status word is only ever OR'd."""
    from mobocmf_amd.util.tiny_step import TinyConditionedStep
    from tests.test_hip_conditioned import _fitter
    fitter, _ = _fitter(2, 1, 12)
    g = torch.Generator().manual_seed(2)
    fitter.set_pareto_solution(torch.rand(5, 2, dtype=torch.float64, generator=g), torch.randn(5, 2, dtype=torch.float64, generator=g) * 0.3)
    for _, _, h in fitter._handlers():
        h.mfdgp.fix_variational_hypers_cond(True)
    step = TinyConditionedStep(fitter, lr=1e-3)
    step.use_graph = False
    step.step()
    step.check()
    step.snapshot()
    torch.cuda.synchronize()
    before = [[p.detach().clone() for p in m.parameters()] for m in step.models]
    adam_before = [t.clone() for t in step.exp_avg]
    steps_before = step.steps_done.clone()
    step._barrier.fill_(1)               # the last of the 3 workgroups waits for arrivals that never come
    step.step()
    torch.cuda.synchronize()
    gave_up = [i for i in range(len(step.models)) if int(step.infos[i, 0]) == -1]
    assert len(gave_up) == 1 and int(step._status.item()) == 1
    k = gave_up[0]
    for pa, pb in zip(step.models[k].parameters(), before[k]):
        assert torch.equal(pa, pb)                      # the model that timed out was not touched ...
    assert torch.equal(step.exp_avg[k], adam_before[k]) and int(step.steps_done[k]) == int(steps_before[k])
    other = [i for i in range(len(step.models)) if i != k]
    assert all(int(step.steps_done[i]) == int(steps_before[i]) + 1 for i in other)      # ... its peers completed
    step._barrier.zero_()                # the disturbance was transient: the following iterations run normally
    for _ in range(3):
        step.step()
    torch.cuda.synchronize()
    assert not bool((step.infos != 0).any()) and bool(torch.isfinite(step.losses).all())      # overwritten by now
    with pytest.raises(FloatingPointError):
        step.check()                     # ... but the status word remembers
    step.restore()
    step.step()
    step.check()


def test_one_launch_iteration_refused_beyond_the_devices_residency_falls_back():
    """mobocmf_tiny_elbo_step refuses mode 4 for more models than the device keeps resident at once (bounded by 64 and by
    hipOccupancyMaxActiveBlocksPerMultiprocessor x CU count for the kernel and its LDS size): MOBOCMF_BAD_ARG, nothing enqueued;
    TinyConditionedStep then issues the three-launch form."""
    import ctypes
    from mobocmf_amd import _lib
    from mobocmf_amd.util.tiny_step import TinyConditionedStep
    from tests.test_hip_conditioned import _fitter
    fitter, _ = _fitter(2, 1, 12)
    g = torch.Generator().manual_seed(2)
    fitter.set_pareto_solution(torch.rand(5, 2, dtype=torch.float64, generator=g), torch.randn(5, 2, dtype=torch.float64, generator=g) * 0.3)
    for _, _, h in fitter._handlers():
        h.mfdgp.fix_variational_hypers_cond(True)
    step = TinyConditionedStep(fitter, lr=1e-3)
    lib = _lib.require_device()
    n = len(step.models)
    big = (_lib.TinyModel * 65)(*[step.host[i % n] for i in range(65)])
    dev = torch.frombuffer(bytearray(bytes(big)), dtype=torch.uint8).to(step.device)
    rc = lib.mobocmf_tiny_elbo_step(ctypes.cast(big, ctypes.c_void_p), ctypes.c_void_p(dev.data_ptr()), 65, 1e-3, 0.9, 0.999, 1e-8,
                                    4, ctypes.c_void_p(step.stream.cuda_stream))
    assert rc == _lib.BAD_ARG
    # a launch that does not match its coupling record (here: fewer workgroups than the record was made for) stops in every
    # workgroup before anyone arrives at the barrier: status bit 1, nothing updated, nothing hangs
    torch.cuda.synchronize()
    sd = step.steps_done.clone()
    rc = lib.mobocmf_tiny_elbo_step(ctypes.cast(step.host, ctypes.c_void_p), ctypes.c_void_p(step._dev_table.data_ptr()), n - 1,
                                    1e-3, 0.9, 0.999, 1e-8, 4, ctypes.c_void_p(step.stream.cuda_stream))
    assert rc == _lib.OK
    torch.cuda.synchronize()
    assert int(step._status.item()) == 2 and torch.equal(step.steps_done, sd) and int(step._barrier.item()) == 0
    step._status.zero_()
    step.infos.zero_()
    step.use_graph = False
    step.step()
    step.check()
