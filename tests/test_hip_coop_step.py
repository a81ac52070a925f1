"""mobocmf_coop_elbo_step -- the whole ELBO step of mid-size surrogates (M <= 128) in ONE launch by several workgroups per
surrogate -- against the oracle (mfdgp.py:174-196 + variational_elbo_mf.py:24-51 + blackbox_mfdgp_fitter.py:161-171 restated),
against the one-workgroup kernel and against the layer path."""
import numpy as np
import pytest
import torch

from mobocmf_amd.util import synthetic
from oracle import mfdgp_oracle as O
from tests.test_hip_model import _model_param_for, _raw_from_model, rel
from tests.test_hip_tiny_step import _problem

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (the reference's loop: M = N, S = 1, two fidelities; BASELINE config 2: d = 2, M = 128, N = 512, S = 8; plus ragged / small / 3-layer shapes)
CASES = [dict(d=2, L=2, M=64, N=64, S=1, seed=0), dict(d=2, L=2, M=48, N=48, S=1, seed=1),
         dict(d=2, L=2, M=96, N=120, S=2, seed=2), dict(d=2, L=2, M=128, N=256, S=4, seed=3),
         dict(d=5, L=3, M=40, N=57, S=2, seed=4), dict(d=8, L=3, M=75, N=75, S=1, seed=5),
         dict(d=3, L=1, M=33, N=50, S=1, seed=6), dict(d=1, L=2, M=16, N=16, S=4, seed=7),
         dict(d=4, L=3, M=7, N=23, S=1, seed=8), dict(d=2, L=2, M=128, N=512, S=8, seed=9),
         # M <= 6: the variational means of TWO layers fall into one wavefront of the gradient assembly (found by
         # tools/fuzz_tiny_step.py ... coop: the layer's workspace base was formed wave-uniformly there)
         dict(d=1, L=2, M=4, N=64, S=1, seed=10), dict(d=1, L=3, M=4, N=100, S=4, seed=11), dict(d=4, L=2, M=2, N=40, S=2, seed=12)]
IDS = ["d%d_L%d_M%d_N%d_S%d" % (c["d"], c["L"], c["M"], c["N"], c["S"]) for c in CASES]


def _coop(models, xs, ys, fids, epss, lr=1e-2, want_grad=False, wgs=0):
    from mobocmf_amd.util.coop_step import CoopELBOStep
    step = CoopELBOStep(models, [x.shape[0] for x in xs], [x.to(DEV) for x in xs], [y.to(DEV) for y in ys],
                        [f.to(DEV) for f in fids], lr=lr,
                        fixed_eps=[None if e is None else [None if v is None else v.to(DEV) for v in e] for e in epss],
                        want_grad=want_grad, force=True)
    step.wgs_per_model = wgs
    return step


@pytest.mark.parametrize("cfg", CASES, ids=IDS)
def test_coop_step_gradients_match_oracle(cfg):
    """ELBO, scaled KL and every raw-parameter gradient of one launch (no update) vs the oracle's autograd through its dense
    evaluation of every layer at every row, on a shuffled batch.  Gates as for the one-workgroup kernel and the layer path:
    1e-5 on the gradients, the north star's 1e-4 where dozens of inducing points crowd [0,1]^2 (cond(K_mm + 1e-6 I) ~ 1e9:
    either implementation carries ~cond * eps, test_hip_fullsize.py at C2)."""
    prob, x, y, fid, eps = _problem(cfg)
    L, S = cfg["L"], cfg["S"]
    gate = 1e-4 if (cfg["d"] <= 2 and cfg["M"] >= 48) else 1e-5
    model = synthetic.model_from_problem(prob, num_samples_for_training=S, device=DEV)
    raw = _raw_from_model(model, L)
    e_o, skl_o = O.elbo(O.state_from_raw(raw), x, y, fid, eps=eps, S=S)
    (-e_o).backward()
    step = _coop([model], [x], [y], [fid], [eps], want_grad=True)
    grads = step.gradients()[0]
    step.check()
    out = step.losses[0].cpu()
    assert rel(out[0], e_o) < 1e-9 and rel(out[1], skl_o) < 1e-9 and rel(out[2], -e_o) < 1e-9, (out, e_o, skl_o)
    for l in range(L):
        for key, tt in raw["layers"][l].items():
            p = _model_param_for(model, l, key)
            gref = tt.grad if key != "L_S" else torch.tril(tt.grad)
            assert rel(grads[p].reshape(gref.shape), gref) < gate, (l, key, rel(grads[p].reshape(gref.shape), gref))
        lk = getattr(model, f"hidden_layer_likelihood_{l}")
        assert rel(grads[lk.raw_noise].reshape(()), raw["raw_noise"][l].grad) < gate


@pytest.mark.parametrize("wgs", [1, 2, 3, 7, 16, 40])
def test_any_number_of_workgroups_per_surrogate_gives_the_same_gradients(wgs):
    """The phases deal their tiles / column blocks to whatever workgroups there are: 1 (no barrier traffic at all) ... 40 (more
    than any phase has tasks) give the same ELBO and gradients up to summation order."""
    cfg = CASES[2]
    prob, x, y, fid, eps = _problem(cfg)
    a = synthetic.model_from_problem(prob, num_samples_for_training=cfg["S"], device=DEV)
    b = synthetic.model_from_problem(prob, num_samples_for_training=cfg["S"], device=DEV)
    sa = _coop([a], [x], [y], [fid], [eps], want_grad=True, wgs=wgs)
    sb = _coop([b], [x], [y], [fid], [eps], want_grad=True, wgs=4)
    ga, gb = sa.gradients()[0], sb.gradients()[0]
    sa.check()
    assert sa.wgs_used == wgs
    assert rel(sa.losses[0], sb.losses[0]) < 1e-12
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert rel(ga[pa], gb[pb]) < 1e-8, rel(ga[pa], gb[pb])


@pytest.mark.parametrize("cfg", CASES[:5], ids=IDS[:5])
def test_coop_step_trajectory_matches_oracle(cfg):
    """Three fused steps vs oracle.elbo_step with torch.optim.Adam: the loss of every step and every parameter afterwards."""
    prob, x, y, fid, eps = _problem(cfg)
    L, S = cfg["L"], cfg["S"]
    model = synthetic.model_from_problem(prob, num_samples_for_training=S, device=DEV)
    raw = _raw_from_model(model, L)
    opt = torch.optim.Adam(O.flatten_raw(raw), lr=1e-2)
    step = _coop([model], [x], [y], [fid], [eps], lr=1e-2)
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


def test_coop_group_equals_single_models_and_the_one_workgroup_kernel():
    """Three surrogates of different shapes in one launch == each alone; and at a size both kernels take (M = 16) the
    cooperative kernel follows the one-workgroup kernel's trajectory, frozen parameters untouched."""
    from tests.test_hip_tiny_step import _tiny
    cfgs = [CASES[1], CASES[4], CASES[7]]
    built = []
    for cfg in cfgs:
        prob, x, y, fid, eps = _problem(cfg)
        a = synthetic.model_from_problem(prob, num_samples_for_training=cfg["S"], device=DEV)
        b = synthetic.model_from_problem(prob, num_samples_for_training=cfg["S"], device=DEV)
        for mdl in (a, b):
            mdl.fix_variational_hypers(True)
        built.append((a, b, x, y, fid, eps))
    group = _coop([t[0] for t in built], [t[2] for t in built], [t[3] for t in built], [t[4] for t in built], [t[5] for t in built])
    singles = [_coop([t[1]], [t[2]], [t[3]], [t[4]], [t[5]]) for t in built[:2]] + \
        [_tiny([built[2][1]], [built[2][2]], [built[2][3]], [built[2][4]], [built[2][5]])]
    before = [t[0].hidden_layer_0.variational_strategy._variational_distribution.chol_variational_covar.clone() for t in built]
    for _ in range(4):
        group.step()
        for s in singles:
            s.step()
    group.check()
    for i, (a, b, *_r) in enumerate(built):
        singles[i].check()
        assert rel(group.losses[i], singles[i].losses[0]) < 1e-8, (i, rel(group.losses[i], singles[i].losses[0]))
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert rel(pa, pb) < 1e-8
        assert torch.equal(a.hidden_layer_0.variational_strategy._variational_distribution.chol_variational_covar, before[i])


def test_coop_step_draws_the_same_eps_as_the_layer_path():
    """Without explicit eps both paths draw from the layers' Philox streams: the same model state gives the same trajectory
    through GraphedELBOStep (layer entry points, HIP graph) and through the cooperative launch, at the reference's M = N = 64."""
    from mobocmf_amd.mlls import VariationalELBOMF
    from mobocmf_amd.util.graphed_step import GraphedELBOStep
    cfg = dict(d=2, L=2, M=64, N=64, S=2, seed=3)
    prob, x, y, fid, _ = _problem(cfg)
    torch.manual_seed(5)
    a = synthetic.model_from_problem(prob, num_samples_for_training=2, device=DEV)
    torch.manual_seed(5)
    b = synthetic.model_from_problem(prob, num_samples_for_training=2, device=DEV)
    torch.manual_seed(77)
    ga = GraphedELBOStep(a, VariationalELBOMF(a, 64, 2), x.to(DEV), y[:, None].to(DEV), fid[:, None].to(DEV), lr=3e-3)
    for la, lb in zip(a._layers(), b._layers()):
        lb._rng(torch.device(DEV, torch.cuda.current_device())).copy_(la._rng(la._rng_state.device))
    tb = _coop([b], [x], [y], [fid], [None], lr=3e-3)
    for k in range(5):
        loss, kl = ga.step()
        tb.step()
        ga.stream.synchronize()
        tb.check()
        assert rel(tb.loss[0], loss) < 1e-9, (k, rel(tb.loss[0], loss))
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert rel(pb, pa) < 1e-7


def test_coop_step_reports_a_failed_cholesky_and_replays_from_a_graph():
    from mobocmf_amd.layers.mfdgp_hidden_layer import NotPSDError
    cfg = CASES[1]
    prob, x, y, fid, eps = _problem(cfg)
    model = synthetic.model_from_problem(prob, num_samples_for_training=1, device=DEV)
    ref = synthetic.model_from_problem(prob, num_samples_for_training=1, device=DEV)
    step, sref = _coop([model], [x], [y], [fid], [eps]), _coop([ref], [x], [y], [fid], [eps])
    g = torch.cuda.CUDAGraph()      # an ordinary launch: capturable
    with torch.cuda.graph(g, stream=step.stream, capture_error_mode="thread_local"):
        step.step()
    for _ in range(3):
        with torch.cuda.stream(step.stream):
            g.replay()
        sref.step()
    step.check()
    sref.check()
    for pa, pb in zip(model.parameters(), ref.parameters()):
        assert torch.equal(pa, pb)
    with torch.no_grad():
        model.hidden_layer_1.covar_module.kernels[1].base_kernel.raw_lengthscale.fill_(float("nan"))
    step.step()
    with pytest.raises((NotPSDError, FloatingPointError)):
        step.check()


def test_C2_seeds_match_oracle():
    """BASELINE config 2 (d = 2, 2 fidelities, M = 128, N = 512, S = 8; SURVEY 8(d) inputs), seeds 0-2: ELBO, scaled KL and every
    gradient of the cooperative launch vs the oracle's dense evaluation.  128 inducing points in 2-D: cond(K_mm + 1e-6 I) ~ 1e9,
    either implementation carries ~cond * eps -- gates as test_hip_fullsize.py uses at C2 (north star: 1e-4)."""
    for seed in (0, 1, 2):
        cfg = dict(synthetic.CONFIGS["C2"])
        prob = synthetic.make_problem(d=cfg["d"], L=cfg["L"], M=cfg["M"], N=cfg["N"], S=cfg["S"], seed=seed)
        tc = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
        x, y, fid = tc(prob["x"]), tc(prob["y"]), tc(prob["fid"])
        eps = [None] + [tc(e) for e in prob["eps"][1:]]
        model = synthetic.model_from_problem(prob, num_samples_for_training=cfg["S"], device=DEV)
        raw = _raw_from_model(model, cfg["L"])
        e_o, skl_o = O.elbo(O.state_from_raw(raw), x, y, fid, eps=eps, S=cfg["S"])
        (-e_o).backward()
        step = _coop([model], [x], [y], [fid], [eps], want_grad=True)
        grads = step.gradients()[0]
        step.check()
        out = step.losses[0].cpu()
        assert rel(out[0], e_o) < 1e-7 and rel(out[1], skl_o) < 1e-7, (seed, out, e_o)
        for l in range(cfg["L"]):
            for key, tt in raw["layers"][l].items():
                p = _model_param_for(model, l, key)
                gref = tt.grad if key != "L_S" else torch.tril(tt.grad)
                assert rel(grads[p].reshape(gref.shape), gref) < 1e-4, (seed, l, key, rel(grads[p].reshape(gref.shape), gref))


# ------------------------------------------------------------------ conditioned training at the loop's later sizes
def _cond_setup(M, N, P=7, T=10, d=2, seed=0):
    from tests.helpers import oracle_state, to_t
    from tests.test_hip_conditioned import _fitter
    n_obj, n_con = 2, 2
    fitter, probs = _fitter(n_obj, n_con, N, M=M, d=d)
    fitter.thresholds_cons = torch.tensor([0.1, -0.05], dtype=torch.float64)
    g = torch.Generator().manual_seed(seed)
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
    return fitter, objs, cons, pareto_set, pareto_front, x_tilde, eps_all


@pytest.mark.parametrize("M,N,d", [(48, 56, 4), (64, 80, 5), (24, 40, 2)])
def test_coop_conditioned_iteration_matches_oracle(M, N, d):
    """CoopConditionedStep on 2 objectives + 2 constraints vs the oracle's joint loss (blackbox_mfdgp_fitter.py:270-343 restated):
    the loss and d loss / d (m, L_S) of every layer of every surrogate, explicit x~ and eps; then one iteration moves only m and
    L_S, by Adam's first step.  (d >= 4 for the larger M: the factor terms are Phi((threshold - mean) / sd) at points whose
    predictive variance k_nn - q sits at the jitter's scale when 48+ inducing points crowd [0,1]^2 -- there the float64 oracle
    and the kernel differ by cond(K_mm) * eps in sd.  N > M: with N = M the unshuffled batch IS the inducing set and the oracle,
    like GPyTorch, takes the equal-inputs shortcut for it, which the reordered rows of the step never do.)"""
    from mobocmf_amd.util.coop_step import CoopConditionedStep
    fitter, objs, cons, pareto_set, pareto_front, x_tilde, eps_all = _cond_setup(M, N, d=d)
    loss_o = O.conditioned_loss(objs, cons, pareto_set, pareto_front, x_tilde, fitter.thresholds_cons, fitter.eps)
    loss_o.backward()
    step = CoopConditionedStep(fitter, lr=1e-3, fixed_x_tilde=x_tilde.to(DEV), fixed_eps=eps_all, want_grad=True)
    grads = step.gradients()
    step.check()
    assert rel(step.loss, loss_o) < 1e-8, rel(step.loss, loss_o)
    for k, (rec, (tag, i, h)) in enumerate(zip(objs + cons, fitter._handlers())):
        for l in range(2):
            vd = getattr(h.mfdgp, f"hidden_layer_{l}").variational_strategy._variational_distribution
            assert rel(grads[k][vd.variational_mean], rec["state"]["layers"][l]["m"].grad) < 1e-5, (tag, i, l)
            assert rel(grads[k][vd.chol_variational_covar], torch.tril(rec["state"]["layers"][l]["L_S"].grad)) < 1e-5, (tag, i, l)
    before = [[p.detach().clone() for p in h.mfdgp.parameters()] for _, _, h in fitter._handlers()]
    step.step()
    step.check()
    for k, (_, _, h) in enumerate(fitter._handlers()):
        for p, p0 in zip(h.mfdgp.parameters(), before[k]):
            if p.requires_grad:
                gk = grads[k][p]
                moved = (p.detach() - p0)
                big = gk.abs() > 1e-9 * gk.abs().max()
                assert torch.allclose(moved[big], -1e-3 * torch.sign(gk[big]), rtol=1e-4, atol=0)
            else:
                assert torch.equal(p.detach(), p0)


def test_coop_one_launch_conditioned_iteration_equals_the_three_launch_form():
    """Mode 4 (the whole grid meets once after the forward, every surrogate's first workgroup forms its factor gradients) == the
    forward-only launch + mobocmf_cond_factors_forward launches + step launch: losses, factor terms, parameters after 3 iterations."""
    from mobocmf_amd.util.coop_step import CoopConditionedStep
    runs = {}
    for one in (True, False):
        fitter, *_rest = _cond_setup(48, 48, d=4, seed=3)
        step = CoopConditionedStep(fitter, lr=1e-3)
        step.xrng.copy_(torch.tensor([1234567, 0], dtype=torch.int64))
        for layer_owner in step.models:
            for j, layer in enumerate(layer_owner._layers()):
                layer._rng(torch.device(DEV, torch.cuda.current_device())).copy_(torch.tensor([99 + j, 0, 0], dtype=torch.int64))
        step.one_launch = one
        hist = []
        for _ in range(3):
            step.step()
            step.check()
            hist.append((step.losses.clone(), step.factor_losses.clone()))
        assert step.one_launch is one
        runs[one] = (hist, [p.detach().clone() for _, _, h in fitter._handlers() for p in h.mfdgp.parameters()])
    for (la, fa), (lb, fb) in zip(runs[True][0], runs[False][0]):
        assert rel(la, lb) < 1e-10 and rel(fa, fb) < 1e-10, (rel(la, lb), rel(fa, fb))
    for pa, pb in zip(runs[True][1], runs[False][1]):
        assert rel(pa, pb) < 1e-9


def test_fitter_trains_mid_size_surrogates_through_the_cooperative_launch():
    """BlackBoxMFDGPFitter.train_mfdgps / train_conditioned_mfdgps at M = N = 48 (iteration ~33 of the reference's own loop,
    toy_synthetic_2D_JESMOCMF.py:305-331): the one-workgroup kernel does not take the size, the cooperative launch does -- the
    loss decreases, nothing falls back to the layer path."""
    from mobocmf_amd.util import coop_step, tiny_step
    from tests.test_hip_conditioned import _fitter
    fitter, _ = _fitter(2, 1, 48, M=48)
    hs = fitter._handlers()
    x, _, fid = hs[0][2].train_dataset.tensors
    assert not tiny_step.eligible(hs[0][2].mfdgp, x, fid) and coop_step.eligible(hs[0][2].mfdgp, x, fid)
    done, step = fitter._train_mfdgp_tiny(False, 30, 1e-3)
    assert done == 30 and isinstance(step, coop_step.CoopELBOStep)
    l0 = step.losses[:, 2].clone()
    done, step = fitter._train_mfdgp_tiny(False, 200, 3e-3)
    assert done == 200 and bool((step.losses[:, 2] < l0).all())
    g = torch.Generator().manual_seed(1)
    fitter.set_pareto_solution(torch.rand(5, 2, dtype=torch.float64, generator=g), torch.randn(5, 2, dtype=torch.float64, generator=g) * 0.3)
    for _, _, h in hs:
        h.mfdgp.fix_variational_hypers_cond(True)
    done, cstep = fitter._train_conditioned_tiny(50)
    assert done == 50 and isinstance(cstep, coop_step.CoopConditionedStep) and bool(torch.isfinite(cstep.loss))


PRED = [dict(M=64, N=64, d=2, S=5, f=1, T=9), dict(M=48, N=60, d=3, S=4, f=0, T=20), dict(M=100, N=100, d=2, S=6, f=1, T=33),
        dict(M=128, N=128, d=2, S=8, f=1, T=40), dict(M=75, N=75, d=8, S=3, f=1, T=17), dict(M=24, N=30, d=2, S=5, f=1, T=7)]


@pytest.mark.parametrize("c", PRED, ids=["M%d_d%d_S%d_f%d_T%d" % (c["M"], c["d"], c["S"], c["f"], c["T"]) for c in PRED])
def test_coop_predict_group_matches_predict_for_acquisition_and_its_input_gradient(c):
    """The acquisition search at the reference's later loop sizes (JESMOC_MFDGP.py:137-184 against M = N = 33 ... 75 surrogates):
    CoopPredictGroup -- mode 2 of the cooperative launch for the moments of all models, mode 3 for d/dX -- against
    MFDGP.predict_for_acquisition through the layer entry points (pinned to the oracle in test_hip_model.py), T test points,
    S fixed samples.  Inputs of <= 3 dimensions with M >= 48 give cond(K_mm + 1e-6 I) ~ 1e9: either side carries cond * eps."""
    from mobocmf_amd.util.coop_step import CoopPredictGroup, fits_predict
    from tests.test_hip_model import build_model
    M, N, d, S, fidelity, T = c["M"], c["N"], c["d"], c["S"], c["f"], c["T"]
    models = [build_model(synthetic.make_problem(d=d, L=2, M=M, N=N, S=S, seed=s), S_train=S, S_acq=S) for s in (1, 2, 3)]
    assert all(fits_predict(m, fidelity, T, d) for m in models)
    g = torch.Generator().manual_seed(3)
    X = torch.rand(T, d, dtype=torch.float64, generator=g).to(DEV)
    wm = torch.randn(len(models), T, dtype=torch.float64, generator=g).to(DEV)
    wv = torch.randn(len(models), T, dtype=torch.float64, generator=g).to(DEV)
    Xa = X.clone().requires_grad_(True)
    ref_m, ref_v = [], []
    for m in models:
        m.eval()
        mu, v = m.predict_for_acquisition(Xa, fidelity)
        m.train()
        ref_m.append(mu), ref_v.append(v)
    ref_m, ref_v = torch.stack(ref_m), torch.stack(ref_v)
    ((ref_m * wm).sum() + (ref_v * wv).sum()).backward()
    grp = CoopPredictGroup(models, fidelity, T, d)
    Xb = X.clone().requires_grad_(True)
    mus, v = grp.acquisition_moments(Xb)
    ((mus * wm).sum() + (v * wv).sum()).backward()
    hard = d <= 3 and M >= 48
    assert rel(mus, ref_m) < (1e-6 if hard else 1e-8) and rel(v, ref_v) < (1e-5 if hard else 1e-7), (rel(mus, ref_m), rel(v, ref_v))
    assert rel(Xb.grad, Xa.grad) < (1e-4 if hard else 1e-6), rel(Xb.grad, Xa.grad)
    # a second evaluation at other points reuses the group (descriptors, workspace, arrival counters)
    X2 = torch.rand(T, d, dtype=torch.float64, generator=g).to(DEV)
    m2, v2 = grp.acquisition_moments(X2)
    with torch.no_grad():
        for i, m in enumerate(models):
            m.eval()
            mu, vv = m.predict_for_acquisition(X2, fidelity)
            m.train()
            assert rel(m2[i], mu) < (1e-6 if hard else 1e-8) and rel(v2[i], vv) < (1e-5 if hard else 1e-7)
    # constant parameters for a whole search (JESMOC_MFDGP._optimize): the chains are formed by the first launch after freeze()
    # only (MOBOCMF_STEP_CHAIN_VALID) -- bit for bit the same moments and gradients
    grp.freeze()
    for Xq in (X2, X):
        Xf = Xq.clone().requires_grad_(True)
        mf, vf = grp.acquisition_moments(Xf)
        ((mf * wm).sum() + (vf * wv).sum()).backward()
        if Xq is X:
            assert grp._chain_ready and torch.equal(mf, mus.detach()) and torch.equal(vf, v.detach()) and torch.equal(Xf.grad, Xb.grad)
        else:
            assert torch.equal(mf, m2) and torch.equal(vf, v2)
    grp.thaw()
    assert not grp._frozen and not grp._chain_ready
    if M <= 32:      # inside the one-workgroup kernel's window both kernels answer: same numbers
        from mobocmf_amd.util.tiny_step import TinyPredictGroup
        tg = TinyPredictGroup(models, fidelity, T, d)
        Xc = X.clone().requires_grad_(True)
        mt, vt = tg.acquisition_moments(Xc)
        ((mt * wm).sum() + (vt * wv).sum()).backward()
        assert rel(mus, mt) < 1e-8 and rel(v, vt) < 1e-7 and rel(Xb.grad, Xc.grad) < 1e-6


def test_coupled_jes_at_mid_sizes_goes_through_the_cooperative_launch():
    """JESMOC_MFDGP.coupled_acq (JESMOC_MFDGP.py:38-52,125-135) for surrogates of M = N = 40 (iteration ~25 of the reference's
    loop): value and candidate gradient through CoopPredictGroup (two launches for all six models) equal the layer path's."""
    from mobocmf_amd.acquisition_functions.JESMOC_MFDGP import JESMOC_MFDGP
    from mobocmf_amd.util.coop_step import CoopPredictGroup
    from tests.test_hip_conditioned import _fitter
    fu, _ = _fitter(2, 1, 40, M=40)
    fc, _ = _fitter(2, 1, 40, M=40)
    g = torch.Generator().manual_seed(1)
    for _, _, h in fc._handlers():
        for layer in h.mfdgp._layers():
            vd = layer.variational_strategy._variational_distribution
            with torch.no_grad():
                vd.variational_mean.add_(0.05 * torch.randn(vd.variational_mean.shape, dtype=torch.float64, generator=g).to(DEV))
                vd.chol_variational_covar.mul_(0.7)
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
    X = torch.rand(11, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(9)).to(DEV)
    for f in (0, 1):
        res = {}
        for one_launch in (True, False):
            acq.use_tiny_step = one_launch
            Xg = X.clone().requires_grad_(True)
            v = acq.coupled_acq(Xg, fidelity=f)
            v.sum().backward()
            res[one_launch] = (v.detach(), Xg.grad)
        assert isinstance(acq._tiny_groups[(f, 11, 2)], CoopPredictGroup)
        assert float(res[False][0].abs().max()) > 0
        assert rel(res[True][0], res[False][0]) < 1e-6, (f, rel(res[True][0], res[False][0]))
        assert rel(res[True][1], res[False][1]) < 1e-4, (f, rel(res[True][1], res[False][1]))
