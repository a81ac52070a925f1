"""JES acquisition on MFDGP surrogates -- host mirror of mobocmf/acquisition_functions/JESMOC_MFDGP.py
(``_JES_MFDGP.forward`` :38-52, ``JESMOC_MFDGP`` :55-184).

The per-black-box value 0.5 * clamp(log v_uncond - log v_cond, 0) runs on the HIP path (predict_for_acquisition of
both models + functional.jes) and is differentiable w.r.t. X.  botorch's ``optimize_acqf`` (absent here, SURVEY row
N3) is replaced by ``optimize_acqf_multistart``: the same recipe -- ``raw_samples`` uniform candidates, the best
``num_restarts`` refined by projected gradient ascent (Adam) for ``maxiter`` steps, ALL restarts in one batch so every
iteration is a single pair of model evaluations.  With surrogates sharded over ranks the coupled acquisition is the
all-gather + sum of mobocmf_amd.parallel.coupled_acquisition.
"""
import contextlib

import torch

from .. import functional as F
from .. import parallel


class _JES_MFDGP:

    def __init__(self, fidelity, mfdgp_uncond, mfdgp_cond, model=None):
        assert model is None
        self.fidelity = fidelity
        self.mfdgp_uncond = mfdgp_uncond
        self.mfdgp_cond = mfdgp_cond

    def forward(self, X):
        """Evaluate JES at X (T, d) or (T, 1, d).  Both models are switched to ``.eval()`` (full predictive branch,
        no clamp) exactly as the reference does (:42-50)."""
        self.mfdgp_uncond.eval()
        _, v_u = self.mfdgp_uncond.predict_for_acquisition(X, self.fidelity)
        self.mfdgp_uncond.train()
        self.mfdgp_cond.eval()
        _, v_c = self.mfdgp_cond.predict_for_acquisition(X, self.fidelity)
        self.mfdgp_cond.train()
        return F.jes(v_u, v_c)

    __call__ = forward

    @contextlib.contextmanager
    def frozen(self):
        """Both models' parameters are constants while the acquisition is optimised: their M x M chains are computed
        once (MFDGP.frozen_chains) instead of at each of the ~400 evaluations."""
        with self.mfdgp_uncond.frozen_chains(), self.mfdgp_cond.frozen_chains():
            yield self


def optimize_acqf_multistart(acq_function, bounds, num_restarts=5, raw_samples=200, maxiter=200, lr=0.02,
                             generator=None):
    """Maximise ``acq_function`` over the box ``bounds`` (2, d).  Returns (candidate (1, d), value)."""
    lo, hi = bounds[0], bounds[1]
    d = lo.numel()
    dev, dt = lo.device, lo.dtype
    with torch.no_grad():
        Xraw = lo + (hi - lo) * torch.rand(raw_samples, d, dtype=dt, device=dev, generator=generator)
        parallel.broadcast_(Xraw)     # sharded surrogates: every rank scores (and all-gathers values of) the same points
        vals = acq_function(Xraw)
        X = Xraw[torch.topk(vals, min(num_restarts, raw_samples)).indices].clone()
    X.requires_grad_(True)
    if X.is_cuda and X.dtype == torch.float64:
        # the library's one-launch Adam (torch.optim.Adam's update; also spares the process the ~0.6 s of lazy imports that
        # the first torch.optim step pulls in -- more than a whole search at the reference's sizes)
        opt = F.FusedAdam([X], lr=lr * float((hi - lo).mean()))
    else:
        opt = torch.optim.Adam([X], lr=lr * float((hi - lo).mean()))
    # every iterate X_0 ... X_maxiter is scored ONCE: the value of the evaluation that also yields its gradient is the score of the
    # iterate (round 4 scored X_{t+1} after the step and evaluated it again, with gradient, at the top of the next iteration --
    # a third of the search's launches)
    best_x, best_v = X.detach().clone(), None
    for it in range(maxiter + 1):
        last = it == maxiter
        opt.zero_grad()
        if last:
            with torch.no_grad():
                v = acq_function(X)
        else:
            v = acq_function(X)
        with torch.no_grad():
            vd = v.detach()
            if best_v is None:
                best_v = vd.clone()
            else:
                better = vd > best_v
                best_v = torch.where(better, vd, best_v)
                best_x[better] = X.detach()[better]
        if last:
            break
        (-v.sum()).backward()
        opt.step()
        with torch.no_grad():
            X.clamp_(min=lo, max=hi)
            parallel.broadcast_(X)    # (no-op on one rank) summation order may differ by an ulp between ranks
    k = int(torch.argmax(best_v))
    return best_x[k:k + 1].detach(), best_v[k].detach()


class JESMOC_MFDGP:

    def __init__(self, model, num_fidelities=1, model_cond=None, standard_bounds=None, eval_highest_fidelity=False):
        self.standard_bounds = standard_bounds
        self.eval_highest_fidelity = eval_highest_fidelity
        self.blackbox_mfdgp_fitter_uncond = model.copy_uncond()
        if model_cond is None:
            # reference (:64-66): sample a Pareto solution (RFF posterior samples + MOOP) unless one was provided
            if getattr(model, "pareto_set", None) is None:
                model.sample_and_store_pareto_solution()
            self.pareto_set, self.pareto_front = model.pareto_set, model.pareto_front
            self.samples_objs, self.samples_cons = getattr(model, "samples_objs", None), getattr(model, "samples_cons", None)
            model.train_conditioned_mfdgps()
            self.blackbox_mfdgp_fitter_cond = model
        else:
            self.pareto_set, self.pareto_front = model_cond.pareto_set, model_cond.pareto_front
            self.blackbox_mfdgp_fitter_cond = model_cond
        self.num_fidelities = num_fidelities
        self.objectives, self.constraints, self.costs_blackboxes = {}, {}, {}
        for n_f in range(num_fidelities):
            self.objectives[n_f] = {}
            self.constraints[n_f] = {}
            self.costs_blackboxes[n_f] = {"total": 0.0}

    def add_blackbox(self, fidelity, blackbox_name, cost_evaluation=1.0, is_constraint=False):
        mfdgp_uncond = self.blackbox_mfdgp_fitter_uncond.get_model(blackbox_name, is_constraint=is_constraint)
        mfdgp_cond = self.blackbox_mfdgp_fitter_cond.get_model(blackbox_name, is_constraint=is_constraint)
        jes_mfdgp = _JES_MFDGP(fidelity, mfdgp_uncond, mfdgp_cond)
        (self.constraints if is_constraint else self.objectives)[fidelity][blackbox_name] = jes_mfdgp
        self.costs_blackboxes[fidelity]["total"] += cost_evaluation
        self.costs_blackboxes[fidelity][blackbox_name] = cost_evaluation
        return jes_mfdgp

    def decoupled_acq(self, X, fidelity, blackbox_name, is_constraint=True):
        d = self.constraints if is_constraint else self.objectives
        return d[fidelity][blackbox_name](X.double())

    use_tiny_step = True      # False: always the layer path (A/B, tests)

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_tiny_groups", None)      # device descriptors: rebuilt on first use
        return state

    def _tiny_group(self, jess, fidelity, T, d):
        """TinyPredictGroup (M <= 32) or CoopPredictGroup (M <= 128) over (uncond, cond) of every black-box of ``fidelity`` for T
        test points -- when all of them fit a one-launch kernel (util/tiny_step.py, util/coop_step.py) -- else None.  Built once
        per (fidelity, T)."""
        cache = self.__dict__.setdefault("_tiny_groups", {})
        key = (fidelity, T, d)
        if key not in cache:
            from ..util import tiny_step as TS
            models = [m for jes in jess for m in (jes.mfdgp_uncond, jes.mfdgp_cond)]
            on_gpu = bool(models) and all(p.is_cuda for p in models[0].parameters())
            if on_gpu and all(TS.fits_predict(m, fidelity, T, d) for m in models):
                cache[key] = TS.TinyPredictGroup(models, fidelity, T, d)
            else:      # mid-size surrogates (M <= 128): the cooperative launch, several workgroups per model
                from ..util import coop_step as CS
                ok = on_gpu and all(CS.fits_predict(m, fidelity, T, d) for m in models)
                cache[key] = CS.CoopPredictGroup(models, fidelity, T, d) if ok else None
        return cache[key]

    def coupled_acq(self, X, fidelity):
        """Sum over all black-boxes (:125-135).  Sharded surrogates: each rank adds its own, one all-gather sums.
        Small surrogates (the reference's own sizes): the predictive moments of ALL models -- unconditioned and conditioned,
        every black-box -- come from one launch, their gradient w.r.t. X from one more (TinyPredictGroup); the JES value
        0.5 clamp(log v_uncond - log v_cond, 0) (:38-52) is then a handful of element-wise operations over all of them."""
        X = X.double()
        jess = list(self.objectives[fidelity].values()) + list(self.constraints[fidelity].values())
        if self.use_tiny_step and jess and X.is_cuda and parallel.world()[1] == 1:
            X2 = X[:, 0, :] if X.dim() > 2 else X
            grp = self._tiny_group(jess, fidelity, X2.shape[0], X2.shape[1])
            if grp is not None:
                if self.__dict__.get("_search_running") and hasattr(grp, "freeze"):
                    grp.freeze()      # (constant parameters for the whole search: the chains are formed by its first launch only)
                _, v = grp.acquisition_moments(X2)
                return (0.5 * torch.clamp(torch.log(v[0::2]) - torch.log(v[1::2]), min=0.0)).sum(0)
        local = [obj(X) for obj in self.objectives[fidelity].values()] + \
                [con(X) for con in self.constraints[fidelity].values()]
        if not local:
            local = [torch.zeros(X.shape[0], dtype=X.dtype, device=X.device)]
        acq = torch.stack(local).sum(0)
        _, w = parallel.world()
        if w > 1:       # every rank enters the exchange, also one that holds no black-box of this fidelity
            acq = acq + (parallel.coupled_acquisition(acq.detach()[None]) - acq.detach())
        return acq

    @contextlib.contextmanager
    def _frozen_groups(self):
        """The one-launch predict groups used inside keep their chains (util/coop_step.py CoopPredictGroup.freeze)."""
        self._search_running = True
        try:
            yield
        finally:
            self._search_running = False
            for grp in self.__dict__.get("_tiny_groups", {}).values():
                if grp is not None and hasattr(grp, "thaw"):
                    grp.thaw()

    def _optimize(self, fidelity, **kw):
        with contextlib.ExitStack() as stack:       # fitted models: freeze every surrogate's chain for the whole search
            for jes in list(self.objectives[fidelity].values()) + list(self.constraints[fidelity].values()):
                stack.enter_context(jes.frozen())
            stack.enter_context(self._frozen_groups())
            return optimize_acqf_multistart(lambda x: self.coupled_acq(x, fidelity=fidelity), self.standard_bounds,
                                            num_restarts=5, raw_samples=200, maxiter=kw.get("maxiter", 200))

    def _get_nextpoint_coupled_highest_fidelity(self, iteration=None, verbose=False, maxiter=200):
        """Reference name (:137-149): search the highest fidelity only."""
        keep, self.eval_highest_fidelity = self.eval_highest_fidelity, True
        try:
            return self.get_nextpoint_coupled(iteration=iteration, verbose=verbose, maxiter=maxiter)
        finally:
            self.eval_highest_fidelity = keep

    def _get_nextpoint_coupled(self, iteration=None, verbose=False, maxiter=200):
        """Reference name (:151-176): search every fidelity, pick the best cost-weighted value."""
        keep, self.eval_highest_fidelity = self.eval_highest_fidelity, False
        try:
            return self.get_nextpoint_coupled(iteration=iteration, verbose=verbose, maxiter=maxiter)
        finally:
            self.eval_highest_fidelity = keep

    def get_nextpoint_coupled(self, iteration=None, verbose=False, maxiter=200):
        """Next point + fidelity by cost-weighted acquisition (:137-184)."""
        fids = [self.num_fidelities - 1] if self.eval_highest_fidelity else list(range(self.num_fidelities))
        best = None
        for fidelity in fids:
            cand, val = self._optimize(fidelity, maxiter=maxiter)
            cost = self.costs_blackboxes[0 if self.eval_highest_fidelity else fidelity]["total"]
            w = val / cost
            if best is None or best[0] < w:
                best = (w, cand, fidelity)
        w, cand, fidelity = best
        if verbose:
            print("Iter:", iteration, "Acquisition:", float(w * self.costs_blackboxes[fidelity]["total"]),
                  " Evaluating fidelity", fidelity, "at", cand[0].cpu().numpy())
        return cand[0, :], fidelity
