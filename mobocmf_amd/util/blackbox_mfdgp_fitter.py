"""Trainer/orchestrator -- host mirror of mobocmf/util/blackbox_mfdgp_fitter.py (unconditioned training:
``MFDGPHandler`` :22-39, ``BlackBoxMFDGPFitter.__init__`` :43-81, ``initialize_mfdgp`` :84-115,
``_train_mfdgp`` :117-152, ``train_mfdgps`` :154-178; the ELBO step is :161-171).

Conditioned training (:245-354) and Pareto sampling of posterior function samples (:181-225, row N2)
are mirrored below over the same HIP path.

MI355X additions: ``device`` (models and data live on the GPU), ``num_inducing`` / ``num_samples_for_training``
pass-through, and surrogate sharding over ranks (one process per GPU, see mobocmf_amd.parallel).

Which GPyTorch branch the training batch takes: the reference trains on ``DataLoader(..., shuffle=True)`` (:35), so with
its default Z = x_train the full batch is a random PERMUTATION of Z and ``torch.equal(x, Z)`` is false: layer 0 goes through
the general path (mu = K_nm (K_mm + eps I)^-1 m, not m itself).  Both trainers here do the same: the eager one iterates the
shuffling loader, the HIP-graph one captures the step on a fixed non-identity permutation of the rows (the full-batch
loss does not depend on the row order).
"""
import sys
import warnings
from copy import deepcopy

import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset

from ..mlls.variational_elbo_mf import VariationalELBOMF
from ..models.mfdgp import MFDGP, TL

ITER_PRINT = 1000


def _prod(t, dim):
    """Product over a (short) dimension as a chain of multiplications: torch.prod's backward counts zeros on the host
    (.item()), which a stream capture forbids."""
    parts = t.unbind(dim)
    out = parts[0]
    for p in parts[1:]:
        out = out * p
    return out


def _ncdf(z):
    """Standard normal cdf (the reference uses torch.distributions Normal(0, 1).cdf, :18)."""
    return 0.5 * (1.0 + torch.erf(z * 0.7071067811865476))


class MFDGPHandler:

    MAX_TRIES_FOR_FEASIBLE_GRID = 50

    def __init__(self, x_train, y_train, fidelities_train, num_fidelities, batch_size, type_lengthscale,
                 previously_trained_model=None, init_params_to_prior_and_fix_them=False,
                 use_only_highest_fidelity=False, device="cuda", **model_kwargs):
        self.mfdgp = MFDGP(x_train, y_train, fidelities_train, num_fidelities=num_fidelities,
                           type_lengthscale=type_lengthscale, previously_trained_model=previously_trained_model,
                           use_only_highest_fidelity=use_only_highest_fidelity,
                           init_params_to_prior_and_fix_them=init_params_to_prior_and_fix_them, **model_kwargs)
        self.mfdgp.double()  # float64 end to end, as the reference (:32)
        self.mfdgp.to(device)
        self.elbo = VariationalELBOMF(self.mfdgp, x_train.shape[-2], num_fidelities=num_fidelities)
        dev = torch.device(device)
        self.train_dataset = TensorDataset(x_train.double().to(dev), y_train.double().to(dev),
                                           fidelities_train.double().to(dev))
        self.batch_size = batch_size
        self.train_loader = DataLoader(self.train_dataset, batch_size=batch_size, shuffle=True)
        self.iter_train_loader = None
        self.num_data = x_train.shape[0]
        self.num_fidelities = num_fidelities
        self.global_index = None       # position among ALL objectives (or constraints) when surrogates are sharded over ranks


class BlackBoxMFDGPFitter:

    def __init__(self, num_fidelities, batch_size, lr_1=0.003, lr_2=0.001, num_epochs_1=5000, num_epochs_2=15000,
                 pareto_set_size=50, opt_grid_size=1000, eps=1e-8, decoupled_evals=False,
                 type_lengthscale=TL.MEDIAN, device="cuda", **model_kwargs):
        self.num_obj = 0
        self.num_con = 0
        self.models_uncond_trained = False
        self.mfdgp_handlers_objs = {}
        self.mfdgp_handlers_cons = {}
        self.thresholds_cons = torch.tensor([], dtype=torch.double)
        self.x_train = None
        self.objs_train = torch.tensor([], dtype=torch.double)
        self.cons_train = torch.tensor([], dtype=torch.double)
        self.num_fidelities = num_fidelities
        self.batch_size = batch_size
        self.points_to_sample = batch_size
        self.lr_1, self.lr_2 = lr_1, lr_2
        self.num_epochs_1, self.num_epochs_2 = num_epochs_1, num_epochs_2
        self.pareto_set_size = pareto_set_size
        self.opt_grid_size = opt_grid_size
        self.eps = eps
        self.decoupled_evals = decoupled_evals
        self.type_lengthscale = type_lengthscale
        self.device = device
        self.model_kwargs = model_kwargs
        self.verbose = True
        self.thresholds_cons_global = None      # sharded surrogates: thresholds of ALL constraints, global order

    def set_global_constraint_thresholds(self, thresholds):
        """Surrogates sharded over ranks (mobocmf_amd.parallel): the omega factors (:235-243) see every constraint of the
        problem, so each rank needs the whole threshold vector in global constraint order (``global_index`` of
        ``initialize_mfdgp``).  Single process: not needed (the local vector is the whole one)."""
        self.thresholds_cons_global = torch.as_tensor(thresholds, dtype=torch.double).reshape(-1)
        self._thr_cache = None

    def initialize_mfdgp(self, x_train, y_train, fidelities, blackbox_name, threshold_constraint=0.0,
                         is_constraint=False, previously_trained_model=None,
                         init_params_to_prior_and_fix_them=False, use_only_highest_fidelity=False, global_index=None):
        """``global_index`` (sharded surrogates only): this black-box's position among ALL objectives -- the column of the
        Pareto front it is conditioned on -- or among ALL constraints; default: its position on this rank."""
        if self.x_train is None:
            self.x_train = x_train
        else:
            assert torch.equal(self.x_train, x_train), "The inputs for this new mfdgp do not match with inputs for " \
                "previous mfdgp models. This class is not currently prepared for a decoupled evaluation setting."
        handler = MFDGPHandler(x_train, y_train, fidelities, self.num_fidelities, self.batch_size,
                               type_lengthscale=self.type_lengthscale,
                               previously_trained_model=previously_trained_model,
                               init_params_to_prior_and_fix_them=init_params_to_prior_and_fix_them,
                               use_only_highest_fidelity=use_only_highest_fidelity, device=self.device,
                               **self.model_kwargs)
        handler.global_index = global_index
        if is_constraint:
            self.cons_train = torch.cat((self.cons_train, y_train.cpu().double()), 1)
            self.mfdgp_handlers_cons[blackbox_name] = handler
            self.thresholds_cons = torch.cat((self.thresholds_cons, torch.tensor([threshold_constraint]).double()), 0)
            self.num_con += 1
        else:
            self.objs_train = torch.cat((self.objs_train, y_train.cpu().double()), 1)
            self.mfdgp_handlers_objs[blackbox_name] = handler
            self.num_obj += 1

    def _handlers(self):
        return [("OBJ", n, h) for n, h in enumerate(self.mfdgp_handlers_objs.values())] + \
               [("CON", n, h) for n, h in enumerate(self.mfdgp_handlers_cons.values())]

    def get_model(self, blackbox_name, is_constraint=False):
        d = self.mfdgp_handlers_cons if is_constraint else self.mfdgp_handlers_objs
        return d[blackbox_name].mfdgp

    def _train_mfdgp(self, func_update_model, fix_variational_hypers, num_epochs, lr):
        opts = []
        for _, _, h in self._handlers():
            h.mfdgp.fix_variational_hypers(fix_variational_hypers)
            from ..functional import FusedAdam
            opts.append(FusedAdam(list(h.mfdgp.parameters()), lr=lr))      # torch.optim.Adam's update in one launch
        for (tag, n, h), optimizer in zip(self._handlers(), opts):
            for i in range(num_epochs):
                loss_iter, kl_iter = func_update_model(h.mfdgp, h.elbo, optimizer, h.train_loader)
                if self.verbose and ((i % ITER_PRINT) == 0 or (i + 1) == num_epochs):
                    print("[%s: " % tag, n, "] Epoch:", i, "/", num_epochs, ". Avg. Neg. ELBO per epoch:",
                          loss_iter.item(), "\t KL per epoch:", kl_iter.item())
                    sys.stdout.flush()

    @staticmethod
    def update_model(model, elbo, optimizer, train_loader):
        """One epoch = the reference's ``_update_model`` closure (:156-173); one ELBO step per batch."""
        loss_iter = 0.0
        kl_iter = 0.0
        for (x_batch, y_batch, fidelities) in train_loader:
            optimizer.zero_grad()
            output = model(x_batch)
            res = elbo(output, y_batch.T, fidelities)
            loss, kl = -res[0], res[1]
            loss.backward()
            optimizer.step()
            loss_iter += loss.detach()
            kl_iter += kl.detach()
        return loss_iter, kl_iter

    def _stream_for(self, slot, device):
        """One HIP stream per surrogate slot, created once per fitter and reused by every training phase.  Replaying a
        captured step on a stream created LATER in the process is ~1.5x slower at the sizes where the step is bound by
        graph-node dispatch (0.37 vs 0.55 ms per Forrester step, measured: the first streams get hardware queues of their
        own), and torch hands stream handles out of a small pool anyway."""
        pool = self.__dict__.setdefault("_step_streams", {})
        key = (str(device), slot)
        if key not in pool:
            pool[key] = torch.cuda.Stream(device=device)
        return pool[key]

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_step_streams", None)       # streams are process-local: never copied / pickled (copy_uncond, dill)
        state.pop("_thr_cache", None)
        return state

    @staticmethod
    def shuffled_rows(n, device):
        """One draw of the loader's shuffle (:35), never the identity for n > 1: the row order the captured step keeps.
        (An identity draw -- probability 1/n! in the reference -- would send layer 0 through GPyTorch's equal-inputs
        shortcut; the general branch is what the reference's training executes.)"""
        perm = torch.randperm(n)
        if n > 1 and bool((perm == torch.arange(n)).all()):
            perm = torch.roll(perm, 1)
        return perm.to(device)

    def _train_mfdgp_graphed(self, fix_variational_hypers, num_epochs, lr):
        """Full-batch fast path: every surrogate's ELBO step is captured into a HIP graph (mobocmf_amd.util.graphed_step)
        and the independent surrogates advance in lockstep on separate streams (the reference loops over them one
        after the other, :134-152; they share nothing, so the result only differs in which N(0,1) draws each gets).
        The rows are shuffled once (see the module docstring): the step runs GPyTorch's general branch, as the
        reference's shuffled batches do."""
        from .graphed_step import GraphedELBOStep
        # the reference's own sizes (M = N = tens of points): every surrogate's whole step in ONE launch per epoch
        done, tiny = self._train_mfdgp_tiny(fix_variational_hypers, num_epochs, lr)
        if done >= num_epochs:
            return
        num_epochs -= done
        steps = []
        for slot, (tag, n, h) in enumerate(self._handlers()):
            h.mfdgp.fix_variational_hypers(fix_variational_hypers)
            x, y, fid = h.train_dataset.tensors
            perm = self.shuffled_rows(x.shape[0], x.device)
            steps.append((tag, n, GraphedELBOStep(h.mfdgp, h.elbo, x[perm].contiguous(), y[perm].contiguous(),
                                                  fid[perm].contiguous(), lr=lr, stream=self._stream_for(slot, x.device))))
            if tiny is not None:      # a Cholesky failed in the one-launch step: this path (jitter ladder) takes over its state
                tiny.export_adam_state(slot, steps[-1][2].optimizer)
        from ..layers.mfdgp_hidden_layer import NotPSDError
        for _, _, g in steps:
            g.snapshot()
            g.last_good = -1                # last epoch whose state was verified
        for i in range(num_epochs):
            for _, _, g in steps:
                g.step()
            if (i % ITER_PRINT) == 0 or (i + 1) == num_epochs:
                for tag, n, g in steps:
                    try:
                        g.check()
                        g.snapshot()
                        g.last_good = i
                    except (NotPSDError, FloatingPointError) as err:
                        # a replayed graph cannot retry with more jitter: roll back to the last verified state and redo
                        # the epochs since then eagerly (per-step jitter ladder, as the reference); the surrogate then
                        # stays eager for the rest of the phase
                        redo = i - g.last_good
                        warnings.warn("%s %d: %s -- rolling back %d epochs and redoing them eagerly" % (tag, n, err, redo))
                        g.restore_and_go_eager()
                        for _ in range(redo):
                            g.step()
                        g.check()
                        g.snapshot()
                        g.last_good = i
                    if self.verbose:
                        print("[%s: " % tag, n, "] Epoch:", i, "/", num_epochs, ". Avg. Neg. ELBO per epoch:",
                              g.loss.item(), "\t KL per epoch:", g.kl.item())
                        sys.stdout.flush()
        for _, _, g in steps:
            g.stream.synchronize()
            g.model.set_check_pd(True)
            g.retire()

    use_tiny_step = True      # False: always the layer path (A/B, tests)

    def _train_mfdgp_tiny(self, fix_variational_hypers, num_epochs, lr):
        """The training phase through mobocmf_tiny_elbo_step (util/tiny_step.py) when EVERY surrogate fits it: one launch
        per epoch for all of them.  Returns (epochs completed, step object or None): fewer than ``num_epochs`` when the
        surrogates do not fit (0, None) or when a Cholesky failed -- the state is then rolled back to the last verified epoch
        and the layer path, which can retry with more jitter as the reference does at every step, continues from there."""
        from . import coop_step
        from .tiny_step import TinyELBOStep, eligible
        from ..layers.mfdgp_hidden_layer import NotPSDError
        hs = self._handlers()
        if not self.use_tiny_step or not hs:
            return 0, None
        for _, _, h in hs:
            h.mfdgp.fix_variational_hypers(fix_variational_hypers)
        data = [h.train_dataset.tensors for _, _, h in hs]
        # M <= 32 and narrow panels: one workgroup per surrogate (csrc/tiny_step.hip); up to M = 128: several workgroups per
        # surrogate, MFMA products (csrc/coop_step.hip); beyond that, or for wide panels, the layer path
        if all(x.is_cuda and eligible(h.mfdgp, x, fid) for (_, _, h), (x, _, fid) in zip(hs, data)):
            cls = TinyELBOStep
        elif all(x.is_cuda and coop_step.worthwhile(h.mfdgp, x, fid) for (_, _, h), (x, _, fid) in zip(hs, data)):
            cls = coop_step.CoopELBOStep
        else:
            return 0, None
        dev = data[0][0].device
        step = cls([h.mfdgp for _, _, h in hs], [h.num_data for _, _, h in hs], [t[0] for t in data],
                   [t[1] for t in data], [t[2] for t in data], lr=lr, stream=self._stream_for(0, dev))
        step.stream.wait_stream(torch.cuda.current_stream(dev))
        step.snapshot()
        last_good = -1
        for i in range(num_epochs):
            step.step()
            if (i % ITER_PRINT) == 0 or (i + 1) == num_epochs:
                try:
                    step.check()
                except (NotPSDError, FloatingPointError) as err:
                    warnings.warn("%s -- rolling back %d epochs; the layer path continues" % (err, i - last_good))
                    step.restore()
                    torch.cuda.current_stream(dev).wait_stream(step.stream)
                    return last_good + 1, step
                step.snapshot()
                last_good = i
                if self.verbose:
                    out = step.losses.cpu()
                    for k, (tag, n, _) in enumerate(hs):
                        print("[%s: " % tag, n, "] Epoch:", i, "/", num_epochs, ". Avg. Neg. ELBO per epoch:",
                              out[k, 2].item(), "\t KL per epoch:", out[k, 1].item())
                    sys.stdout.flush()
        step.stream.synchronize()
        return num_epochs, step

    def train_mfdgps(self, use_graphs=None):
        """2-phase Adam schedule of the reference (:175-176).  ``use_graphs`` (default: automatically when every
        handler trains on the full batch, as all the reference's examples do) selects the HIP-graph fast path."""
        full_batch = all(h.batch_size >= h.num_data for _, _, h in self._handlers())
        if use_graphs is None:
            use_graphs = full_batch and str(self.device).startswith("cuda")
        if use_graphs and not full_batch:
            raise ValueError("the graphed step needs batch_size >= number of training points")
        if use_graphs:
            self._train_mfdgp_graphed(True, self.num_epochs_1, self.lr_1)
            self._train_mfdgp_graphed(False, self.num_epochs_2, self.lr_2)
        else:
            self._train_mfdgp(self.update_model, fix_variational_hypers=True, num_epochs=self.num_epochs_1, lr=self.lr_1)
            self._train_mfdgp(self.update_model, fix_variational_hypers=False, num_epochs=self.num_epochs_2, lr=self.lr_2)
        self.models_uncond_trained = True

    # ------------------------------------------------------------------ Pareto solution of posterior samples (row N2)
    def _sample_and_store_pareto_solution(self, nFeatures=500, generator=None, rng=None):
        """One posterior function sample per black-box (top layer), then the feasible Pareto set of the sampled
        problem on a random grid + the training inputs (blackbox_mfdgp_fitter.py:181-216)."""
        from .moop import MOOP, NotFeasiblePoints
        samples_objs = [h.mfdgp.sample_function_from_each_layer(nFeatures=nFeatures, generator=generator)[-1]
                        for h in self.mfdgp_handlers_objs.values()]
        inputs = self.x_train.detach().cpu().double().numpy()
        feasible = -1.0 * self.thresholds_cons.numpy()
        optimizer = None
        for _ in range(MFDGPHandler.MAX_TRIES_FOR_FEASIBLE_GRID):
            samples_cons = [h.mfdgp.sample_function_from_each_layer(nFeatures=nFeatures, generator=generator)[-1]
                            for h in self.mfdgp_handlers_cons.values()]
            optimizer = MOOP(samples_objs, samples_cons, input_dim=inputs.shape[1],
                             grid_size=self.opt_grid_size * inputs.shape[1], pareto_set_size=self.pareto_set_size,
                             feasible_values=feasible, rng=rng)
            res = optimizer.compute_pareto_solution_from_samples(inputs)
            if res is not None:
                break
        else:   # no feasible grid point in any try: settle for the least infeasible points of the last samples
            res = optimizer.compute_pareto_solution_from_samples(inputs, allow_negative_constraints=True)
            if res is None:
                raise NotFeasiblePoints("[ERROR] No feasible points were found in the constraint space! # tries: %d." %
                                        MFDGPHandler.MAX_TRIES_FOR_FEASIBLE_GRID)
        pareto_set, pareto_front, self.samples_objs, self.samples_cons = res
        self.set_pareto_solution(pareto_set, pareto_front)
        return self.pareto_set, self.pareto_front, self.samples_objs, self.samples_cons

    def sample_and_store_pareto_solution(self, **kw):
        from .moop import NotFeasiblePoints
        while True:
            try:
                return self._sample_and_store_pareto_solution(**kw)
            except NotFeasiblePoints:
                print("Not feasible solution found, trying another time!")
                sys.stdout.flush()

    # ------------------------------------------------------------------ conditioned training
    def set_pareto_solution(self, pareto_set, pareto_front):
        """Pareto set (P, d) / front (P, n_obj) to condition on: from ``sample_and_store_pareto_solution`` (the
        reference's route, :181-225) or from any other optimiser."""
        dev = torch.device(self.device)
        self.pareto_set = pareto_set.double().to(dev)
        self.pareto_front = pareto_front.double().to(dev)

    def _thresholds_on(self, device, all_constraints=False):
        """Device copy of the constraint thresholds (uploaded once: a host->device copy is not capturable).
        ``all_constraints``: the vector over the constraints of every rank (omega factors), else this rank's own."""
        src = self.thresholds_cons_global if (all_constraints and self.thresholds_cons_global is not None) \
            else self.thresholds_cons
        c = getattr(self, "_thr_cache", None) or {}
        hit = c.get(all_constraints)
        if hit is None or hit[0] is not src or hit[1].device != torch.device(device):
            hit = (src, src.to(device))
            c[all_constraints] = hit
            self._thr_cache = c
        return hit[1]

    @staticmethod
    def _global_index(h, local_index):
        return local_index if getattr(h, "global_index", None) is None else h.global_index

    def loss_theta_factors(self, cs_mean, cs_var, threshold):
        """:227-233.  On the GPU one launch (functional.cond_factors); host tensors: the plain torch statement."""
        if cs_mean.is_cuda:
            from .. import functional as F
            thr = threshold.reshape(1) if torch.is_tensor(threshold) else torch.tensor([float(threshold)], dtype=cs_mean.dtype,
                                                                                     device=cs_mean.device)
            return F.cond_factors([], [], [cs_mean.reshape(-1)], [cs_var.reshape(-1)], None, thr,
                                  float(np.log(1.0 - self.eps)), float(np.log(self.eps)))
        c = _ncdf((cs_mean - threshold) / torch.sqrt(cs_var))
        return torch.sum(np.log(1.0 - self.eps) * c + np.log(self.eps) * (1.0 - c))

    def loss_omega_factors(self, fs_mean, fs_var, cs_mean, cs_var, pareto_front):
        """:235-243.  fs_* (n_obj, T), cs_* (n_con, T) -- stacked tensors or lists of rows.  On the GPU one launch
        (functional.cond_factors: forward, gradients included); host tensors: the plain torch statement."""
        rows = lambda t: list(t) if isinstance(t, (list, tuple)) else list(t.unbind(0))
        fm, fv, cm, cv = rows(fs_mean), rows(fs_var), rows(cs_mean), rows(cs_var)
        ref = fm[0] if fm else cm[0]
        thr = self._thresholds_on(ref.device, all_constraints=True)
        if thr.numel() != len(cm):
            raise ValueError("omega factors: %d constraint rows but %d thresholds (sharded surrogates need "
                             "set_global_constraint_thresholds)" % (len(cm), thr.numel()))
        if ref.is_cuda and len(fm) <= 8 and len(cm) <= 8:
            from .. import functional as F
            front = self._cached_const(("front", id(pareto_front)), lambda: pareto_front.contiguous())
            return F.cond_factors(fm, fv, cm, cv, front, thr, float(np.log(self.eps)), float(np.log(1.0 - self.eps)))
        fs_mean, fs_var = torch.stack(fm), torch.stack(fv)
        c = torch.ones(fs_mean.shape[-1], dtype=fs_mean.dtype, device=fs_mean.device)
        if cm:
            cs_mean, cs_var = torch.stack(cm), torch.stack(cv)
            c = _prod(_ncdf((cs_mean - thr[:, None]) / torch.sqrt(cs_var)), 0)
        c = c * _prod(_ncdf((pareto_front[:, :, None] - fs_mean) / torch.sqrt(fs_var)), 1)
        return torch.sum(np.log(self.eps) * c + np.log(1 - self.eps) * (1.0 - c))

    def next_conditioned_batch(self, h):
        """The training batch of one model for one conditioned iteration (:281-285, :296-300): the whole data set when the
        handler trains full-batch (every example of the reference: batch_size = N; the loss does not depend on the row
        order), otherwise the next batch of the model's own shuffling loader, re-armed when it runs out."""
        if h.batch_size >= h.num_data:
            return h.train_dataset.tensors
        try:
            return next(h.iter_train_loader)
        except (TypeError, StopIteration):
            h.iter_train_loader = iter(h.train_loader)
            return next(h.iter_train_loader)

    def conditioned_loss(self, x_tilde, eps=None, batches=None):
        """The joint loss of one conditioned-training iteration (:270-343).  ``batches``: optional {(tag, i): (x, y, fid)}
        replacing the loader draw (tests).

        The reference runs three forwards per model (training batch, Pareto set, x_tilde); the layer is separable over
        rows, so here they are ONE forward on the concatenated rows: one Cholesky chain and one set of GEMMs per layer
        instead of three.  ``eps``: optional {name: [None, eps_l1, ...]} with draws for the concatenated rows.
        With surrogates sharded over ranks the omega factors need every model's (mean, var) at x_tilde: the local ones
        carry gradient, the others arrive as constants through one all-gather (mobocmf_amd.parallel)."""
        from .. import parallel
        from .. import functional as F
        from ..gp import MultivariateNormal as MVN
        P, T = self.pareto_set.shape[0], x_tilde.shape[0]
        # The loss is a signed sum of scalar terms.  They are collected and combined in ONE launch at the end, the rows of every
        # layer's moments are split into their three ranges (training batch | Pareto set | x~) by one autograd node per layer,
        # and the theta / omega factors are one launch each: as framework ops (a subtraction per term, slice / stack backward
        # per range, cdf / product / sum chains) this glue was ~100 element-wise launches per iteration of a loop whose cost IS
        # its launch count.
        terms, coefs = [], []
        tilde = {}
        k = 0
        log_e, log_1me = float(np.log(self.eps)), float(np.log(1.0 - self.eps))
        for tag, i, h in self._handlers():
            xb, yb, fb = batches[(tag, i)] if batches is not None else self.next_conditioned_batch(h)
            B = xb.shape[0]
            S = h.mfdgp.num_samples_for_training
            top = h.num_fidelities - 1
            out = h.mfdgp(torch.cat([xb, self.pareto_set, x_tilde], 0), eps=None if eps is None else eps[(tag, i)])
            parts = []
            for d in out:
                rpb = d.mean.numel() // (B + P + T)      # rows of the layer per input row
                parts.append(F.split_rows(d.mean, d.variance, [B * rpb, P * rpb, T * rpb]))
            batch = [MVN(pm[0], pv[0]) for pm, pv in parts]
            terms.append(h.elbo(batch, yb.T, fb)[0])
            coefs.append(-float(h.num_data) / B)
            mu_p, var_p = parts[top][0][1], parts[top][1][1]
            if tag == "OBJ":
                pf = self._cached_const(("pf", top, P), lambda: torch.full((P, 1), float(top), dtype=xb.dtype, device=xb.device))
                pl = [None] * top + [MVN(mu_p, var_p)]
                gi = self._global_index(h, i)                 # the front's columns follow the GLOBAL objective order
                col = self._cached_const(("front_col", gi), lambda: self.pareto_front[:, gi:gi + 1].T.contiguous())
                terms.append(h.elbo(pl, col, pf, include_kl_term=False))
                coefs.append(-1.0)
            else:
                if S > 1:
                    raise NotImplementedError("theta factors are defined for one sample per row (reference: S = 1)")
                thr_k = self._thresholds_on(xb.device)[k:k + 1]
                terms.append(F.cond_factors([], [], [mu_p], [var_p], None, thr_k, log_1me, log_e))      # :227-233
                coefs.append(-1.0)
                k += 1
            tilde[(tag, i)] = (parts[top][0][2], parts[top][1][2])
        fm = [tilde[(t, i)][0] for t, i, _ in self._handlers() if t == "OBJ"]
        fv = [tilde[(t, i)][1] for t, i, _ in self._handlers() if t == "OBJ"]
        cm = [tilde[(t, i)][0] for t, i, _ in self._handlers() if t == "CON"]
        cv = [tilde[(t, i)][1] for t, i, _ in self._handlers() if t == "CON"]
        if parallel.world()[1] > 1:
            oi = [self._global_index(h, i) for t, i, h in self._handlers() if t == "OBJ"]
            ci = [self._global_index(h, i) for t, i, h in self._handlers() if t == "CON"]
            stk = lambda rows: torch.stack(rows) if rows else x_tilde.new_zeros((0, T))
            gfm, gfv, gcm, gcv = parallel.gather_with_local_grad(stk(fm), stk(fv), stk(cm), stk(cv), oi, ci)
            fm, fv, cm, cv = list(gfm.unbind(0)), list(gfv.unbind(0)), list(gcm.unbind(0)), list(gcv.unbind(0))
        terms.append(self.loss_omega_factors(fm, fv, cm, cv, self.pareto_front))
        coefs.append(-1.0)
        return F.scalar_combine(terms, coefs)

    def _cached_const(self, key, make):
        """Small constant device tensors of the conditioned loss, built once (a fill / copy launch per iteration otherwise)."""
        c = self.__dict__.setdefault("_const_cache", {})
        hit = c.get(key)
        if hit is None or hit[0] is not self.pareto_front:
            hit = (self.pareto_front, make())
            c[key] = hit
        return hit[1]

    def train_conditioned_mfdgps(self, num_iters=None, use_graphs=None):
        """ONE Adam over all models' parameters, kernel hyper-parameters frozen (:245-268, :345-354).  On the GPU the
        whole iteration (x~ draw, joint loss over all surrogates, backward, Adam) is replayed from a HIP graph; a failed
        Cholesky inside a replay rolls back to the last verified state and continues eagerly (jitter ladder)."""
        from .. import parallel
        from ..layers.mfdgp_hidden_layer import NotPSDError
        from .graphed_step import GraphedConditionedStep
        for _, _, h in self._handlers():
            h.mfdgp.fix_variational_hypers_cond(True)
        num_iters = self.num_epochs_2 if num_iters is None else num_iters
        full_batch = all(h.batch_size >= h.num_data for _, _, h in self._handlers())
        if use_graphs is None:
            use_graphs = self.pareto_set.is_cuda and parallel.world()[1] == 1 and full_batch
        if use_graphs and not full_batch:
            raise ValueError("a captured conditioned step needs batch_size >= number of training points (mini-batches come "
                             "from a host-side loader)")
        tiny = None
        if use_graphs and self.use_tiny_step and parallel.world()[1] == 1:
            # the reference's own sizes: the whole iteration in 3 + n_con launches (util/tiny_step.py)
            done, tiny = self._train_conditioned_tiny(num_iters)
            num_iters -= done
        if num_iters > 0:
            step = GraphedConditionedStep(self, lr=self.lr_2, use_graph=use_graphs,
                                          stream=self._stream_for(0, self.pareto_set.device) if self.pareto_set.is_cuda else None)
            if tiny is not None:      # a Cholesky failed there: this path (jitter ladder) continues with its optimiser state
                for k in range(len(tiny.models)):
                    tiny.export_adam_state(k, step.optimizer)
            step.snapshot()
        last_good = -1
        for i in range(num_iters):
            step.step()
            if (i % ITER_PRINT) == 0 or (i + 1) == num_iters:
                try:
                    step.check()
                    step.snapshot()
                except (NotPSDError, FloatingPointError) as err:
                    warnings.warn("conditioned training: %s -- rolling back %d iterations and redoing them eagerly" %
                                  (err, i - last_good))
                    step.restore_and_go_eager()
                    for _ in range(i - last_good):
                        step.step()
                    step.check()
                    step.snapshot()
                last_good = i
                if self.verbose:
                    print("Iter:", i, "/", num_iters, ". Neg. ELBO per iter:", step.loss.item())
                    sys.stdout.flush()
        if num_iters > 0:
            step.stream.synchronize()
            torch.cuda.current_stream(self.pareto_set.device).wait_stream(step.stream)
            step.retire()
        for _, _, h in self._handlers():
            h.iter_train_loader = None
            h.mfdgp.set_check_pd(True)

    def _train_conditioned_tiny(self, num_iters):
        """Conditioned training through TinyConditionedStep when every surrogate fits it.  Returns (iterations completed, step
        or None): fewer than ``num_iters`` when the surrogates do not fit (0, None) or after a failed Cholesky (state rolled
        back to the last verified iteration; the layer path continues)."""
        from .. import _lib
        from ..layers.mfdgp_hidden_layer import NotPSDError
        from .coop_step import CoopConditionedStep
        from .tiny_step import TinyConditionedStep
        dev = self.pareto_set.device
        step = None
        for cls in (TinyConditionedStep, CoopConditionedStep):      # M <= 32 in one workgroup per surrogate, M <= 128 in several
            try:
                step = cls(self, lr=self.lr_2, stream=self._stream_for(0, dev))
                break
            except _lib.MobocmfError:
                continue
        if step is None:
            return 0, None
        step.stream.wait_stream(torch.cuda.current_stream(dev))
        step.snapshot()
        last_good = -1
        for i in range(num_iters):
            step.step()
            if (i % ITER_PRINT) == 0 or (i + 1) == num_iters:
                try:
                    step.check()
                except (NotPSDError, FloatingPointError) as err:
                    warnings.warn("conditioned training: %s -- rolling back %d iterations; the layer path continues" %
                                  (err, i - last_good))
                    step.restore()
                    torch.cuda.current_stream(dev).wait_stream(step.stream)
                    return last_good + 1, step
                step.snapshot()
                last_good = i
                if self.verbose:
                    print("Iter:", i, "/", num_iters, ". Neg. ELBO per iter:", step.loss.item())
                    sys.stdout.flush()
        step.stream.synchronize()
        torch.cuda.current_stream(dev).wait_stream(step.stream)
        return num_iters, step

    def mfdgps_to_train_mode(self):
        for _, _, h in self._handlers():
            h.mfdgp.train()

    def mfdgps_to_eval_mode(self):
        for h in self.mfdgp_handlers_objs.values():
            h.mfdgp.eval()
        for h in self.mfdgp_handlers_cons.values():
            h.mfdgp.train()                      # as written in the reference (:363-368)

    def copy_uncond(self):
        """Deep copy of the fitter (:372-397): the models only hold tensors, so ``deepcopy`` just works."""
        for _, _, h in self._handlers():
            h.mfdgp.eval()
        self_copy = deepcopy(self)
        for _, _, h in self._handlers() + self_copy._handlers():
            h.mfdgp.train()
        return self_copy
