"""The ELBO step of SMALL surrogates (zero_grad + MFDGP.forward + VariationalELBOMF + backward + Adam,
blackbox_mfdgp_fitter.py:161-171) as ONE launch for a whole group of models: mobocmf_tiny_elbo_step.

At the reference's own sizes -- M = N = tens of points (examples/example_acquisition_mfdgp_forrester/...py:51-62) -- a step
through the layer entry points is ~50 dependent launches of ~4.7 us whatever they compute; here one workgroup per surrogate
runs the whole step as barrier-separated phases (csrc/tiny_step.hip).  ``TinyELBOStep`` has the surface of
``GraphedELBOStep`` that the fitter's training loop uses (step / check / snapshot / loss), for several models at once.
"""
import ctypes
import os

import torch

from .. import _lib
from .. import gp

MAX_COLUMNS = 1024      # rows[l] * S per layer: hard limit of this binding
LAYER_PATH_US = 240.0   # what a step of a small surrogate costs through the layer entry points (HIP-graph replay, MI355X)


def estimated_us(M, columns):
    """Duration of one launch, from tools/tiny_sweep.py on MI355X (profiles/r04_tiny_step.txt): a fixed part that grows with
    the chain (M^2) and a part per panel column -- one workgroup does everything, so beyond a few dozen columns at M = 32
    (a few hundred at M = 16) the grid-filling kernels of the layer path win and ``eligible`` says no."""
    return 20.0 + 0.15 * M * M + (M * M / 900.0) * float(sum(columns))


def _hyper_params(layer):
    return [getattr(m, n) for m, n in gp._hyper_sources(layer.covar_module, layer.kind)]


def fits_predict(model, fidelity, T, d, speed_rule=True, max_m=None):
    """True when ``model``'s predictive moments at T test points up to layer ``fidelity`` fit the one-launch kernel (the
    structural limits of ``eligible``; the training flags do not matter: prediction runs the eval branch).  ``max_m``: the
    inducing-point limit of the kernel asked about (default: the one-workgroup kernel's)."""
    try:
        layers = model._layers()[:fidelity + 1]
        if not (1 <= len(layers) <= _lib.TINY_MAX_LAYERS) or model.use_only_highest_fidelity or not (1 <= d <= _lib.TINY_MAX_D):
            return False
        Z0 = layers[0].variational_strategy._inducing_points
        M = Z0.shape[0]
        if not (1 <= M <= (max_m or _lib.TINY_MAX_M)) or Z0.shape[1] != d or not Z0.is_cuda:
            return False
        S = model.num_samples_for_acquisition
        cols = [T] + [T * S] * (len(layers) - 1)
        if max(cols) > MAX_COLUMNS * 4 or (speed_rule and estimated_us(M, cols) > LAYER_PATH_US):
            return False
        jit = layers[0].variational_strategy.jitter_val
        for l, layer in enumerate(layers):
            vs = layer.variational_strategy
            vd = vs._variational_distribution
            Zl = vs._inducing_points
            if layer.kind != (0 if l == 0 else 1) or vs.jitter_val != jit or Zl.shape[0] != M:
                return False
            if l and (not torch.equal(Zl[:, :-1], Z0) or layer.samples.numel() != S):
                return False
            lik = getattr(model, model.name_hidden_layer_likelihood + str(l))
            c = lik.raw_noise_constraint
            if type(c) is not gp.Interval or not (c.upper_bound > c.lower_bound) or c.upper_bound == float("inf"):
                return False
            ps = _hyper_params(layer) + [vd.variational_mean, vd.chol_variational_covar, lik.raw_noise]
            if not all(p.is_cuda and p.dtype == torch.float64 and p.is_contiguous() for p in ps):
                return False
            if not all(type(getattr(m, n + "_constraint")) is gp.Positive
                       for m, n in gp._hyper_sources(layer.covar_module, layer.kind)):
                return False
        return True
    except AttributeError:
        return False


def eligible(model, x, fidelities, speed_rule=True, max_m=None, max_columns=None):
    """(``max_m`` / ``max_columns``: the limits of the kernel asked about; default: the one-workgroup kernel's.)  True when ``model`` on the batch ``x`` fits the one-launch step: <= 3 layers sharing one set of <= 32 inducing inputs
    (Z~_l = [Z_x, m_{l-1}]), d <= 8, softplus / Interval constraints, float64 parameters on the GPU, every fidelity's
    prefix non-empty -- and, with ``speed_rule``, small enough for one workgroup to beat the layer path (estimated_us)."""
    try:
        layers = model._layers()
        L = len(layers)
        if not (1 <= L <= _lib.TINY_MAX_LAYERS) or model.use_only_highest_fidelity or model._eval_mode:
            return False
        if not x.is_cuda or x.dtype != torch.float64 or x.dim() != 2 or not (1 <= x.shape[1] <= _lib.TINY_MAX_D):
            return False
        Z0 = layers[0].variational_strategy._inducing_points
        M = Z0.shape[0]
        if not (1 <= M <= (max_m or _lib.TINY_MAX_M)) or Z0.requires_grad or Z0.shape[1] != x.shape[1]:
            return False
        S = model.num_samples_for_training
        fidv = fidelities.reshape(-1)
        N = fidv.numel()
        if N != x.shape[0] or N * max(S, 1) > (max_columns or MAX_COLUMNS):
            return False
        counts = [int((fidv >= l).sum()) for l in range(L)]
        if counts[0] != N or counts[-1] < 1:
            return False
        if speed_rule and estimated_us(M, [c * (S if l else 1) for l, c in enumerate(counts)]) > LAYER_PATH_US:
            return False
        jit = layers[0].variational_strategy.jitter_val
        for l, layer in enumerate(layers):
            vs = layer.variational_strategy
            vd = vs._variational_distribution
            if layer.kind != (0 if l == 0 else 1) or vs.jitter_val != jit or not layer.training:
                return False
            Zl = vs._inducing_points
            if Zl.requires_grad or Zl.shape[0] != M:
                return False
            if l and not torch.equal(Zl[:, :-1], Z0):      # layers >= 1 share Z_x; their f column is m_{l-1} (F9)
                return False
            ps = _hyper_params(layer) + [vd.variational_mean, vd.chol_variational_covar]
            lik = getattr(model, model.name_hidden_layer_likelihood + str(l))
            c = lik.raw_noise_constraint
            if type(c) is not gp.Interval or not (c.upper_bound > c.lower_bound) or c.upper_bound == float("inf"):
                return False
            ps.append(lik.raw_noise)
            if not all(p.is_cuda and p.dtype == torch.float64 and p.is_contiguous() for p in ps):
                return False
            if not all(type(getattr(m, n + "_constraint")) is gp.Positive
                       for m, n in gp._hyper_sources(layer.covar_module, layer.kind)):
                return False
        return True
    except AttributeError:
        return False


class TinyELBOStep:
    """``step()`` == one full-batch ELBO step of EVERY model of the group, one launch.  ``models[i]`` trains on
    ``(xs[i], ys[i], fids[i])`` (y: (N, 1) or (N,); fidelities as the ELBO takes them).  The rows are ordered once by
    descending fidelity (``row_order[i]``; a full-batch ELBO is a sum over rows) and layer l runs on the rows of fidelity
    >= l, as ``GraphedELBOStep(prune_rows=True)``.  ``losses`` is an (n, 3) device tensor: ELBO, scaled KL, -ELBO per model,
    as of the last step (before its update)."""

    def __init__(self, models, num_data, xs, ys, fids, lr, betas=(0.9, 0.999), eps=1e-8, stream=None, fixed_eps=None,
                 want_grad=False, prepared=None, force=False):
        """``force``: take every size the kernel accepts, also those where the layer path is faster (tests, sweeps).
        ``prepared`` (TinyConditionedStep): per model a dict with the rows ALREADY in the kernel's order and the optional
        fields of mobocmf_tiny_model -- x, y, fid, rows, row_weight, kl_scale, seeds (bool), rand (row0, rows), xrng, eps
        (per layer, prefix columns)."""
        lib = _lib.require_device()
        self.models = list(models)
        n = len(self.models)
        dev = (xs[0] if prepared is None else prepared[0]["x"]).device
        self.device = dev
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        self.host = (_lib.TinyModel * n)()
        self.losses = torch.zeros(n, 3, dtype=torch.float64, device=dev)
        self.infos = torch.zeros(n, _lib.TINY_MAX_LAYERS, dtype=torch.int32, device=dev)
        self.steps_done = torch.zeros(n, dtype=torch.int64, device=dev)
        self.row_order, self.layer_rows = [], []
        self._keep = []          # tensors the descriptors point at
        self.exp_avg, self.exp_avg_sq, self.grads, self._work = [], [], [], []
        self._segments = []      # per model: [(parameter tensor, flat offset, length)]
        self.top_moments, self.seeds, self.x_rows = [], [], []      # prepared models: (2, ncol_top) tensors, the x array
        for i, model in enumerate(self.models):
            prep = None if prepared is None else prepared[i]
            x, y, fid = (xs[i], ys[i], fids[i]) if prep is None else (prep["x"], prep["y"], prep["fid"])
            if not self._eligible(model, x, fid, force):
                raise _lib.MobocmfError("%s: model %d does not fit the one-launch step (see eligible())" % (type(self).__name__, i))
            layers = model._layers()
            L, S = len(layers), model.num_samples_for_training
            fidv = fid.reshape(-1).to(torch.float64)
            N = fidv.numel()
            if prep is None:
                counts = [int((fidv >= l).sum()) for l in range(L)]
                order = torch.argsort(fidv, descending=True, stable=True)
                xo, yo, fo = x[order].contiguous(), y.reshape(-1)[order].to(torch.float64).contiguous(), fidv[order].contiguous()
            else:
                counts, order = [int(r) for r in prep["rows"]], None
                xo, yo, fo = x.contiguous(), y.reshape(-1).to(torch.float64).contiguous(), fidv.contiguous()
            self.row_order.append(order)
            self.layer_rows.append(counts)
            T = self.host[i]
            T.L, T.M, T.d, T.S, T.N = L, layers[0].variational_strategy._inducing_points.shape[0], x.shape[1], S, N
            Zx = layers[0].variational_strategy._inducing_points.detach().contiguous()
            T.x, T.y, T.fid, T.Zx = xo.data_ptr(), yo.data_ptr(), fo.data_ptr(), Zx.data_ptr()
            self._keep += [xo, yo, fo, Zx]
            T.kl_scale = N / float(num_data[i]) if prep is None else float(prep["kl_scale"])
            T.jitter = layers[0].variational_strategy.jitter_val
            if prep is not None:
                ntop = counts[-1] * (S if L > 1 else 1)
                if prep.get("row_weight") is not None:
                    w = prep["row_weight"].to(torch.float64).contiguous()
                    T.row_weight = w.data_ptr()
                    self._keep.append(w)
                tm = torch.zeros(2, ntop, dtype=torch.float64, device=dev)
                T.top_mean, T.top_var = tm[0].data_ptr(), tm[1].data_ptr()
                self.top_moments.append(tm)
                if prep.get("seeds"):
                    sd = torch.zeros(2, ntop, dtype=torch.float64, device=dev)
                    T.seed_gmean, T.seed_gvar, T.seed_scale = sd[0].data_ptr(), sd[1].data_ptr(), float(prep.get("seed_scale", 1.0))
                    self.seeds.append(sd)
                if prep.get("xrng") is not None:
                    T.xrng = prep["xrng"].data_ptr()
                    T.rand_row0, T.rand_rows = prep["rand"]
                    self._keep.append(prep["xrng"])
                self.x_rows.append(xo)
            segs = []
            off = 0
            for l, layer in enumerate(layers):
                vd = layer.variational_strategy._variational_distribution
                lik = getattr(model, model.name_hidden_layer_likelihood + str(l))
                T.rows[l] = counts[l]
                tr = 0
                for s, p in enumerate(_hyper_params(layer)):
                    T.raw[l][s] = p.data_ptr()
                    tr |= int(p.requires_grad) << s
                    segs.append((p, off, p.numel()))
                    off += p.numel()
                for bit, p in ((7, vd.variational_mean), (8, vd.chol_variational_covar)):
                    tr |= int(p.requires_grad) << bit
                    segs.append((p, off, p.numel()))
                    off += p.numel()
                T.m[l], T.L_S[l] = vd.variational_mean.data_ptr(), vd.chol_variational_covar.data_ptr()
                tr |= int(lik.raw_noise.requires_grad) << 9
                T.trainable[l] = tr
                T.raw_noise[l] = lik.raw_noise.data_ptr()
                T.noise_lo[l], T.noise_hi[l] = lik.raw_noise_constraint.lower_bound, lik.raw_noise_constraint.upper_bound
                if l:
                    if prep is not None:
                        e = None if prep.get("eps") is None else prep["eps"][l]
                        if e is not None:
                            e = e.reshape(-1)[:counts[l] * S].contiguous()
                    else:
                        e = None if fixed_eps is None or fixed_eps[i] is None else fixed_eps[i][l]
                        if e is not None:      # given for the batch as passed in (N * S values): follow the rows
                            e = e.reshape(N, S)[order][:counts[l]].reshape(-1).contiguous()
                    if e is not None:
                        T.eps[l] = e.data_ptr()
                        self._keep.append(e)
                    rng = layer._rng(dev)
                    T.rng[l] = rng.data_ptr()
                    self._keep.append(rng)
            for l in range(L):
                lik = getattr(model, model.name_hidden_layer_likelihood + str(l))
                segs.append((lik.raw_noise, off, 1))
                off += 1
            flat = ctypes.c_int64()
            _lib.check(lib.mobocmf_tiny_flat_len(ctypes.byref(T), ctypes.byref(flat)), "mobocmf_tiny_flat_len")
            assert flat.value == off, (flat.value, off)
            wb = ctypes.c_size_t()
            _lib.check(getattr(lib, self._work_bytes_fn)(ctypes.byref(T), ctypes.byref(wb)), self._work_bytes_fn)
            # MOBOCMF_POISON (as functional._scratch): NaN-filled workspace, so a read of anything the launch did not write shows
            work = torch.full((wb.value // 8,), float("nan") if os.environ.get("MOBOCMF_POISON") else 0.0,
                              dtype=torch.float64, device=dev)
            ea, eq = torch.zeros(off, dtype=torch.float64, device=dev), torch.zeros(off, dtype=torch.float64, device=dev)
            self._work.append(work)
            self.exp_avg.append(ea)
            self.exp_avg_sq.append(eq)
            T.work, T.adam_m, T.adam_v = work.data_ptr(), ea.data_ptr(), eq.data_ptr()
            T.steps_done = self.steps_done[i:i + 1].data_ptr()
            T.out = self.losses[i].data_ptr()
            T.info = self.infos[i].data_ptr()
            if want_grad:
                gflat = torch.zeros(off, dtype=torch.float64, device=dev)
                self.grads.append(gflat)
                T.grad = gflat.data_ptr()
            self._segments.append(segs)
        raw = bytes(self.host)
        self._dev_table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self._snap = None
        self._order_after_setup()

    def _order_after_setup(self):
        """Everything above was allocated, zero-filled and uploaded on the CURRENT stream; the launches run on ``self.stream``.
        Without this edge a fill kernel could still be pending when the first launch starts (found with a cooperative launch
        whose arrival counters were zeroed under it: tools/coop_concurrency_probe.py)."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    # ------------------------------------------------------------------ the kernel this class drives
    _work_bytes_fn = "mobocmf_tiny_work_bytes"

    @staticmethod
    def _eligible(model, x, fid, force):
        return eligible(model, x, fid, speed_rule=not force)

    # ------------------------------------------------------------------ the step
    def _launch(self, mode):
        """mode: 0 gradients only, 1 the step, 2 forward only (mobocmf_tiny_elbo_step's do_update)."""
        lib = _lib.require_device()
        _lib.check(lib.mobocmf_tiny_elbo_step(ctypes.cast(self.host, ctypes.c_void_p), ctypes.c_void_p(self._dev_table.data_ptr()),
                                              len(self.models), self.lr, self.betas[0], self.betas[1], self.eps,
                                              int(mode), ctypes.c_void_p(self.stream.cuda_stream)),
                   "mobocmf_tiny_elbo_step")

    def step(self):
        """Enqueues one step of every model on ``self.stream``."""
        self._launch(1)
        return self.losses

    def gradients(self):
        """-ELBO and its raw-parameter gradients at the current parameters, no update (``want_grad=True``): per model a dict
        parameter -> gradient tensor of its shape."""
        assert self.grads, "construct with want_grad=True"
        self._launch(0)
        self.stream.synchronize()
        return [{p: self.grads[i][off:off + n].reshape(p.shape) for p, off, n in segs} for i, segs in enumerate(self._segments)]

    @property
    def loss(self):
        return self.losses[:, 2]

    @property
    def kl(self):
        return self.losses[:, 1]

    # ------------------------------------------------------------------ verdicts / roll-back (as GraphedELBOStep)
    def check(self):
        """Synchronising: raises if a Cholesky of the last step failed or a loss is not finite."""
        from ..layers.mfdgp_hidden_layer import NotPSDError
        self.stream.synchronize()
        if bool((self.infos != 0).any()):
            i, l = [int(v) for v in torch.nonzero(self.infos)[0]]
            code = int(self.infos[i, l])
            if code < 0:      # (-1: a workgroup gave up waiting at an in-launch barrier; -2: the launch did not match its coupling record)
                raise FloatingPointError("one-launch step: in-launch barrier abandoned in model %d (info %d)" % (i, code))
            raise NotPSDError("K_mm not positive definite in model %d layer %d (pivot %d)" % (i, l, code))
        if not bool(torch.isfinite(self.losses).all()):
            raise FloatingPointError("non-finite ELBO")

    def snapshot(self):
        with torch.cuda.stream(self.stream):
            self._snap = ([p.detach().clone() for m in self.models for p in m.parameters()],
                          [t.clone() for t in self.exp_avg], [t.clone() for t in self.exp_avg_sq], self.steps_done.clone(),
                          [l._rng(self.device).clone() for m in self.models for l in m._layers()])

    def restore(self):
        """Back to the last snapshot (parameters, optimiser state, eps streams)."""
        self.stream.synchronize()
        ps, ea, eq, steps, rngs = self._snap[:5]
        with torch.no_grad(), torch.cuda.stream(self.stream):      # (ordered with the launches that follow on this stream)
            for p, s0 in zip([p for m in self.models for p in m.parameters()], ps):
                p.copy_(s0)
            for t, s0 in zip(self.exp_avg, ea):
                t.copy_(s0)
            for t, s0 in zip(self.exp_avg_sq, eq):
                t.copy_(s0)
            self.steps_done.copy_(steps)
            for layer, s0 in zip([l for m in self.models for l in m._layers()], rngs):
                layer._rng(self.device).copy_(s0)

    def export_adam_state(self, i, optimizer):
        """Copies model i's moment estimates and step count into a FusedAdam over ``list(models[i].parameters())`` (the
        layer-path step that takes over after a failed Cholesky keeps the optimiser's memory)."""
        by_param = {id(p): (off, n) for p, off, n in self._segments[i]}
        with torch.no_grad():
            for k, p in enumerate(optimizer.params):
                if id(p) in by_param:
                    off, n = by_param[id(p)]
                    optimizer.state[k]["exp_avg"].copy_(self.exp_avg[i][off:off + n].reshape(p.shape))
                    optimizer.state[k]["exp_avg_sq"].copy_(self.exp_avg_sq[i][off:off + n].reshape(p.shape))
            optimizer.steps_done.copy_(self.steps_done[i])


def _ptrs(vals):
    return (ctypes.c_void_p * len(vals))(*vals) if vals else None


class TinyConditionedStep(TinyELBOStep):
    """One iteration of the conditioned training (blackbox_mfdgp_fitter.py:245-354: fresh x~ ~ U[0,1]^(n_tilde x d), the joint
    loss over ALL surrogates, one Adam) in ONE launch instead of ~180 (mobocmf_tiny_elbo_step mode 4: every model on [Pareto
    set | x~ | its batch], x~ drawn in the launch; after the forward the models' workgroups meet at a barrier and each forms
    the theta / omega factor gradients of its model from all models' top-layer moments), or, ``one_launch = False``, in
    3 + n_con: forward-only launch, mobocmf_cond_factors_forward for the theta factors of every constraint and for the omega
    factors on the top layers' moments, step launch with those gradients entering at the top layers' columns.
    Rows per model: the P Pareto points (objectives: scored against their column of the Pareto front, weight 1, :288-291;
    constraints: weight 0, theta factors :227-233), the x~ (weight 0, omega factors :235-243), the batch (weight num_data / B,
    KL weight 1: -elbo / B * num_data, :281-303).  One sample per row (the reference's S = 1)."""

    def __init__(self, fitter, lr, betas=(0.9, 0.999), eps=1e-8, stream=None, n_tilde=10, fixed_x_tilde=None, fixed_eps=None,
                 want_grad=False):
        hs = fitter._handlers()
        dev = fitter.pareto_set.device
        P, d = fitter.pareto_set.shape
        Tn = n_tilde if fixed_x_tilde is None else fixed_x_tilde.shape[0]
        if Tn < 1 or P < 1:
            raise _lib.MobocmfError("TinyConditionedStep: at least one x~ point and one Pareto point (got %d, %d)" % (Tn, P))
        self.fitter, self.P, self.T = fitter, P, Tn
        self.xrng = None
        if fixed_x_tilde is None:
            self.xrng = torch.tensor([int(torch.randint(1, 2 ** 62, (), dtype=torch.int64)), 0], dtype=torch.int64, device=dev)
        prepared, self._orders = [], []
        for tag, i, h in hs:
            xb, yb, fb = h.train_dataset.tensors
            if h.mfdgp.num_samples_for_training != 1:
                raise _lib.MobocmfError("TinyConditionedStep: one training sample per row (the reference's S = 1)")
            B, top = xb.shape[0], h.num_fidelities - 1
            fv = fb.reshape(-1).to(torch.float64)
            order = torch.argsort(fv, descending=True, stable=True)
            self._orders.append(order)
            xt = torch.zeros(Tn, d, dtype=torch.float64, device=dev) if fixed_x_tilde is None else fixed_x_tilde.to(dev).double()
            z = lambda n: torch.zeros(n, dtype=torch.float64, device=dev)
            if tag == "OBJ":
                gi = fitter._global_index(h, i)
                yp, wp = fitter.pareto_front[:, gi].to(torch.float64), torch.ones(P, dtype=torch.float64, device=dev)
            else:
                yp, wp = z(P), z(P)
            e = None
            if fixed_eps is not None and fixed_eps.get((tag, i)) is not None:
                # given for the rows [batch | Pareto | x~] (fitter.conditioned_loss): into this step's [Pareto | x~ | batch sorted]
                idx = torch.cat([torch.arange(B, B + P + Tn, device=dev), order])
                e = [None if v is None else v.reshape(-1)[idx] for v in fixed_eps[(tag, i)]]
            prepared.append(dict(
                x=torch.cat([fitter.pareto_set.double(), xt, xb[order].double()], 0),
                y=torch.cat([yp, z(Tn), yb.reshape(-1)[order].double()]),
                fid=torch.cat([torch.full((P + Tn,), float(top), dtype=torch.float64, device=dev), fv[order]]),
                rows=[P + Tn + int((fv >= l).sum()) for l in range(h.num_fidelities)],
                row_weight=torch.cat([wp, z(Tn), torch.full((B,), float(h.num_data) / B, dtype=torch.float64, device=dev)]),
                kl_scale=1.0, seeds=True, seed_scale=-1.0, xrng=self.xrng, rand=(P, Tn), eps=e))
        self._roles = [(0 if tag == "OBJ" else 1) for tag, _, _ in hs]
        super().__init__([h.mfdgp for _, _, h in hs], None, None, None, None, lr, betas=betas, eps=eps, stream=stream,
                         want_grad=want_grad, prepared=prepared)
        # the factor launches: pointer tables into the models' top-layer moments / seed arrays, built once
        obj = [k for k, (tag, _, _) in enumerate(hs) if tag == "OBJ"]
        con = [k for k, (tag, _, _) in enumerate(hs) if tag == "CON"]
        if len(obj) > 8 or len(con) > 8:
            raise _lib.MobocmfError("TinyConditionedStep: at most 8 objectives and 8 constraints")
        self._obj, self._con = obj, con
        self.factor_losses = torch.zeros(len(con) + 1, dtype=torch.float64, device=dev)
        log_e, log_1me = float(torch.log(torch.tensor(fitter.eps, dtype=torch.float64))), \
            float(torch.log1p(torch.tensor(-fitter.eps, dtype=torch.float64)))
        thr = fitter._thresholds_on(dev).to(torch.float64).contiguous()
        front = fitter.pareto_front.to(torch.float64).contiguous()
        self._keep += [thr, front]
        mom = lambda k, r, off: self.top_moments[k][r].data_ptr() + 8 * off
        sd = lambda k, r, off: self.seeds[k][r].data_ptr() + 8 * off
        self._theta = []
        for j, k in enumerate(con):      # theta factors of constraint j at the Pareto points (columns [0, P) of its top layer)
            self._theta.append((0, 1, 1, P, None, None, _ptrs([mom(k, 0, 0)]), _ptrs([mom(k, 1, 0)]), None,
                                ctypes.c_void_p(thr.data_ptr() + 8 * j), log_1me, log_e,
                                ctypes.c_void_p(self.factor_losses[j:j + 1].data_ptr()), None, None,
                                _ptrs([sd(k, 0, 0)]), _ptrs([sd(k, 1, 0)])))
        # omega factors at the x~ (columns [P, P + T)) over all objectives and constraints
        self._omega = (len(obj), len(con), P, Tn, _ptrs([mom(k, 0, P) for k in obj]), _ptrs([mom(k, 1, P) for k in obj]),
                       _ptrs([mom(k, 0, P) for k in con]), _ptrs([mom(k, 1, P) for k in con]),
                       ctypes.c_void_p(front.data_ptr()), ctypes.c_void_p(thr.data_ptr()), log_e, log_1me,
                       ctypes.c_void_p(self.factor_losses[len(con):].data_ptr()),
                       _ptrs([sd(k, 0, P) for k in obj]), _ptrs([sd(k, 1, P) for k in obj]),
                       _ptrs([sd(k, 0, P) for k in con]), _ptrs([sd(k, 1, P) for k in con]))
        self._setup_coupling(front, thr, log_e, log_1me)

    def _setup_coupling(self, front, thr, log_e, log_1me):
        """mobocmf_tiny_coupling for the one-launch iteration (mode 4) and every model's role in it; the descriptor table is
        uploaded again with those fields set."""
        cp = _lib.TinyCoupling()
        cp.n_obj, cp.n_con, cp.P, cp.T = len(self._obj), len(self._con), self.P, self.T
        for j, k in enumerate(self._obj):
            cp.obj_model[j] = k
            self.host[k].role, self.host[k].role_index = 0, j
        for j, k in enumerate(self._con):
            cp.con_model[j] = k
            self.host[k].role, self.host[k].role_index = 1, j
        cp.front, cp.thresholds = front.data_ptr(), thr.data_ptr()
        cp.log_eps, cp.log_1m_eps = log_e, log_1me
        cp.losses = self.factor_losses.data_ptr()
        self._barrier = torch.zeros(1, dtype=torch.int64, device=self.device)
        cp.barrier = self._barrier.data_ptr()
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)      # sticky: OR'd by the launches, read by check()
        cp.status = self._status.data_ptr()
        cp.n_models = len(self.models)
        self._coupling = torch.frombuffer(bytearray(bytes(cp)), dtype=torch.uint8).to(self.device)
        for k in range(len(self.models)):
            self.host[k].coupling = self._coupling.data_ptr()
        self._dev_table = torch.frombuffer(bytearray(bytes(self.host)), dtype=torch.uint8).to(self.device)
        self._order_after_setup()

    def _factors(self):
        lib = _lib.require_device()
        st = ctypes.c_void_p(self.stream.cuda_stream)
        for a in self._theta + [self._omega]:
            _lib.check(lib.mobocmf_cond_factors_forward(*a, st), "mobocmf_cond_factors_forward")

    use_graph = True      # the iteration's launch(es) replayed from one HIP graph
    one_launch = True     # the whole iteration as ONE launch (mode 4: the factor terms formed inside, after an in-launch barrier
    #                       of the models' workgroups); False: forward-only launch + factor launches + step launch

    def _capture(self):
        """The launch(es) of an iteration captured once: every argument is static (pointers, sizes, the learning rate), x~ and
        eps come from device-side counters, so a replay IS the next iteration.  The capture pass itself executes nothing."""
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
            self._issue()
        self._graph = g

    def _issue(self):
        if self.one_launch and self.T <= 256 and len(self.models) <= 64:
            try:
                self._launch(4)
                return
            except _lib.MobocmfError:
                # refused (MOBOCMF_BAD_ARG: more models than this device keeps resident at once -- the in-launch barrier
                # needs them all): nothing was enqueued; the three-launch form has no such requirement
                self.one_launch = False
        self._launch(2)
        self._factors()
        self._launch(1)

    def check(self):
        """As TinyELBOStep.check, plus the sticky status word of the one-launch form: a workgroup that gave up waiting at the
        in-launch barrier -- in ANY iteration since the last check, not only the last one -- left its model untouched while its
        peers moved on; that is reported like a failed Cholesky (the fitter rolls back to its last snapshot)."""
        super().check()
        st = int(self._status.item())
        if st:
            self._status.zero_()
            raise FloatingPointError("one-launch conditioned iteration: %s (status %d)" %
                                     ("a workgroup timed out at the in-launch barrier" if st & 1 else
                                      "the launch did not match its coupling record", st))

    def step(self):
        if self.use_graph:
            if self.__dict__.get("_graph") is None:
                self._capture()
            with torch.cuda.stream(self.stream):
                self._graph.replay()
        else:
            self._issue()
        return self.losses

    def gradients(self):
        assert self.grads, "construct with want_grad=True"
        self._launch(2)
        self._factors()
        self._launch(0)
        self.stream.synchronize()
        return [{p: self.grads[i][off:off + n].reshape(p.shape) for p, off, n in segs} for i, segs in enumerate(self._segments)]

    @property
    def loss(self):
        """The joint loss of the last iteration (:270-343): the models' terms minus the factor terms (0-dim device tensor)."""
        return self.losses[:, 2].sum() - self.factor_losses.sum()

    @property
    def x_tilde(self):
        return self.x_rows[0][self.P:self.P + self.T]

    def snapshot(self):
        super().snapshot()
        self._snap = self._snap + (None if self.xrng is None else self.xrng.clone(),)

    def restore(self):
        super().restore()
        xr = self._snap[-1]
        with torch.cuda.stream(self.stream):
            if xr is not None:
                self.xrng.copy_(xr)
            self._status.zero_()


class _TinyMomentsFn(torch.autograd.Function):
    """(n_models, 2, columns) top-layer moments of a TinyPredictGroup at X; backward: d / d X through every model, one launch."""

    @staticmethod
    def forward(ctx, group, X):
        ctx.group = group
        ctx.save_for_backward(X.detach())
        group.x.copy_(X.detach().reshape(group.T, group.d))
        group._launch(2)
        return group.moments.clone()

    @staticmethod
    def backward(ctx, g):
        group = ctx.group
        (X,) = ctx.saved_tensors
        group.x.copy_(X.reshape(group.T, group.d))      # (another evaluation may have used the group since the forward)
        group.seeds.copy_(g)
        group._launch(3)
        return None, group.gx.sum(0).reshape(X.shape)


class TinyPredictGroup:
    """Predictive moments of SEVERAL fitted small models at the same T test points (eval branch, the layers' fixed
    ``samples``: MFDGP.predict_for_acquisition, mfdgp.py:237-262) in ONE launch -- mobocmf_tiny_elbo_step in its forward-only
    mode -- and their gradient w.r.t. the test points in one more (mode 3): what an acquisition search evaluates hundreds of
    times against constant parameters (JESMOC_MFDGP.py:137-184).  ``moments_at(X)`` -> (n_models, 2, T * S) (T for
    fidelity 0): mean and variance of the top layer's columns, WITHOUT the likelihood noise; differentiable w.r.t. X."""

    _work_bytes_fn = "mobocmf_tiny_work_bytes"

    @staticmethod
    def _fits(model, fidelity, T, d):
        return fits_predict(model, fidelity, T, d, speed_rule=False)

    def __init__(self, models, fidelity, T, d, stream=None):
        lib = _lib.require_device()
        self.models, self.fidelity, self.T, self.d = list(models), fidelity, int(T), int(d)
        n = len(self.models)
        dev = next(self.models[0].parameters()).device
        self.device = dev
        L = fidelity + 1
        self.S = self.models[0].num_samples_for_acquisition if L > 1 else 1
        ncol = self.T * self.S
        self.x = torch.zeros(self.T, d, dtype=torch.float64, device=dev)
        self.moments = torch.zeros(n, 2, ncol, dtype=torch.float64, device=dev)
        self.seeds = torch.zeros(n, 2, ncol, dtype=torch.float64, device=dev)
        self.gx = torch.zeros(n, self.T, d, dtype=torch.float64, device=dev)
        self.host = (_lib.TinyModel * n)()
        self._keep = []
        zrow = torch.zeros(self.T, dtype=torch.float64, device=dev)
        nofid = torch.full((self.T,), -1.0, dtype=torch.float64, device=dev)      # no row is scored: moments only
        self._keep += [zrow, nofid]
        for i, model in enumerate(self.models):
            if not self._fits(model, fidelity, self.T, d) or \
                    (L > 1 and model.num_samples_for_acquisition != self.S):
                raise _lib.MobocmfError("TinyPredictGroup: model %d does not fit the one-launch kernel" % i)
            layers = model._layers()[:L]
            Tm = self.host[i]
            Tm.L, Tm.M, Tm.d, Tm.S, Tm.N = L, layers[0].variational_strategy._inducing_points.shape[0], d, self.S, self.T
            Tm.branch = 1
            Zx = layers[0].variational_strategy._inducing_points.detach().contiguous()
            Tm.x, Tm.y, Tm.fid, Tm.Zx = self.x.data_ptr(), zrow.data_ptr(), nofid.data_ptr(), Zx.data_ptr()
            Tm.kl_scale, Tm.jitter = 0.0, layers[0].variational_strategy.jitter_val
            self._keep.append(Zx)
            for l, layer in enumerate(layers):
                vd = layer.variational_strategy._variational_distribution
                lik = getattr(model, model.name_hidden_layer_likelihood + str(l))
                Tm.rows[l] = self.T
                for s, p in enumerate(_hyper_params(layer)):
                    Tm.raw[l][s] = p.data_ptr()
                Tm.m[l], Tm.L_S[l] = vd.variational_mean.data_ptr(), vd.chol_variational_covar.data_ptr()
                Tm.raw_noise[l] = lik.raw_noise.data_ptr()
                Tm.noise_lo[l], Tm.noise_hi[l] = lik.raw_noise_constraint.lower_bound, lik.raw_noise_constraint.upper_bound
                if l:      # eval_mode's draws: the layer's fixed samples, tiled over the test points (mfdgp_hidden_layer.py:263-270)
                    e = layer.samples.reshape(-1).to(torch.float64).repeat(self.T).contiguous()
                    Tm.eps[l] = e.data_ptr()
                    self._keep.append(e)
            flat, wb = ctypes.c_int64(), ctypes.c_size_t()
            _lib.check(lib.mobocmf_tiny_flat_len(ctypes.byref(Tm), ctypes.byref(flat)), "mobocmf_tiny_flat_len")
            _lib.check(getattr(lib, self._work_bytes_fn)(ctypes.byref(Tm), ctypes.byref(wb)), self._work_bytes_fn)
            work = torch.zeros(wb.value // 8, dtype=torch.float64, device=dev)
            dummy = torch.zeros(2, flat.value, dtype=torch.float64, device=dev)      # (never written: modes 2 / 3 only)
            misc = torch.zeros(8, dtype=torch.int64, device=dev)
            out = torch.zeros(3, dtype=torch.float64, device=dev)
            self._keep += [work, dummy, misc, out]
            Tm.work, Tm.adam_m, Tm.adam_v = work.data_ptr(), dummy[0].data_ptr(), dummy[1].data_ptr()
            Tm.steps_done, Tm.info, Tm.out = misc.data_ptr(), misc[4:].data_ptr(), out.data_ptr()
            Tm.top_mean, Tm.top_var = self.moments[i, 0].data_ptr(), self.moments[i, 1].data_ptr()
            Tm.seed_gmean, Tm.seed_gvar, Tm.seed_scale = self.seeds[i, 0].data_ptr(), self.seeds[i, 1].data_ptr(), 1.0
            Tm.grad = self.gx[i].data_ptr()
        self._dev_table = torch.frombuffer(bytearray(bytes(self.host)), dtype=torch.uint8).to(dev)

    def _launch(self, mode):
        lib = _lib.require_device()
        _lib.check(lib.mobocmf_tiny_elbo_step(ctypes.cast(self.host, ctypes.c_void_p), ctypes.c_void_p(self._dev_table.data_ptr()),
                                              len(self.models), 0.0, 0.9, 0.999, 1e-8, int(mode),
                                              ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                   "mobocmf_tiny_elbo_step")

    def moments_at(self, X):
        X = X.reshape(self.T, self.d)
        if X.requires_grad and torch.is_grad_enabled():
            return _TinyMomentsFn.apply(self, X)
        self.x.copy_(X.detach())
        self._launch(2)
        return self.moments.clone()

    def noise(self, refresh=False):
        """(n_models,) likelihood noise of the top layer (constrained values).  The parameters are constants while a group is
        in use (an acquisition search against fitted models), so the values are formed once; ``refresh=True`` re-reads them."""
        if refresh or self.__dict__.get("_noise") is None:
            with torch.no_grad():
                self._noise = torch.stack([getattr(m, m.name_hidden_layer_likelihood + str(self.fidelity)).noise.reshape(())
                                           for m in self.models])
        return self._noise

    def acquisition_moments(self, X):
        """(mus, vars), each (n_models, T): MFDGP.predict_for_acquisition of every model (noise added, moments over the S
        fixed samples, mfdgp.py:237-262)."""
        mom = self.moments_at(X)
        mean, var = mom[:, 0], mom[:, 1] + self.noise()[:, None]
        if self.fidelity == 0:
            return mean, var
        n = mean.shape[0]
        mu = mean.reshape(n, self.T, self.S)
        mus = mu.mean(2)
        second = (var.reshape(n, self.T, self.S) + mu * mu).mean(2)
        return mus, second - mus * mus
