"""The ELBO step (zero_grad + MFDGP.forward + VariationalELBOMF + backward + Adam, blackbox_mfdgp_fitter.py:161-171)
captured once into a HIP graph and replayed.

A step of one surrogate is ~300 small launches (M x M Cholesky chain, partial reductions, elementwise glue) around a
dozen large GEMMs; issued eagerly from Python that costs ~2 ms of host time per step -- more than the GPU needs for
the small configurations (C1, C2) and half of what it needs at C3.  All library calls only enqueue work on the
caller's stream and use caller-owned memory, so the whole step is capturable; a replay costs tens of microseconds.
Fresh eps is drawn inside the graph (torch's captured Philox state advances on every replay).
"""
import os

import torch


class GraphedELBOStep:
    """step() == one full-batch ELBO step on static (x, y, fidelities).  Falls back to eager with ``use_graph=False``.
    With ``prune_rows`` the rows are reordered once by descending fidelity: ``self.x`` etc. are ``x[self.row_order]``
    (``row_order`` is None when the caller's order was kept); per-row quantities map back through it."""

    exchanges = False      # True in subclasses whose step has a collective between backward and update

    def __init__(self, model, elbo, x, y, fidelities, lr, betas=(0.9, 0.999), eps=1e-8, use_graph=True, stream=None,
                 warmup=3, fixed_eps=None, prune_rows=True):
        self.model, self.elbo = model, elbo
        self.S = model.num_samples_for_training
        self.L = model.num_hidden_layers
        # Dead rows: the ELBO keeps, per layer, the rows of that layer's fidelity (variational_elbo_mf.py:33-38), so a row of
        # fidelity f reaches the loss through layers 0..f only.  The batch is static: order it ONCE by descending fidelity
        # (the full-batch ELBO is a sum over rows, their order is free -- the reference's loader shuffles it every epoch,
        # blackbox_mfdgp_fitter.py:35) and evaluate layer l on the prefix of rows with fidelity >= l (MFDGP.forward(rows=)).
        # Same ELBO and gradients as evaluating every layer at every row; the upper layers' panels shrink to their share.
        self.layer_rows = None
        self.row_order = None      # permutation applied to the caller's rows (None: kept): step.x == x[row_order]
        from ..functional import ELBO_MAX_LAYERS
        if prune_rows and self.L > ELBO_MAX_LAYERS:
            prune_rows = False     # pruned layer outputs need the fused ELBO (<= ELBO_MAX_LAYERS fidelities): reference layout
        if prune_rows and self.L > 1:
            fidv = fidelities.reshape(-1)
            counts = [int((fidv >= l).sum()) for l in range(self.L)]
            if counts[-1] >= 1 and counts[0] == fidv.numel():
                order = torch.argsort(fidv, descending=True, stable=True)
                x, y, fidelities = x[order].contiguous(), y[order].contiguous(), fidelities[order].contiguous()
                if fixed_eps is not None:      # given for the batch as passed in (N*S per layer): follow the rows
                    N = fidv.numel()
                    fixed_eps = [None if e is None else e.reshape(N, self.S)[order][:counts[l]].reshape(-1).contiguous()
                                 for l, e in enumerate(fixed_eps)]
                self.layer_rows = counts
                self.row_order = order
        self.x, self.y, self.fid = x, y, fidelities
        self.use_graph = use_graph
        self.stream = stream if stream is not None else torch.cuda.Stream(device=x.device)
        params = [p for p in model.parameters()]
        from ..functional import FusedAdam
        if os.environ.get("MOBOCMF_TORCH_ADAM"):      # A/B knob: torch's capturable Adam (seven foreach launches)
            self.optimizer = torch.optim.Adam(params, lr=lr, betas=betas, eps=eps, capturable=True)
        else:
            self.optimizer = FusedAdam(params, lr=lr, betas=betas, eps=eps)     # one launch; step count on the device
        self.loss = torch.zeros((), dtype=torch.float64, device=x.device)
        self._minus_one = torch.full((), -1.0, dtype=torch.float64, device=x.device)
        self.kl = torch.zeros((), dtype=torch.float64, device=x.device)
        self.graph = None
        self.graph_update = None
        self._snap = None
        self.fixed_eps = fixed_eps     # list (eps[l] for layer l >= 1) reused every step: deterministic tests
        model.set_check_pd(False)      # no host sync inside the step; call check() when a verdict is needed
        model.clear_kl_cache()         # an older graph would pin AccumulateGrad nodes to another stream (capture-illegal)
        for layer in model._layers():  # the layers' eps streams get their seeds here, in a fixed order, eager or captured
            layer._rng(x.device)
        if use_graph:
            self._capture(warmup)

    def _fwd_bwd(self):
        self.optimizer.zero_grad(set_to_none=True)
        # eps: explicit (deterministic tests) or drawn inside the layers' propagation launches (no torch generator in the
        # captured graph: its replay support costs two fill launches per replay, the draw a third)
        out = self.model(self.x, eps=self.fixed_eps, rows=self.layer_rows)
        res = self.elbo(out, self.y.T, self.fid)
        # d(-ELBO): the sign goes in as the upstream gradient (no negation node, no ones fill, no negation backward)
        res[0].backward(gradient=self._minus_one)
        neg = getattr(self.elbo, "last_neg_elbo", None)
        if neg is not None:      # the fused ELBO launch wrote -elbo next to elbo: no negation / copy launches
            self.loss, self.kl = neg, res[1].detach()
        else:
            torch.neg(res[0].detach(), out=self.loss)
            self.kl.copy_(res[1].detach())
        self.model.clear_kl_cache()

    def _exchange(self):
        """Between backward and the update; a no-op for a surrogate that lives on one GPU (RowShardedELBOStep
        all-reduces the gradient bucket here)."""

    def _update(self):
        self.optimizer.step()

    def _eager(self):
        self._fwd_bwd()
        self._exchange()
        self._update()

    def _capture(self, warmup):
        from .. import functional as F
        cur = torch.cuda.current_stream(self.x.device)
        self.stream.wait_stream(cur)
        F.take_capture_pins()          # pins left behind by a capture that aborted elsewhere are not this graph's
        try:
            self._capture_on_stream(warmup)
        finally:
            # the graph replays on these buffers: they live as long as it does (also taken when the capture raised, so that
            # a failed capture cannot leak its pins into the next step's list)
            self._pinned_scratch = F.take_capture_pins()
        cur.wait_stream(self.stream)

    def _capture_on_stream(self, warmup):
        with torch.cuda.stream(self.stream):
            # the side-stream warm-up also sizes the per-stream scratch arena and the optimizer state
            snapshot = [p.detach().clone() for p in self.model.parameters()]
            # the layers' eps streams (seed, call counter): warm-up draws must not count either -- the first replay then
            # draws exactly what the first eager step would have drawn
            self._rng_snapshot = [(l, l._rng(self.x.device).clone()) for l in self.model._layers()]
            for _ in range(warmup):
                self._eager()
            self._reset_after_warmup(snapshot)
            self.graph = torch.cuda.CUDAGraph()
            if not self.exchanges:
                with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                    self._eager()
            else:       # the collective stays outside: graph | all-reduce | graph
                with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                    self._fwd_bwd()
                self._exchange()
                self.graph_update = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_update, stream=self.stream, capture_error_mode="thread_local"):
                    self._update()
                self._reset_after_warmup(snapshot)     # the capture pass above ran the exchange + nothing else for real

    def _reset_after_warmup(self, snapshot):
        with torch.no_grad():           # warm-up steps must not count as training
            for p, s0 in zip(self.model.parameters(), snapshot):
                p.copy_(s0)
            for layer, st in getattr(self, "_rng_snapshot", []):
                layer._rng(st.device).copy_(st)
            for st in self.optimizer.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        if not self.exchanges:
            self.optimizer.zero_grad(set_to_none=True)

    def retire(self):
        """Drop the graph and this step's entry in the scratch arena (the fitter calls it when a training phase ends)."""
        from .. import functional as F
        self.stream.synchronize()
        self.graph = self.graph_update = None
        self._pinned_scratch = None
        F.release_scratch(self.stream)

    def step(self):
        """Enqueues one step on ``self.stream``; ``self.loss`` / ``self.kl`` hold the step's -ELBO and scaled KL."""
        with torch.cuda.stream(self.stream):
            if self.graph is not None:
                self.graph.replay()
                if self.exchanges:
                    self._exchange()
                    self.graph_update.replay()
            else:
                self._eager()
        return self.loss, self.kl

    # ------------------------------------------------------------------ snapshot / fallback
    def snapshot(self):
        """Clone of parameters + optimizer state (taken at points where check() passed)."""
        with torch.cuda.stream(self.stream):
            self._snap = ([p.detach().clone() for p in self.model.parameters()],
                          [{k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                           for st in self.optimizer.state.values()])

    def restore_and_go_eager(self):
        """After a failed Cholesky inside a replayed step (no host-side jitter retry is possible there): roll back to
        the last good snapshot and continue eagerly with the psd_safe_cholesky jitter ladder (check_pd=True), which is
        what the reference does at every step."""
        self.stream.synchronize()
        ps, sts = self._snap
        with torch.no_grad():
            for p, s0 in zip(self.model.parameters(), ps):
                p.copy_(s0)
            for st, s0 in zip(self.optimizer.state.values(), sts):
                for k, v in st.items():
                    if torch.is_tensor(v):
                        v.copy_(s0[k])
        self.graph = None
        self.model.set_check_pd(True)

    def check(self):
        """Synchronising: raises if a Cholesky of the last step failed or the loss is not finite."""
        from .. import functional as F
        from ..layers.mfdgp_hidden_layer import NotPSDError
        self.stream.synchronize()
        for layer in self.model._layers():
            if layer._info is not None:
                pivot = F.check_info(layer._info)
                F.raise_if_abandoned(pivot, "layer %d" % layer.num_layer)
                if pivot != 0:
                    raise NotPSDError("K_mm not positive definite in layer %d" % layer.num_layer)
        if not bool(torch.isfinite(self.loss)):
            raise FloatingPointError("non-finite ELBO")


class _ModelGroup:
    """The models of a joint step seen as one (parameters / layers / housekeeping of GraphedELBOStep)."""

    def __init__(self, models):
        self.models = list(models)

    def parameters(self):
        for m in self.models:
            yield from m.parameters()

    def _layers(self):
        for m in self.models:
            yield from m._layers()

    def clear_kl_cache(self):
        for m in self.models:
            m.clear_kl_cache()

    def set_check_pd(self, value):
        for m in self.models:
            m.set_check_pd(value)


class GraphedConditionedStep(GraphedELBOStep):
    """One iteration of the conditioned training (blackbox_mfdgp_fitter.py:245-354: fresh x~ ~ U[0,1]^(10 x d), the joint
    loss over ALL surrogates, one Adam) captured into a HIP graph.  ``fitter.conditioned_loss`` must be capture-safe
    (no host reads); x~ is drawn inside the graph, so every replay sees new points."""

    def __init__(self, fitter, lr, betas=(0.9, 0.999), eps=1e-8, use_graph=True, stream=None, warmup=3, n_tilde=10,
                 fixed_x_tilde=None):
        self.fitter = fitter
        self.fixed_x_tilde = fixed_x_tilde      # deterministic tests: the same x~ at every iteration
        models = [h.mfdgp for _, _, h in fitter._handlers()]
        self.model = _ModelGroup(models)
        dev = fitter.pareto_set.device
        self.device, self.d, self.n_tilde = dev, fitter.pareto_set.shape[1], n_tilde
        self.use_graph = use_graph
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        from ..functional import FusedAdam
        self.optimizer = FusedAdam(list(self.model.parameters()), lr=lr, betas=betas, eps=eps)
        self.loss = torch.zeros((), dtype=torch.float64, device=dev)
        self.kl = torch.zeros((), dtype=torch.float64, device=dev)
        self.graph = self.graph_update = self._snap = None
        self.x = fitter.pareto_set            # (only its device is used by the base class)
        self.model.set_check_pd(False)
        self.model.clear_kl_cache()
        for layer in self.model._layers():
            layer._rng(dev)
        if use_graph:
            self._capture(warmup)

    def _fwd_bwd(self):
        self.optimizer.zero_grad(set_to_none=True)
        x_tilde = self.fixed_x_tilde if self.fixed_x_tilde is not None else \
            torch.rand(self.n_tilde, self.d, dtype=torch.float64, device=self.device)
        from .. import parallel
        if self.fixed_x_tilde is None and parallel.world()[1] > 1:
            parallel.broadcast_(x_tilde)      # sharded surrogates: the gathered moments must refer to the SAME points
        loss = self.fitter.conditioned_loss(x_tilde)
        loss.backward()
        self.loss.copy_(loss.detach())
        self.model.clear_kl_cache()
