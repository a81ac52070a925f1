"""The ELBO step of MID-SIZE surrogates (M <= 128 inducing points) as ONE launch for a whole group of models, several
workgroups per surrogate: mobocmf_coop_elbo_step (csrc/coop_step.hip).

The reference's own BO loop trains M = N from 15 to 75 points (examples/toy_synthetic_2D_JESMOCMF/...py:25,102-103,305-331 with
mfdgp.py:295-298); the one-workgroup kernel (util/tiny_step.py) covers M <= 32, the layer path costs 57 launches per step there.
``CoopELBOStep`` has the surface of ``TinyELBOStep`` / ``GraphedELBOStep`` that the fitter's loop uses (step / check / snapshot /
restore / losses / export_adam_state); ``CoopConditionedStep`` is the conditioned iteration (blackbox_mfdgp_fitter.py:245-354).
"""
import ctypes

import torch

from .. import _lib
from . import tiny_step as TS

MAX_COLUMNS = 16384      # rows[l] * S per layer this binding accepts


def eligible(model, x, fidelities):
    """The structural limits of the cooperative kernel: <= 3 layers sharing one set of <= 128 inducing inputs, d <= 8, softplus /
    Interval constraints, float64 parameters on the GPU (``tiny_step.eligible`` with this kernel's limits; no speed rule: the
    launch beats the layer path's ~57 launches per step wherever it applies)."""
    return TS.eligible(model, x, fidelities, speed_rule=False, max_m=_lib.COOP_MAX_M, max_columns=MAX_COLUMNS)


MAX_WORTHWHILE_COLUMNS = 2048      # summed over the layers: beyond, the layer path's grid-filling launches win (tools/coop_sweep.py)


def worthwhile(model, x, fidelities):
    """``eligible`` and small enough for the launch to beat the layer path (profiles/r05_coop_step.txt: BASELINE config 2 --
    M = 128, 512 + 1024 columns -- 1.3x; the reference's loop sizes 2-2.5x; a single workgroup round per phase is what wins, so
    the gate is the number of columns)."""
    if not eligible(model, x, fidelities):
        return False
    L = len(model._layers())
    S = model.num_samples_for_training
    fidv = fidelities.reshape(-1)
    cols = sum(int((fidv >= l).sum()) * (S if l else 1) for l in range(L))
    return cols <= MAX_WORTHWHILE_COLUMNS


class _CoopLaunch:
    """The launch of mobocmf_coop_elbo_step for the descriptor table a Tiny* step object built."""
    _work_bytes_fn = "mobocmf_coop_work_bytes"
    wgs_per_model = 0      # 0: chosen by the library from the widest phase

    @staticmethod
    def _eligible(model, x, fid, force):
        return eligible(model, x, fid)

    @staticmethod
    def _eligible_conditioned(model, x, fid):
        return eligible(model, x, fid)

    def _sync_words(self):
        sw = self.__dict__.get("_sync")
        if sw is None:
            sw = torch.zeros(16 * (len(self.models) + 1), dtype=torch.int64, device=self.device)
            self._sync = sw
            self._order_after_setup()      # (the zero fill runs on the current stream, the launches on self.stream)
        return sw

    def _launch(self, mode):
        lib = _lib.require_device()
        used = ctypes.c_int32(0)
        _lib.check(lib.mobocmf_coop_elbo_step(ctypes.cast(self.host, ctypes.c_void_p), ctypes.c_void_p(self._dev_table.data_ptr()),
                                              len(self.models), int(self.wgs_per_model), ctypes.c_void_p(self._sync_words().data_ptr()),
                                              self.lr, self.betas[0], self.betas[1], self.eps, int(mode), ctypes.byref(used),
                                              ctypes.c_void_p(self.stream.cuda_stream)),
                   "mobocmf_coop_elbo_step")
        self.wgs_used = used.value


class CoopELBOStep(_CoopLaunch, TS.TinyELBOStep):
    """``step()`` == one full-batch ELBO step of EVERY model of the group, one launch (see ``TinyELBOStep`` for the arguments)."""

    def restore(self):
        super().restore()
        # a barrier that was abandoned leaves its arrival counter out of step: start the counters afresh
        with torch.cuda.stream(self.stream):
            self._sync_words().zero_()


class CoopConditionedStep(_CoopLaunch, TS.TinyConditionedStep):
    """One iteration of the conditioned training in ONE launch (mode 4), or forward-only launch + factor launches + step launch."""

    def _issue(self):
        if self.one_launch and self.T <= 256:
            try:
                self._launch(4)
                return
            except _lib.MobocmfError:
                self.one_launch = False
        self._launch(2)
        self._factors()
        self._launch(1)

    def restore(self):
        super().restore()
        with torch.cuda.stream(self.stream):
            self._sync_words().zero_()


MAX_PREDICT_COLUMNS = 4096      # T * S per layer beyond which the layer path (frozen chains) is the better search engine


def fits_predict(model, fidelity, T, d):
    """``tiny_step.fits_predict`` with the cooperative kernel's limits (M <= 128) and its column bound."""
    S = model.num_samples_for_acquisition if fidelity > 0 else 1
    return T * S <= MAX_PREDICT_COLUMNS and TS.fits_predict(model, fidelity, T, d, speed_rule=False, max_m=_lib.COOP_MAX_M)


class CoopPredictGroup(TS.TinyPredictGroup):
    """``TinyPredictGroup`` for mid-size models (32 < M <= 128): predictive moments of several fitted models at the same T test
    points in ONE cooperative launch (mode 2), their gradient w.r.t. the test points in one more (mode 3) -- the acquisition
    search of the reference's later BO iterations (JESMOC_MFDGP.py:137-184 against M = N = 33 ... 75 surrogates)."""
    _work_bytes_fn = "mobocmf_coop_work_bytes"
    wgs_per_model = 0
    _frozen = False            # inside freeze() ... thaw(): the models' parameters do not change between launches
    _chain_ready = False       # ... and a launch since freeze() has left their chains (L^-1, U, a) in the workspaces

    @staticmethod
    def _fits(model, fidelity, T, d):
        return fits_predict(model, fidelity, T, d)

    def freeze(self):
        """The parameters are constants until ``thaw()`` (an acquisition search, JESMOC_MFDGP._optimize): the first launch
        computes the chains, the following ones reuse them (MOBOCMF_STEP_CHAIN_VALID)."""
        if not self._frozen:
            self._frozen, self._chain_ready = True, False

    def thaw(self):
        self._frozen = self._chain_ready = False

    def _launch(self, mode):
        lib = _lib.require_device()
        sw = self.__dict__.get("_sync")
        if sw is None:
            sw = self._sync = torch.zeros(16 * (len(self.models) + 1), dtype=torch.int64, device=self.device)
        if self._frozen and self._chain_ready:
            mode = int(mode) | _lib.STEP_CHAIN_VALID
        self._chain_ready = self._frozen
        used = ctypes.c_int32(0)
        _lib.check(lib.mobocmf_coop_elbo_step(ctypes.cast(self.host, ctypes.c_void_p), ctypes.c_void_p(self._dev_table.data_ptr()),
                                              len(self.models), int(self.wgs_per_model), ctypes.c_void_p(sw.data_ptr()),
                                              0.0, 0.9, 0.999, 1e-8, int(mode), ctypes.byref(used),
                                              ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                   "mobocmf_coop_elbo_step")
        self.wgs_used = used.value
