"""One whole BO iteration through the mirrored caller surface (fitter -> RFF samples + MOOP -> conditioned fit -> JES
acquisition search), every model evaluation on the HIP path: the glue between the rows N1-N3 of SURVEY 8(f)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_one_bo_iteration_end_to_end():
    from bo_iteration_toy2d import blackboxes, run
    fitter, acq, cand, fidelity = run(epochs=60, cond_iters=30, acq_iters=8, grid=40, seed=0, verbose=False)
    ps, pf = fitter.pareto_set.cpu().numpy(), fitter.pareto_front.cpu().numpy()
    assert ps.ndim == 2 and ps.shape[1] == 2 and 1 <= ps.shape[0] <= 10 and pf.shape == (ps.shape[0], 2)
    assert ps.min() >= 0.0 and ps.max() <= 1.0
    # the stored solution is what the stored samples say: front = objective samples at the set, constraints feasible
    for j, s in enumerate(fitter.samples_objs):
        assert np.allclose(s(ps), pf[:, j], atol=1e-9)
    for s in fitter.samples_cons:
        assert np.all(s(ps) >= -1e-6)
    # non-dominated among themselves
    for p in pf:
        assert np.all(pf <= p, axis=1).sum() == 1
    assert cand.shape == (2,) and 0.0 <= float(cand.min()) and float(cand.max()) <= 1.0 and fidelity in (0, 1)
    # conditioned and unconditioned surrogates now differ, the unconditioned copy kept the fitted state
    mu = acq.blackbox_mfdgp_fitter_uncond.get_model("obj1").hidden_layer_1.variational_strategy
    mc = acq.blackbox_mfdgp_fitter_cond.get_model("obj1").hidden_layer_1.variational_strategy
    dm = (mu._variational_distribution.variational_mean - mc._variational_distribution.variational_mean).abs().max()
    assert float(dm) > 0.0
    X = torch.rand(16, 2, dtype=torch.float64, device="cuda")
    v = acq.coupled_acq(X, fidelity=1)
    assert v.shape == (16,) and bool(torch.isfinite(v).all()) and float(v.min()) >= 0.0


def test_fitter_and_acquisition_survive_dill_round_trip():
    """The reference's examples pickle the fitter and the acquisition object between stages
    (example_acquisition_mfdgp_forrester.py:116-118,140-142): same values after the round trip."""
    import dill
    from bo_iteration_toy2d import run
    fitter, acq, _, _ = run(epochs=30, cond_iters=10, acq_iters=3, grid=30, seed=1, verbose=False)
    acq2 = dill.loads(dill.dumps(acq))
    fitter2 = dill.loads(dill.dumps(fitter))
    X = torch.rand(9, 2, dtype=torch.float64, device="cuda")
    for f in (0, 1):
        assert torch.equal(acq2.coupled_acq(X, fidelity=f), acq.coupled_acq(X, fidelity=f))
    m1, v1 = fitter.get_model("obj2").predict_for_acquisition(X, 1)       # fixed samples: deterministic
    m2, v2 = fitter2.get_model("obj2").predict_for_acquisition(X, 1)
    assert torch.equal(m1, m2) and torch.equal(v1, v2)
    assert np.allclose(fitter2.samples_objs[0](X.cpu().numpy()), fitter.samples_objs[0](X.cpu().numpy()), atol=1e-12)


def test_two_bo_iterations_grow_the_data_set():
    from bo_iteration_toy2d import loop
    x, fid, hist = loop(iters=2, seed=3, verbose=False, epochs=40, cond_iters=20, acq_iters=5, grid=30)
    assert x.shape == (22, 2) and fid.shape == (22,) and len(hist) == 2
    assert np.all((x >= 0.0) & (x <= 1.0)) and set(np.unique(fid)) <= {0.0, 1.0}
