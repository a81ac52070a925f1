"""SURVEY 8(e) level 2 on the GPU: ONE surrogate with its batch rows sharded over two ranks (two processes on the one
card, gloo transport -- RCCL refuses two ranks on one device) must follow the single-process full-batch trajectory:
same -ELBO per step, same parameters after k steps, and the two replicas bit-identical."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from mobocmf_amd.util import synthetic
from tests.helpers import to_t

pytestmark = pytest.mark.gpu
DEV = "cuda"
CFG = dict(d=3, L=2, M=24, N=75, S=2, seed=11)       # ragged: 38 + 37 rows
STEPS = 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(sharded, use_graph):
    from mobocmf_amd.mlls import VariationalELBOMF
    from mobocmf_amd.parallel import RowShardedELBOStep
    from mobocmf_amd.util.graphed_step import GraphedELBOStep
    from tests.test_hip_model import build_model
    prob = synthetic.make_problem(**CFG)
    t = lambda a: to_t(a).to(DEV)
    model = build_model(prob, S_train=CFG["S"])
    elbo = VariationalELBOMF(model, CFG["N"], CFG["L"])
    cls = RowShardedELBOStep if sharded else GraphedELBOStep
    g = cls(model, elbo, t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None], lr=1e-2, use_graph=use_graph,
            fixed_eps=[None, t(prob["eps"][1])])
    losses = []
    for _ in range(STEPS):
        l, _ = g.step()
        g.stream.synchronize()
        losses.append(float(l))
    g.check()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    _run.names = [(n, p.numel()) for n, p in model.named_parameters()]
    return losses, flat


def _where(idx):
    off = 0
    for n, k in _run.names:
        if idx < off + k:
            return "%s[%d]" % (n, idx - off)
        off += k


def _worker(rank, world, port, use_graph, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    losses, flat = _run(True, use_graph)
    q.put((rank, losses, flat.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graphs"])
def test_row_sharded_step_follows_full_batch_trajectory(use_graph):
    ref_losses, ref_flat = _run(False, use_graph)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, use_graph, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, l0, f0), (_, l1, f1) = res
    assert l0 == l1 and (f0 == f1).all()                                   # replicas stay bit-identical
    scale = float(ref_flat.abs().max())
    diff = (torch.as_tensor(f0) - ref_flat).abs()
    assert float(diff.max()) / scale < 1e-9, (_where(int(diff.argmax())), float(diff.max()), ref_losses, l0)
    for a, b in zip(l0, ref_losses):
        assert abs(a - b) <= 1e-9 * abs(b)
    assert l0[-1] < l0[0]


# ---------------------------------------------------------------------------------------------------------------
# Conditioned training with the SURROGATES sharded over ranks (SURVEY 8(e) level 1): 2 objectives + 1 constraint on 2
# ranks -- the ragged layout round-robin sharding produces (rank 0: obj 0 + the constraint, rank 1: obj 1, no constraint).
# Every rank's loss must differentiate to the single-process gradients for the models it owns: that needs the Pareto
# front's columns and the thresholds indexed GLOBALLY and the omega-factor exchange to work on ragged / empty shards.
# ---------------------------------------------------------------------------------------------------------------
def _cond_setup(names_mine):
    import numpy as np
    from torch.utils.data import TensorDataset
    from mobocmf_amd.mlls import VariationalELBOMF
    from mobocmf_amd.util.blackbox_mfdgp_fitter import BlackBoxMFDGPFitter, MFDGPHandler
    N, P, T, d = 12, 5, 10, 2
    layout = [("obj0", False, 0), ("obj1", False, 1), ("con0", True, 0)]      # name, is_constraint, global index
    fitter = BlackBoxMFDGPFitter(2, N, device=DEV)
    fitter.verbose = False
    g = torch.Generator().manual_seed(0)
    pareto_set = torch.rand(P, d, dtype=torch.float64, generator=g)
    pareto_front = torch.randn(P, 2, dtype=torch.float64, generator=g) * 0.5
    x_tilde = torch.rand(T, d, dtype=torch.float64, generator=g)
    eps_all = {}
    for o, (name, is_con, gi) in enumerate(layout):
        e = torch.randn(N + P + T, dtype=torch.float64, generator=g)       # drawn for every black-box: same on all ranks
        if name not in names_mine:
            continue
        prob = synthetic.make_problem(d=d, L=2, M=8, N=N, S=1, output=o, seed=o)
        prob["noise"] = [np.array(1e-2), np.array(2e-2)]
        model = synthetic.model_from_problem(prob, num_samples_for_training=1, device=DEV)
        h = MFDGPHandler.__new__(MFDGPHandler)
        h.mfdgp, h.num_data, h.num_fidelities, h.batch_size, h.global_index = model, N, 2, N, gi
        h.elbo = VariationalELBOMF(model, N, 2)
        t = lambda a: to_t(a).to(DEV)
        h.train_dataset = TensorDataset(t(prob["x"]), t(prob["y"])[:, None], t(prob["fid"])[:, None])
        h.iter_train_loader = None
        (fitter.mfdgp_handlers_cons if is_con else fitter.mfdgp_handlers_objs)[name] = h
        local_i = (len(fitter.mfdgp_handlers_cons) if is_con else len(fitter.mfdgp_handlers_objs)) - 1
        eps_all[("CON" if is_con else "OBJ", local_i)] = [None, e.to(DEV)]
    fitter.num_obj, fitter.num_con = len(fitter.mfdgp_handlers_objs), len(fitter.mfdgp_handlers_cons)
    fitter.thresholds_cons = torch.tensor([0.1] * fitter.num_con, dtype=torch.float64)
    fitter.set_global_constraint_thresholds([0.1])
    fitter.set_pareto_solution(pareto_set, pareto_front)
    return fitter, x_tilde.to(DEV), eps_all


def _cond_grads(fitter, x_tilde, eps_all):
    loss = fitter.conditioned_loss(x_tilde, eps=eps_all)
    loss.backward()
    out = {}
    for name, h in list(fitter.mfdgp_handlers_objs.items()) + list(fitter.mfdgp_handlers_cons.items()):
        for l in range(2):
            vd = getattr(h.mfdgp, f"hidden_layer_{l}").variational_strategy._variational_distribution
            out[(name, l, "m")] = vd.variational_mean.grad.detach().cpu()
            out[(name, l, "L_S")] = vd.chol_variational_covar.grad.detach().cpu()
    return float(loss), out


def _cond_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mobocmf_amd import parallel
    mine, owner = parallel.shard_blackboxes(["obj0", "obj1", "con0"])
    fitter, x_tilde, eps_all = _cond_setup(mine)
    loss, grads = _cond_grads(fitter, x_tilde, eps_all)
    q.put((rank, mine, loss, {k: v.numpy() for k, v in grads.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_conditioned_loss_with_sharded_surrogates_matches_single_process():
    ref_loss, ref = _cond_grads(*_cond_setup(["obj0", "obj1", "con0"]))
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cond_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] == ["obj0", "con0"] and res[1][1] == ["obj1"]              # ragged: rank 1 holds no constraint
    seen = set()
    for _, _, loss, grads in res:
        assert loss == loss and abs(loss) < 1e12
        for key, g in grads.items():
            r = ref[key]
            err = float((torch.as_tensor(g) - r).abs().max() / r.abs().max().clamp_min(1e-300))
            assert err < 1e-9, (key, err)
            seen.add(key)
    assert seen == set(ref)
