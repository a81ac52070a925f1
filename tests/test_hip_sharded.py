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
