"""N > 1 path on CPU: world-size-2 gloo processes exercise the surrogate sharding table and the single all-gather
exchange (posterior moments / coupled acquisition) of mobocmf_amd.parallel."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mobocmf_amd import parallel
    names = ["obj1", "obj2", "con1", "con2", "con3"]
    mine, owner = parallel.shard_blackboxes(names)
    # every rank holds 2 surrogates' (mus, vars) on a 7-point grid; values encode (rank, surrogate)
    local = torch.stack([torch.full((2, 7), float(10 * rank + k), dtype=torch.float64) for k in range(2)])
    gathered = parallel.all_gather_moments(local)
    acq_local = torch.full((2, 7), float(rank + 1), dtype=torch.float64)
    total = parallel.coupled_acquisition(acq_local)
    # omega-factor coupling: own rows keep autograd history, the other rank's rows are constants
    fm = torch.full((1, 3), float(rank + 1), dtype=torch.float64, requires_grad=True)
    mixed = parallel.gather_with_local_grad(fm, fm * 2, fm[:0], fm[:0])
    mixed[0].prod(0).sum().backward()
    q.put((rank, mine, owner, gathered.shape, gathered[:, 0, 0].tolist(), total.tolist(),
           mixed[0][:, 0].tolist(), fm.grad[0].tolist(), tuple(mixed[2].shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_exchange():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, owner0, shape0, g0, t0, mx0, gr0, cs0), (r1, mine1, owner1, shape1, g1, t1, mx1, gr1, cs1) = res
    assert mx0 == mx1 == [1.0, 2.0]                      # rank order, own row in place
    assert gr0 == [2.0] * 3 and gr1 == [1.0] * 3       # d(prod)/d(own) = the other rank's (constant) value
    assert cs0 == (0, 3)
    assert owner0 == owner1 and sorted(mine0 + mine1) == sorted(owner0)
    assert mine0 == ["obj1", "con1", "con3"] and mine1 == ["obj2", "con2"]
    assert tuple(shape0) == (4, 2, 7) and g0 == g1 == [0.0, 1.0, 10.0, 11.0]      # rank order, identical everywhere
    assert t0 == t1 == [6.0] * 7                                                     # 2*1 + 2*2


def _ragged_worker(rank, world, port, q):
    """Unequal shards, one rank without any constraint (2 objectives + 1 constraint... on 2 ranks: the layout round-robin
    sharding produces): every rank must still enter every collective, and rows must come back in GLOBAL order."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mobocmf_amd import parallel
    T = 3
    # objectives: rank 0 holds global 0 and 2, rank 1 holds global 1; the only constraint lives on rank 1
    obj_idx = [0, 2] if rank == 0 else [1]
    con_idx = [] if rank == 0 else [0]
    fm = torch.stack([torch.full((T,), float(10 + g), dtype=torch.float64) for g in obj_idx]).requires_grad_(True)
    cm = torch.stack([torch.full((T,), float(50 + g), dtype=torch.float64) for g in con_idx]).requires_grad_(True) \
        if con_idx else torch.zeros((0, T), dtype=torch.float64, requires_grad=True)
    afm, afv, acm, acv = parallel.gather_with_local_grad(fm, fm * 2, cm, cm * 3, obj_idx, con_idx)
    (afm.prod(0).sum() + acm.sum()).backward()
    parts = parallel.all_gather_ragged(torch.full((rank, 2), float(rank)))          # 0 rows on rank 0, 1 row on rank 1
    acq = parallel.coupled_acquisition(torch.full((len(obj_idx) + len(con_idx), T), 1.0, dtype=torch.float64))
    x = torch.full((2,), float(rank + 5))
    parallel.broadcast_(x)
    # a layout that changes INSIDE the process group without a reset (here: on rank 0 only): every rank still enters the one
    # per-step collective, and every rank gets the same error afterwards -- no stale slicing, nobody left waiting
    stale = None
    try:
        parallel.gather_with_local_grad(fm[:1] if rank == 0 else fm, fm[:1] if rank == 0 else fm, cm, cm,
                                        obj_idx[:1] if rank == 0 else obj_idx, con_idx)
    except ValueError as err:
        stale = "differs from the one negotiated" in str(err) and "[0]" in str(err)
    # errors of the negotiation itself (after reset_gather_plans() on every rank, the documented way to change a layout):
    bad = None
    parallel.reset_gather_plans()
    try:
        parallel.gather_with_local_grad(fm, fm, cm, cm, [0] * len(obj_idx), con_idx)   # not a permutation
    except ValueError as err:
        bad = str(err)
    # one rank passes indices, the other does not: the SAME error on both ranks, no rank left waiting in a collective
    mismatch = None
    parallel.reset_gather_plans()
    try:
        parallel.gather_with_local_grad(fm[:1], fm[:1], cm[:0], cm[:0], [0] if rank == 0 else None, [])
    except ValueError as err:
        mismatch = "either every rank" in str(err)
    # ONE rank passes a malformed index list (two indices for one row): it says so inside the first collective, and BOTH ranks
    # raise (round 3: the bad rank raised before the collective and the other one waited in it for ever)
    badlen = None
    parallel.reset_gather_plans()
    try:
        parallel.gather_with_local_grad(fm[:1], fm[:1], cm[:0], cm[:0], [0, 1] if rank == 1 else [0], [])
    except ValueError as err:
        badlen = "one global index per local row" in str(err) and "[1]" in str(err)
    # the negotiated layout is cached: a second step costs one collective and gives the same rows
    parallel.reset_gather_plans()
    cfm, _, _, _ = parallel.gather_with_local_grad(fm, fm * 2, cm, cm * 3, obj_idx, con_idx)
    n_plans = len(parallel._gather_plans)
    bfm, _, _, _ = parallel.gather_with_local_grad(fm, fm * 2, cm, cm * 3, obj_idx, con_idx)
    cached = len(parallel._gather_plans) == n_plans == 1 and torch.equal(bfm.detach(), afm.detach()) and \
        torch.equal(cfm.detach(), afm.detach()) and bool(stale) and bool(badlen)
    q.put((rank, afm[:, 0].tolist(), afv[:, 0].tolist(), acm[:, 0].tolist(), acv[:, 0].tolist(), fm.grad[:, 0].tolist(),
           None if cm.grad is None else cm.grad.reshape(-1).tolist(), [tuple(p.shape) for p in parts], acq.tolist(),
           x.tolist(), bad is not None and bool(mismatch) and cached))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_ragged_and_empty_shards_do_not_deadlock():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, afm, afv, acm, acv, gfm, gcm, shapes, acq, x, bad in res:
        assert afm == [10.0, 11.0, 12.0] and afv == [20.0, 22.0, 24.0]          # global objective order on every rank
        assert acm == [50.0] and acv == [150.0]
        assert shapes == [(0, 2), (1, 2)] and acq == [4.0] * 3 and x == [5.0, 5.0] and bad
    assert res[0][5] == [11.0 * 12.0, 10.0 * 11.0] and res[1][5] == [10.0 * 12.0]      # d prod / d own rows only
    assert res[0][6] in (None, []) and res[1][6] == [1.0] * 3


def test_world_size_1_degenerates_to_identity():
    from mobocmf_amd import parallel
    x = torch.arange(6, dtype=torch.float64).reshape(1, 2, 3)
    assert torch.equal(parallel.all_gather_moments(x), x)
    mine, owner = parallel.shard_blackboxes(["a", "b"], rank=0, world_size=1)
    assert mine == ["a", "b"]


def _sharded_worker(rank, world, port, q):
    """Level 2 of SURVEY 8(e): rows of ONE surrogate sharded over the ranks; the host logic (shard_rows, GradBucket,
    the single all-reduce) is exercised with the oracle as the local compute (the HIP path needs a GPU)."""
    import sys
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import oracle_state, small_problem, state_leaves, to_t
    from mobocmf_amd import parallel
    from oracle import mfdgp_oracle as O
    prob = small_problem(d=2, L=2, M=6, N=11, S=3, seed=3)          # ragged: 11 rows over 2 ranks
    x, y, fid = to_t(prob["x"]), to_t(prob["y"]).reshape(-1), to_t(prob["fid"]).reshape(-1)
    N, S = x.shape[0], 3
    eps = [None, to_t(prob["eps"][1]).reshape(-1)]
    # full batch, computed redundantly on every rank: the expected result
    st_full = oracle_state(prob, requires_grad=True)
    e_full, kl_full = O.elbo(st_full, x, y, fid, eps=eps, S=S)
    (-e_full).backward()
    g_full = torch.cat([p.grad.reshape(-1) for p in state_leaves(st_full)])
    # sharded
    st = oracle_state(prob, requires_grad=True)
    leaves = state_leaves(st)
    bucket = parallel.GradBucket(leaves, extra=2)
    idx = parallel.shard_rows(N)
    eps_loc = [None, eps[1].reshape(N, S)[idx].reshape(-1)]
    bucket.zero_()
    e_loc, kl_loc = O.elbo(st, x[idx], y[idx], fid[idx], eps=eps_loc, S=S, num_data=N)
    (-e_loc).backward()
    bucket.extra[0] = -e_loc.detach()
    bucket.extra[1] = kl_loc.detach()
    bucket.all_reduce()
    n = g_full.numel()
    q.put((rank, idx.tolist(), float((bucket.flat[:n] - g_full).abs().max() / g_full.abs().max()),
           float(bucket.extra[0] + e_full.detach()), float(bucket.extra[1] - kl_full.detach()),
           all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in leaves)))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_row_sharded_gradients_sum_to_full_batch():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6, 8, 10] and res[1][1] == [1, 3, 5, 7, 9]
    for _, _, gerr, de, dkl, views in res:
        assert gerr < 1e-12 and abs(de) < 1e-10 and abs(dkl) < 1e-12 and views
