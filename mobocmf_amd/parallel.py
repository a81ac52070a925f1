"""Multi-GPU layout of the path: one process per GPU, independent surrogates sharded over the
ranks with no data-path collective, and ONE all-gather (RCCL over xGMI; gloo on CPU for tests) per exchange
point: the per-output posterior moments for the joint acquisition (JESMOC_MFDGP.py:125-135).

Payloads are tiny (<= 0.6 MB), i.e. latency-bound: a single all-gather on the default communicator is the
whole collective; there is nothing to bucket or overlap.
"""
import torch


def world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _no_group():
    """True when no process group exists: the exchanges below are then identities.  With a group -- also one of a single
    rank -- every exchange runs its collective: a one-rank run (bench.py through the launcher, tests/test_hip_rccl_single_rank.py)
    executes exactly the RCCL calls an N-rank run does."""
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized())


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def launch_ranks(argv, n, port=None, extra_env=None, timeout=None):
    """One fresh process per rank (= per GPU) of ``argv`` (e.g. [sys.executable, "bench.py", ...]), with the
    torch.distributed.run environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT).  The caller
    must not have touched the GPU (children are started with fork+exec).  Rank 0 inherits stdout, the other ranks'
    stdout goes to stderr, so a program whose rank 0 prints one record still prints exactly one.  Returns the list of
    exit codes.

    No rank outlives this call: every rank runs in its own session (= process group); if a rank fails, the timeout
    expires, the parent receives SIGTERM / SIGINT / SIGHUP, or anything raises inside the poll loop, every live rank's
    GROUP gets SIGTERM, then SIGKILL after a grace period (a rank blocked in an RCCL collective would otherwise hold its
    GPU for ever), and all children are reaped.  A signal is re-raised as KeyboardInterrupt / SystemExit(128 + signo)
    after the clean-up.  ``port=None`` picks a free port; since another process may grab it before rank 0 binds it, a
    launch whose ranks all die within the first seconds with rank 0 reporting the address in use is retried once on a
    fresh port."""
    import os
    import signal
    import subprocess
    import sys
    import threading
    import time

    def start(port_):
        procs_ = []
        try:
            for r in range(n):
                env = dict(os.environ)
                env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                            "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port_), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
                env.update(extra_env or {})
                procs_.append(subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else sys.stderr,
                                               start_new_session=True))
        except BaseException:
            stop(procs_)
            raise
        return procs_

    def stop(procs_, grace=20.0):
        """SIGTERM every live rank's process group, SIGKILL what is left after ``grace`` seconds, reap everything."""
        live = [pr for pr in procs_ if pr.poll() is None]
        for pr in live:
            try:
                os.killpg(pr.pid, signal.SIGTERM)
            except (ProcessLookupError, PermissionError):
                pass
        t_end = time.time() + grace
        for pr in live:
            try:
                pr.wait(timeout=max(0.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
                pr.wait()
        for pr in procs_:
            if pr.poll() is None:
                pr.wait()

    class _Signalled(BaseException):
        pass

    got = []

    def on_signal(signo, _frame):
        got.append(signo)
        raise _Signalled()

    handlers = {}
    if threading.current_thread() is threading.main_thread():      # signal handlers can only be set there
        for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            handlers[sg] = signal.signal(sg, on_signal)
    attempts = 2 if port is None else 1
    procs = []
    try:
        for attempt in range(attempts):
            use_port = _free_port() if port is None else port
            procs = start(use_port)
            t0 = time.time()
            codes = [None] * n
            while any(c is None for c in codes):
                for r, pr in enumerate(procs):
                    if codes[r] is None:
                        codes[r] = pr.poll()
                failed = any(c not in (None, 0) for c in codes)
                if failed or (timeout is not None and time.time() - t0 > timeout):
                    stop(procs)
                    codes = [pr.returncode for pr in procs]
                    break
                time.sleep(0.05)
            # the port-probe race: rank 0 could not bind (torch reports EADDRINUSE and exits within seconds)
            lost_race = (attempt + 1 < attempts and n > 1 and codes[0] not in (0, None) and time.time() - t0 < 15.0
                         and not _port_is_free(use_port))
            if not lost_race:
                return codes
        return codes
    except _Signalled:
        stop(procs)
        if got and got[0] == signal.SIGINT:
            raise KeyboardInterrupt()
        raise SystemExit(128 + (got[0] if got else signal.SIGTERM))
    finally:
        stop(procs, grace=5.0)          # no-op when everything has been reaped already
        for sg, h in handlers.items():
            signal.signal(sg, h)


def _port_is_free(port):
    import socket
    sk = socket.socket()
    try:
        sk.bind(("127.0.0.1", port))
        return True
    except OSError:
        return False
    finally:
        sk.close()


def shard_blackboxes(names, rank=None, world_size=None):
    """Round-robin assignment of black-box (objective / constraint) names to ranks; every rank derives the
    same table.  Returns (my_names, owner_of) with owner_of[name] = rank.  Counts per rank may differ (5 names on 2
    ranks) and a rank may end up without any constraint: the exchanges below are written for ragged and empty shards."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    names = list(names)
    owner = {n: i % world_size for i, n in enumerate(names)}
    return [n for n in names if owner[n] == rank], owner


def _collective_device(t):
    """gloo rehearsals of a GPU path hop through the host; RCCL takes HBM buffers."""
    import torch.distributed as dist
    return torch.device("cpu") if (t.device.type == "cuda" and dist.get_backend() == "gloo") else t.device


def all_gather_moments(local):
    """local: (k, ...) tensor of this rank's k surrogates (the SAME k on every rank) -> (world*k, ...) in rank order.
    Without a process group: the identity (no communicator needed)."""
    import torch.distributed as dist
    _, w = world()
    if _no_group():
        return local
    dev = local.device
    local = local.contiguous().to(_collective_device(local))
    out = torch.empty((w * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)      # concatenation along dim 0, rank order
    return out.to(dev)


def all_gather_ragged(local):
    """local: (k_r, ...) with k_r free per rank (0 allowed; the trailing shape must agree) -> list of the W per-rank
    tensors.  EVERY rank enters both collectives (row counts, then rows padded to the largest count) whatever its own
    count is -- a rank that skipped the call because it holds no rows would leave the others waiting."""
    import torch.distributed as dist
    _, w = world()
    if _no_group():
        return [local]
    dev = local.device
    cdev = _collective_device(local)
    counts = torch.empty(w, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=cdev))
    counts = counts.tolist()
    kmax = max(counts)
    if kmax == 0:
        return [local.new_zeros((0,) + tuple(local.shape[1:])) for _ in range(w)]
    pad = torch.zeros((kmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=cdev)
    pad[:local.shape[0]] = local.detach().to(cdev)
    out = torch.empty((w * kmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=cdev)
    dist.all_gather_into_tensor(out, pad)
    return [out[r * kmax:r * kmax + counts[r]].to(dev) for r in range(w)]


def broadcast_(t, src=0):
    """In-place broadcast from ``src`` (points every rank must agree on: x~ of the conditioned training, acquisition
    candidates).  Without a process group: no-op."""
    import torch.distributed as dist
    if _no_group():
        return t
    cdev = _collective_device(t)
    if cdev != t.device:
        host = t.detach().to(cdev)
        dist.broadcast(host, src)
        t.copy_(host)
    else:
        dist.broadcast(t, src)
    return t


def coupled_acquisition(local_acq):
    """Sum over ALL black-boxes of the per-black-box acquisition (JESMOC_MFDGP.py:125-135): each rank holds
    the (k_local, T) values of its own surrogates (any k_local, 0 included); one ragged all-gather, then a local sum."""
    return torch.cat(all_gather_ragged(local_acq), 0).sum(0)


_gather_plans = {}


def reset_gather_plans():
    """Forget the negotiated row layouts (call on every rank when the sharding of the black-boxes changes inside one process
    group; a new process group gets a new plan by itself)."""
    _gather_plans.clear()


def _group_key():
    """Identity of the default process group: a plan never outlives the group it was negotiated in (tests destroy and
    re-create groups inside one process; ``group_count`` grows with every init_process_group)."""
    import torch.distributed as dist
    c10d = dist.distributed_c10d
    return (id(c10d._get_default_group()), getattr(c10d._world, "group_count", 0), dist.get_backend())


def _layout_hash(k_obj, k_con, oi, ci):
    """48 bits: its two 24-bit halves are exactly representable in float32 as well as float64 (the header row of
    ``gather_with_local_grad`` travels in the moments' dtype)."""
    h = 1469598103934665603
    for v in (k_obj, k_con, -1 if oi is None else len(oi)) + (oi or ()) + (-2 if ci is None else len(ci),) + (ci or ()):
        h = ((h ^ (int(v) & 0xFFFFFFFF)) * 1099511628211) & 0x7FFFFFFFFFFFFFF
    return (h ^ (h >> 48)) & 0xFFFFFFFFFFFF


def _gather_plan(k_obj, k_con, obj_index, con_index, device):
    """Row bookkeeping of ``gather_with_local_grad``, negotiated ONCE per process group and cached: per-rank row counts, the
    padded row count of the per-step all-gather, the permutations into global black-box order, and every rank's layout hash.
    The global indices are static for a whole training run, so their exchange and validation (a host sync) happen here, not
    at every step.  Every rank enters the same two collectives whatever it holds -- a rank whose own arguments are malformed
    says so INSIDE the first collective -- so disagreements (one index per row violated somewhere; some ranks pass indices,
    others do not; indices that are no permutation) raise the same error on every rank instead of dead-locking the others.
    The cache is keyed on the process group, not on the local layout: a layout that changes later inside the same group is
    caught by the per-step check of ``gather_with_local_grad`` (on every rank, after the collective)."""
    import torch.distributed as dist
    r, w = world()
    oi = None if obj_index is None else tuple(int(i) for i in obj_index)
    ci = None if con_index is None else tuple(int(i) for i in con_index)
    key = _group_key()
    plan = _gather_plans.get(key)
    if plan is not None:
        return plan
    bad_local = int((oi is not None and len(oi) != k_obj) or (ci is not None and len(ci) != k_con))
    cdev = torch.device("cpu") if (device.type == "cuda" and dist.get_backend() == "gloo") else device
    meta = torch.empty(w, 6, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(meta, torch.tensor([[k_obj, k_con, int(oi is not None), int(ci is not None), bad_local,
                                                     _layout_hash(k_obj, k_con, oi, ci)]], dtype=torch.int64, device=cdev))
    meta = meta.cpu()
    if int(meta[:, 4].sum()) > 0:
        raise ValueError("gather_with_local_grad: one global index per local row (violated on rank(s) %s)" %
                         [q for q in range(w) if int(meta[q, 4])])
    ko, kc = meta[:, 0].tolist(), meta[:, 1].tolist()
    has_o = {int(meta[q, 2]) for q in range(w) if ko[q] > 0}
    has_c = {int(meta[q, 3]) for q in range(w) if kc[q] > 0}
    if len(has_o) > 1 or len(has_c) > 1:
        raise ValueError("gather_with_local_grad: either every rank passes the global indices of its rows or none does "
                         "(objectives %s, constraints %s)" % (meta[:, 2].tolist(), meta[:, 3].tolist()))
    use_o, use_c = has_o == {1}, has_c == {1}
    kmax = max(a + b for a, b in zip(ko, kc))
    order_o = order_c = None
    if (use_o or use_c) and kmax > 0:
        idx = torch.full((1, kmax), -1, dtype=torch.int64, device=cdev)
        loc = (list(oi) if (use_o and oi is not None) else [-1] * k_obj) + (list(ci) if (use_c and ci is not None) else [-1] * k_con)
        if loc:
            idx[0, :len(loc)] = torch.tensor(loc, dtype=torch.int64, device=cdev)
        allidx = torch.empty(w, kmax, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(allidx, idx)
        allidx = allidx.cpu()
        go = [int(v) for q in range(w) for v in allidx[q, :ko[q]]]
        gc = [int(v) for q in range(w) for v in allidx[q, ko[q]:ko[q] + kc[q]]]
        for name, used, g in (("objective", use_o, go), ("constraint", use_c, gc)):
            if used and sorted(g) != list(range(len(g))):
                raise ValueError("gather_with_local_grad: global %s indices of all ranks must be a permutation of 0..n-1, "
                                 "got %s" % (name, g))
        if use_o:
            order_o = torch.argsort(torch.tensor(go, dtype=torch.int64)).to(device)
        if use_c:
            order_c = torch.argsort(torch.tensor(gc, dtype=torch.int64)).to(device)
    plan = {"ko": ko, "kc": kc, "kmax": kmax, "order_o": order_o, "order_c": order_c, "cdev": cdev,
            "hashes": [int(h) for h in meta[:, 5].tolist()]}
    _gather_plans[key] = plan
    return plan


def gather_with_local_grad(fm, fv, cm, cv, obj_index=None, con_index=None):
    """omega-factor coupling of the conditioned training (blackbox_mfdgp_fitter.py:317-341) when the surrogates are
    sharded over ranks: every rank needs ALL models' (mean, var) at the 10 x~ points; its own rows keep their autograd
    history, the other ranks' rows arrive as constants.  ONE padded all-gather per call carries objectives and constraints
    together (the row layout is negotiated once per process group and cached, ``_gather_plan``) plus one header row per rank
    with that rank's CURRENT layout hash: a layout that differs from the negotiated one -- on this rank or on a peer -- raises
    on every rank after the collective, instead of slicing stale rows or hanging.  Ranks may hold different numbers of
    objectives / constraints, none included.  ``obj_index`` / ``con_index``: the GLOBAL position of each local row (the
    column of the Pareto front / the entry of the threshold vector it belongs to); the result is ordered by it.  Without
    them the rows come back in rank order.  Without a process group: identity."""
    import torch.distributed as dist
    r, w = world()
    if _no_group():
        return fm, fv, cm, cv
    k_obj, k_con = fm.shape[0], cm.shape[0]
    plan = _gather_plan(k_obj, k_con, obj_index, con_index, fm.device)
    T = fm.shape[1] if k_obj else cm.shape[1]
    dev, cdev, kmax = fm.device, plan["cdev"], plan["kmax"]
    oi = None if obj_index is None else tuple(int(i) for i in obj_index)
    ci = None if con_index is None else tuple(int(i) for i in con_index)
    local = torch.cat([torch.stack([fm, fv], 1), torch.stack([cm, cv], 1)], 0)        # (k_obj + k_con, 2, T), with gradient
    # row 0 = header (the layout hash, split into two 24-bit halves: exact in float32 and float64), rows 1.. = the moments,
    # zero padded
    h = _layout_hash(k_obj, k_con, oi, ci)
    head = torch.zeros(1, 2, T, dtype=local.dtype)
    head[0, 0, 0], head[0, 1, 0] = float(h >> 24), float(h & ((1 << 24) - 1))
    pad = torch.zeros(kmax + 1, 2, T, dtype=local.dtype, device=cdev)
    pad[:1] = head.to(cdev, non_blocking=True)
    nfit = min(local.shape[0], kmax)
    if nfit:
        pad[1:1 + nfit] = local.detach()[:nfit].to(cdev)
    out = torch.empty(w * (kmax + 1), 2, T, dtype=local.dtype, device=cdev)
    dist.all_gather_into_tensor(out, pad)
    out = out.to(dev).reshape(w, kmax + 1, 2, T)
    hdr = out[:, 0, :, 0].to("cpu", torch.float64)      # ONE device-to-host copy for all ranks' headers
    got = [(int(hdr[q, 0]) << 24) | int(hdr[q, 1]) for q in range(w)]
    if got != plan["hashes"]:
        raise ValueError("gather_with_local_grad: the row layout of rank(s) %s differs from the one negotiated for this "
                         "process group; call parallel.reset_gather_plans() on every rank before changing the sharding" %
                         [q for q in range(w) if got[q] != plan["hashes"][q]])
    if kmax == 0:
        return fm, fv, cm, cv
    objs, cons = [], []
    for q in range(w):
        rows = local if q == r else out[q, 1:]                                         # own rows: with gradient
        objs.append(rows[:plan["ko"][q]])
        cons.append(rows[plan["ko"][q]:plan["ko"][q] + plan["kc"][q]])
    allo, allc = torch.cat(objs, 0), torch.cat(cons, 0)
    if plan["order_o"] is not None:
        allo = allo.index_select(0, plan["order_o"])
    if plan["order_c"] is not None:
        allc = allc.index_select(0, plan["order_c"])
    return allo[:, 0], allo[:, 1], allc[:, 0], allc[:, 1]


# ---------------------------------------------------------------------------------------------------------------
# Level 2 of SURVEY 8(e): ONE surrogate over several GPUs.  The N rows of the batch are sharded over the ranks
# (each rank therefore owns N/W * S of the N' columns of every layer's K_mn panel); parameters, K_mm and its
# Cholesky chain (M^3/3, tiny) are replicated; the gradients of the M^2-sized objects and the hyper-parameters
# need ONE all-reduce(sum) per step, and the two scalars of the ELBO ride in the same buffer.
# ---------------------------------------------------------------------------------------------------------------
def shard_rows(n, rank=None, world_size=None, device=None):
    """Row indices of this rank: rank, rank + W, rank + 2W, ... (strided, so ragged N and the fidelity mix stay balanced)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return torch.arange(rank, n, world_size, device=device)


def all_reduce_sum_(buf):
    """In-place sum over the ranks (RCCL on HBM buffers; the gloo rehearsal hops through the host)."""
    import torch.distributed as dist
    if _no_group():
        return buf
    if buf.device.type == "cuda" and dist.get_backend() == "gloo":
        host = buf.cpu()
        dist.all_reduce(host)
        buf.copy_(host)
    else:
        dist.all_reduce(buf)
    return buf


class GradBucket:
    """One flat float64 buffer holding every trainable parameter's gradient (+ ``extra`` trailing scalars); the
    parameters' ``.grad`` are views into it, so backward accumulates straight into the buffer that is all-reduced."""

    def __init__(self, params, extra=0):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n + extra, dtype=torch.float64, device=dev)
        off = 0
        for p in self.params:
            assert p.dtype == torch.float64
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.extra = self.flat[n:]

    def zero_(self):
        self.flat.zero_()

    def all_reduce(self):
        all_reduce_sum_(self.flat)


def _graphed_base():
    from .util.graphed_step import GraphedELBOStep
    return GraphedELBOStep


class RowShardedELBOStep(_graphed_base()):
    """The ELBO step of ONE surrogate with its batch rows sharded over the ranks of the default process group.

    Every rank builds the same model from the same full data set (identical initialisation), keeps rows
    ``shard_rows(N)`` of (x, y, fidelities) and runs the unmodified hot path on them.  ``VariationalELBOMF`` scales
    the KL by (local batch / num_data) (variational_elbo_mf.py:44-47), so the local ELBOs sum to the full-batch ELBO
    exactly and the summed gradients are the full-batch gradients: all ranks then apply the same Adam update and stay
    bit-identical replicas.  Per step: graph(forward + backward) | all-reduce of one bucket | graph(Adam).
    ``fixed_eps`` (tests) is given for the FULL batch (N*S per layer) and sliced here."""

    exchanges = True

    def __init__(self, model, elbo, x, y, fidelities, lr, fixed_eps=None, **kw):
        idx = shard_rows(x.shape[0], device=x.device)
        S = model.num_samples_for_training
        if fixed_eps is not None:
            fixed_eps = [None if e is None else e.reshape(x.shape[0], S)[idx].reshape(-1).contiguous() for e in fixed_eps]
        self.rows = idx
        self.bucket = GradBucket(model.parameters(), extra=2)
        super().__init__(model, elbo, x[idx].contiguous(), y[idx].contiguous(), fidelities[idx].contiguous(), lr,
                         fixed_eps=fixed_eps, **kw)
        if self.row_order is not None:      # the base class re-ordered the shard by fidelity: self.x == x[self.rows] again
            self.rows = idx[self.row_order]

    def _fwd_bwd(self):
        self.bucket.zero_()
        rows = self.layer_rows if self.layer_rows is not None else [self.x.shape[0]] * self.L
        eps = self.fixed_eps if self.fixed_eps is not None else \
            [None] + [torch.randn(rows[l] * self.S, dtype=torch.float64, device=self.x.device) for l in range(1, self.L)]
        out = self.model(self.x, eps=eps, rows=self.layer_rows)
        res = self.elbo(out, self.y.T, self.fid)
        (-res[0]).backward()
        self.bucket.extra[0].copy_(-res[0].detach())
        self.bucket.extra[1].copy_(res[1].detach())
        self.model.clear_kl_cache()

    def _exchange(self):
        self.bucket.all_reduce()

    def _update(self):
        self.optimizer.step()
        self.loss.copy_(self.bucket.extra[0])
        self.kl.copy_(self.bucket.extra[1])
