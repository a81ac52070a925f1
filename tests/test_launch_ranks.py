"""The rank launcher behind `python bench.py --gpus N` (mobocmf_amd.parallel.launch_ranks): one fresh process per rank with
the torch.distributed.run environment; no rank may outlive the call -- whether a sibling failed, the timeout expired or the
launcher itself was signalled.  CPU-only children here; the gloo rehearsal of bench.py through the launcher is the `gpu` test
at the bottom."""
import json
import os
import signal
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    # a zombie still answers kill(0): look at its state
    try:
        with open("/proc/%d/stat" % pid) as fh:
            return fh.read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def test_three_ranks_get_the_rendezvous_environment(tmp_path):
    from mobocmf_amd import parallel
    code = ("import os, json; r = os.environ['RANK']; "
            "json.dump({k: os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', "
            "'HSA_ENABLE_IPC_MODE_LEGACY')}, open(os.path.join(%r, 'rank' + r + '.json'), 'w'))" % str(tmp_path))
    codes = parallel.launch_ranks([sys.executable, "-c", code], 3, timeout=60)
    assert codes == [0, 0, 0]
    envs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1


def test_a_failing_rank_terminates_its_siblings(tmp_path):
    """Rank 1 exits with code 7; ranks 0 and 2 sit in a `collective` (a long sleep) and must be terminated, not waited for."""
    from mobocmf_amd import parallel
    code = ("import os, sys, time; r = int(os.environ['RANK']); "
            "open(os.path.join(%r, 'pid%%d' %% r), 'w').write(str(os.getpid())); "
            "time.sleep(0.5 if r == 1 else 120); sys.exit(7 if r == 1 else 0)" % str(tmp_path))
    t0 = time.time()
    codes = parallel.launch_ranks([sys.executable, "-c", code], 3, timeout=100)
    assert time.time() - t0 < 30
    assert codes[1] == 7 and codes[0] == -signal.SIGTERM and codes[2] == -signal.SIGTERM
    for r in range(3):
        assert not _alive(int(open(tmp_path / ("pid%d" % r)).read()))


def test_the_timeout_terminates_every_rank(tmp_path):
    from mobocmf_amd import parallel
    t0 = time.time()
    codes = parallel.launch_ranks([sys.executable, "-c", "import time; time.sleep(120)"], 2, timeout=1.0)
    assert time.time() - t0 < 30 and codes == [-signal.SIGTERM, -signal.SIGTERM]


@pytest.mark.parametrize("signo", [signal.SIGTERM, signal.SIGINT])
def test_a_signalled_launcher_takes_its_ranks_down(tmp_path, signo):
    """a driver timeout SIGTERMs the launcher; ranks blocked in a collective must not keep their GPUs.  Children
    that IGNORE SIGTERM are killed after the grace period."""
    child = ("import os, signal, time; r = int(os.environ['RANK']); "
             "open(os.path.join(%r, 'pid%%d' %% r), 'w').write(str(os.getpid())); time.sleep(300)" % str(tmp_path))
    parent = ("import sys; sys.path.insert(0, %r); from mobocmf_amd import parallel; "
              "parallel.launch_ranks([sys.executable, '-c', %r], 2)" % (ROOT, child))
    pr = subprocess.Popen([sys.executable, "-c", parent])
    t0 = time.time()
    while not all(os.path.exists(tmp_path / ("pid%d" % r)) for r in range(2)):
        assert time.time() - t0 < 60 and pr.poll() is None
        time.sleep(0.05)
    time.sleep(0.2)
    pids = [int(open(tmp_path / ("pid%d" % r)).read()) for r in range(2)]
    assert all(_alive(p) for p in pids)
    pr.send_signal(signo)
    rc = pr.wait(timeout=60)
    assert rc != 0
    t1 = time.time()
    while any(_alive(p) for p in pids):
        assert time.time() - t1 < 10, "ranks outlived the launcher"
        time.sleep(0.05)


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env, capture_output=True, timeout=300)
    assert pr.returncode == 2 and b"WORLD_SIZE" in pr.stderr


@pytest.mark.gpu
def test_bench_two_ranks_through_the_launcher_over_gloo():
    """`python bench.py --gpus 2` end to end on one card: the launcher starts two fresh ranks, both train their own C1
    surrogates on the HIP path, the exchange is a world-size-2 all-gather (gloo rehearsal of the RCCL call), rank 0 prints
    the ONE record."""
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--force-device", "0",
                         "--config", "C1", "--steps", "3", "--warmup", "1", "--repeats", "1", "--no-roofline", "--no-cpu-baseline"],
                        capture_output=True, timeout=600)
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    lines = [ln for ln in pr.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["backend"] == "gloo" and rec["finite"]
    assert rec["config"]["surrogates_per_gpu"] == 3 and rec["value"] > 0 and rec["scaling"] == "weak"
