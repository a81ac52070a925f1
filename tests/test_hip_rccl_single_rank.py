"""The exchanges of mobocmf_amd.parallel on the nccl (= RCCL) backend with one rank, on the GPU (reference exchange
points JESMOC_MFDGP.py:125-135, blackbox_mfdgp_fitter.py:317-341): a fresh process creates the process group before any other GPU
call and runs every collective the N-rank job issues -- results must equal the inputs and librccl must be mapped."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_exchange_runs_on_rccl_with_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_single_rank.py")], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec["backend"] == "nccl" and rec["world"] == 1
    assert rec["librccl_mapped"], rec
    assert all(rec["ok"].values()), rec["ok"]
    print("\nRCCL, one rank, median of 5 (ms):", {k: round(v, 3) for k, v in rec.items() if k.endswith("_ms")})
