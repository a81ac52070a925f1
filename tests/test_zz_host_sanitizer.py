"""Host-sanitizer target (SURVEY section 5 aux subsystems): the library's host side under AddressSanitizer +
UBSan, driven by the workspace fuzz of tools/fuzz_workspaces.cpp (tools/asan_host.sh).  CPU only: nothing in it touches a GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("bash") is None, reason="needs the ROCm toolchain")
def test_workspace_carving_is_clean_under_asan_and_ubsan():
    pr = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host.sh"), "400"], capture_output=True, timeout=900)
    out = pr.stdout.decode() + pr.stderr.decode()
    assert pr.returncode == 0, out[-3000:]
    assert "no sanitizer report" in out and "ERROR: AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
