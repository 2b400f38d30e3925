"""The host half of the library under AddressSanitizer and UBSan (tools/host_sanitize.sh: every
scene factory, every build mode, the layouts for the device-side build, the records of a refit):
no report, no violation of the tree invariants.  About a minute: most of it compiling."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_under_sanitizers():
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "host_sanitize.sh")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if " mode " in l]
    assert len(lines) == 32 and all("violations 0" in l for l in lines), p.stdout[-2000:]
