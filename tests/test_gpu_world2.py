"""Two ranks of bench.py on ONE GPU (--one-device): the multi-rank code path of the bench (shares,
merge pipeline, barriers, max-over-ranks timing, self-check, the strong-scaling run beside the weak
one) with real processes and real renders.  RCCL refuses two ranks on one device, so the merge takes
the torch.distributed fallback over gloo; the figures of such a run are not measurements."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.timeout(280)
@pytest.mark.parametrize("extra, port", [([], 29541), (["--scaling", "strong", "--split", "tile"], 29542)])
def test_two_ranks_on_one_device(extra, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--steps", "2", "--warmup", "1",
           "--spp", "8"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]            # rank 0 alone prints, and one line only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["merge_self_check"].startswith("bit-identical")
    assert "rehearsal" in line and line["value"] > 0 and line["vs_baseline"] is None
    paths = 1920 * 1080 * 8
    if extra:
        assert line["scaling"] == "strong" and "strong_scaling" not in line
        assert abs(line["value"] * line["ms_per_step"] * 1e3 - paths) < 1e-6 * paths          # the fixed budget, once
    else:
        assert line["scaling"] == "weak"
        assert abs(line["value"] * line["ms_per_step"] * 1e3 - 2 * paths) < 1e-6 * paths      # every rank renders the whole budget
        s = line["strong_scaling"]
        assert s["split"] == "spp" and abs(s["value"] * s["ms_per_step"] * 1e3 - paths) < 1e-6 * paths
    assert round(line["rays_per_path"], 1) > 1.0            # summed over both ranks' last steps
