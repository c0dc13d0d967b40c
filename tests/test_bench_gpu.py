"""bench.py as the driver calls it for N > 1: `python bench.py --gpus N` WITHOUT a launcher must start its own ranks
(one process per GPU through torch.distributed.run, as the reference spawns its workers, sample_ultra_res.py:235-240), run the
headline workload on every rank and print ONE JSON line on rank 0 that also carries the number of ranks the collective
library connected and the patch throughput of the ultra-res grid.  A one-GPU box cannot hold two RCCL ranks, so the
rehearsal runs the ranks over gloo on the one card (`KD_BENCH_BACKEND=gloo`); the RCCL run is the driver's."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_self_launch_two_ranks_over_gloo(device):
    env = dict(os.environ, KD_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--line-grid-steps", "1", "--grid-n", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["collective_backend"] == "gloo"
    assert d["metric"].startswith("denoising-steps/sec") and d["scaling"] == "weak" and d["value"] > 0
    assert d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "denoising-steps/s"
    g = d["grid"]
    for key, ncan in (("canvases_1", 1), ("canvases_3", 3)):
        assert g[key]["patches"] == 4 * ncan and g[key]["patches_per_s"] > 0
        assert g[key]["schedule_slots"] <= 4 * ncan      # two ranks share the waves of a 2x2 grid
    assert g["canvases_3"]["schedule_slots"] < 3 * g["canvases_1"]["schedule_slots"] or g["canvases_1"]["schedule_slots"] == 3
