import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "kidney-diffusion_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    _limit_cpu_threads()


def _limit_cpu_threads():
    """The CPU oracle runs on torch's intra-op pool, which defaults to every logical CPU of the HOST; a GPU box
    hands a container a CPU share (cgroup quota, 16 of 256 on the MI355X boxes), and an oversubscribed pool runs
    the oracle 3-4x slower.  Same rule as bench.py's cpu_baseline: min(physical cores, affinity, quota)."""
    import os

    import torch

    n = os.cpu_count() or 1
    if hasattr(os, "sched_getaffinity"):
        n = min(n, len(os.sched_getaffinity(0)))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    torch.set_num_threads(max(1, n))


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
