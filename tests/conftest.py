import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cores():
    """Cores this process may use: affinity mask capped by the cgroup CPU quota (a GPU box hands out one GPU's share of a
    large host; an OpenMP runtime that sizes its team by the machine oversubscribes that share many times over)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


# the oracle (OpenMP) is loaded lazily by the fixtures below: size its thread team before that happens
os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cores()))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through the C-ABI HIP library)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/). Built on demand with gcc."""
    from oracle import oracle as o
    o.build()
    return o
