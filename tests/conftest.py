"""pytest configuration: the `gpu` marker and import paths.

`-m "not gpu"`: oracle vs the reference's golden vectors, host logic, C-ABI export check.
`-m gpu`:       parity tests proper -- they call the HIP path through the C-ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "lantern_expected.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding

    binding.build()
    binding.lib()
    return binding


def usable_cores() -> int:
    """Threads this process may really use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


@pytest.fixture(scope="session")
def cores():
    return usable_cores()
