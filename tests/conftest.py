"""pytest configuration: the `gpu` marker and import paths.

`-m "not gpu"`: oracle vs the reference's golden vectors, host logic, C-ABI export check.
`-m gpu`:       parity tests proper -- they call the HIP path through the C-ABI.
"""
import os
import sys

import pytest

# the scan-service tests run four dispatcher lanes in this process: one hardware queue per lane, set by the process's launcher
# (here: the test session) before the HIP runtime initialises -- the library never touches the environment itself
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: at-size comparisons that take minutes of host time (the sequential CPU reference builds); skipped unless "
                                       "LANTERN_TEST_SLOW=1 -- `bash scripts/gpu.sh tests-slow`; each has a fast representative in the default run")


# The comparisons whose cost is a sequential CPU build of 10^5 .. 4 x 10^5 rows (110 - 150 s each on the GPU box's host) are kept out of the
# default `-m gpu` run, which the driver gives 1200 s: mark them `slow_gpu`.  Their fast representatives stay in the default run.
slow_gpu = [pytest.mark.slow, pytest.mark.skipif(os.environ.get("LANTERN_TEST_SLOW", "0") in ("", "0"),
                                                  reason="slow at-size comparison: LANTERN_TEST_SLOW=1 (bash scripts/gpu.sh tests-slow)")]


def _experimental_build() -> bool:
    try:
        from lantern_amd import build, capi

        if not os.path.exists(build.LIB):
            return False
        return capi.experimental_build()
    except Exception:
        return False


# The walk variants that lost their A/B (csrc/experimental/: LANTERN_GPU_SPEC=3, =4) are not in the default library; their parity
# tests run when the library was built with LANTERN_BUILD_EXPERIMENTAL=1 (python -m lantern_amd.build) and are skipped otherwise.
needs_experimental = pytest.mark.skipif(not _experimental_build(), reason="csrc/experimental/ is not in this library (LANTERN_BUILD_EXPERIMENTAL=1 builds it)")


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
