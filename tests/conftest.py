import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _usable_cores():
    """CPU affinity capped by the container's CFS quota (the GPU boxes show 256 logical CPUs behind a 16-CPU quota: 256 OpenMP threads there
    run the oracle several times SLOWER than 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cores()))      # before any OpenMP runtime (the oracle's, torch's) starts
os.environ["GSDF_TEST_THREADS"] = os.environ["OMP_NUM_THREADS"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(autouse=True)
def _flush_c_stdio():
    """The C++ code under test (the drop-in layer, and above all the reference's own sources in oracle/_ref) prints through C stdio, which is
    fully buffered on a pipe: without a flush inside the test its output would spill AFTER pytest's summary line.  Flushed here it is
    captured with the test it belongs to."""
    yield
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    orc.set_threads(int(os.environ.get("OMP_NUM_THREADS", "0")) or _usable_cores())
    return orc


def pytest_sessionfinish(session, exitstatus):
    """The compositing parity measurements of the session (tests/util.py: assert_clean_parity) -> gpurun_out/parity_small_cases.json"""
    try:
        import json
        import util
        if util.PARITY_LOG:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            rows = [dict(tensor=nm, hip_vs_fp64_oracle=st) for nm, st in util.PARITY_LOG]
            json.dump(rows, open(os.path.join(out, "parity_small_cases.json"), "w"), indent=1)
        if util.MATCHED_LOG:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            rows = [dict(case=c, tensor=nm, hip_vs_fp64_oracle_under_matched_decisions=st) for c, nm, st in util.MATCHED_LOG]
            json.dump(rows, open(os.path.join(out, "parity_small_cases_r06.json"), "w"), indent=1)
    except Exception:
        pass
