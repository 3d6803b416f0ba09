import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


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
            json.dump(rows, open(os.path.join(out, "parity_small_cases_r05.json"), "w"), indent=1)
    except Exception:
        pass
