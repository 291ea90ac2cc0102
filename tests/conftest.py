import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure: make sure liboracle.so exists before any test."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)


# What the full-size checks (tests/test_gpu_fullsize.py) verified, at which size: one line each in the terminal summary, so
# that it lands in the captured tail of `pytest -m gpu` (a row of dots says nothing about the size a check ran at).
FULLSIZE_RECORDS = []


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not FULLSIZE_RECORDS:
        return
    terminalreporter.write_sep("=", "full-size verification (tests/test_gpu_fullsize.py, tests/test_min_distance_property.py)")
    for r in FULLSIZE_RECORDS:
        terminalreporter.write_line("FULLSIZE %s cloud, N = %d: %s -- %d levels, %d points checked" % (
            r["cloud"], r["points"], r["check"], r["levels"], r["points_checked"]))
