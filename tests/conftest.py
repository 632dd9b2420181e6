import faulthandler
import os
import sys

import pytest

# A crash inside the HIP runtime or the library (SIGSEGV / SIGABRT / SIGBUS) must leave the Python stack of every thread in the log:
# round 4 lost a host segfault in a full GPU run because nothing had been armed to say where it happened.
faulthandler.enable(all_threads=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tiny-object-detection_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Builds (if stale) the HIP library and the oracle; returns the yolact_amd package."""
    import __graft_entry__ as g
    return g.build()


@pytest.fixture(scope="session")
def oracle(built):
    import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
