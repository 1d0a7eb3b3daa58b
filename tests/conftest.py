import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "multi-purpose-mpc_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # the library every test loads must be the tree's: (re)build it before anything can dlopen a stale one
    # (a no-op when the hash of the sources matches the one the binary was built from)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build_library()


@pytest.fixture(scope="session")
def track():
    import scenarios
    return scenarios.sim_track()


@pytest.fixture(scope="session")
def otrack():
    import mpc_np
    return mpc_np.Track.sim_track()


@pytest.fixture(scope="session")
def emu():
    import mpmpc_testlib
    return mpmpc_testlib.Emul()
