import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "multi-purpose-mpc_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


LIBRARY_BUILD_ERROR = None


@pytest.fixture(scope="session")
def built_library():
    """Path of the tree's libmpmpc.so; fails (never skips) when the session could not build it."""
    if LIBRARY_BUILD_ERROR:
        pytest.fail(LIBRARY_BUILD_ERROR)
    import __graft_entry__ as g
    return os.path.join(g.CSRC, "libmpmpc.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # the library every test loads must be the tree's: (re)build it before anything can dlopen a stale one
    # (a no-op when the hash of the sources matches the one the binary was built from)
    # A box without hipcc (or a failing compile) must not take the oracle / emulation / host tests down with it: the
    # failure is recorded and only what needs libmpmpc.so - test_abi.py, the gpu tests - fails, loudly, on it.
    import subprocess
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    global LIBRARY_BUILD_ERROR
    try:
        g.build_library()
    except (FileNotFoundError, subprocess.CalledProcessError) as e:
        LIBRARY_BUILD_ERROR = "libmpmpc.so could not be built from the tree's sources: %r" % (e,)


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
