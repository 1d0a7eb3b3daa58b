"""The C-ABI library loads and exports every symbol include/mpmpc.h declares (no compute calls:
this runs in the GPU-less container), and the host binding mirrors the header's structs."""
import ctypes as C
import os
import re

import pytest

import mpmpc
import mpmpc_testlib as T

ROOT = T.ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build_library()
    return mpmpc.load_library()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "mpmpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpmpc_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(mpmpc.EXPORTS) == names


def test_struct_layouts_match_header(lib):
    # default settings round trip through the C side proves field order / sizes agree
    s = mpmpc.Settings()
    lib.mpmpc_default_settings(C.byref(s))
    d = mpmpc.default_settings()
    for name, _ in mpmpc.Settings._fields_:
        assert getattr(s, name) == getattr(d, name), name
    assert C.sizeof(mpmpc.Config) == 4 * 4 + 8 * (3 + 2 + 3 + 3 + 3 + 2 + 2 + 2 + 3)
    assert lib.mpmpc_stage_ld(30) == 32 and lib.mpmpc_stage_ld(10) == 16 and lib.mpmpc_stage_ld(50) == 64
    assert b"gfx950" in lib.mpmpc_version()


def test_loaded_library_was_built_from_this_tree(lib):
    """The binary that ships to the GPU box must be the tree's: its version string carries the hash of the sources
    it was compiled from (a stale .so had been passing for the committed code)."""
    import __graft_entry__ as g
    assert g.source_hash().encode() in lib.mpmpc_version(), lib.mpmpc_version()


def test_argument_validation_without_device(lib):
    """Errors that are detected before any device call are reported through the ABI's error channel."""
    h = C.c_void_p()
    cfg = T.stock_config(30)
    cfg.N = 2
    assert lib.mpmpc_create(C.byref(cfg), None, C.byref(h)) == -1
    assert b"horizon" in lib.mpmpc_last_error()
    cfg = T.stock_config(30)
    bad = mpmpc.default_settings(alpha=2.5)
    assert lib.mpmpc_create(C.byref(cfg), C.byref(bad), C.byref(h)) == -1
    assert b"alpha" in lib.mpmpc_last_error()
    if mpmpc.device_count() == 0:
        # no HIP device here: creation must fail loudly, never fall back to a CPU path
        assert lib.mpmpc_create(C.byref(T.stock_config(30)), None, C.byref(h)) == -2
        assert lib.mpmpc_last_error() != b""
    with pytest.raises(TypeError):
        mpmpc.default_settings(no_such_setting=1)
    with pytest.raises(ValueError):
        mpmpc.make_config(30, [1, 0], [0.5, 0], [1, 0, 0], [0] * 3, [0] * 3, [0, 0], [1, 1], 4.0, 0.12)
