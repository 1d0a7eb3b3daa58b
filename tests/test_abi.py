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
    assert C.sizeof(mpmpc.Config) == 4 * 4 + 8 * (3 + 2 + 3 + 3 + 3 + 2 + 2 + 2 + 3 + 3 + 1)
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
    for kw, word in ((dict(early_start=2), b"early_start"), (dict(native_ipm_tol=0.0), b"native_ipm_tol"), (dict(native=3), b"native"),
                     (dict(phase1_accept=-1), b"phase1_accept")):
        bad = mpmpc.default_settings(**kw)
        assert lib.mpmpc_create(C.byref(cfg), C.byref(bad), C.byref(h)) == -1, kw
        assert word in lib.mpmpc_last_error(), (kw, lib.mpmpc_last_error())
    if mpmpc.device_count() == 0:
        # no HIP device here: creation must fail loudly, never fall back to a CPU path
        assert lib.mpmpc_create(C.byref(T.stock_config(30)), None, C.byref(h)) == -2
        assert lib.mpmpc_last_error() != b""
    with pytest.raises(TypeError):
        mpmpc.default_settings(no_such_setting=1)
    with pytest.raises(ValueError):
        mpmpc.make_config(30, [1, 0], [0.5, 0], [1, 0, 0], [0] * 3, [0] * 3, [0, 0], [1, 1], 4.0, 0.12)


# ---- resources of the kernels in the SHIPPED code object (llvm-readelf --notes of libmpmpc.so's gfx950 image)
def _kernel_rows(lib):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import kernel_resources
    return kernel_resources.kernel_table()


def test_reduced_native_kernels_fit_two_waves_per_simd(lib):
    """The batch path of the reference's own weights runs mpmpc_reduced_kernel<G, C>: every instantiation must stay within
    256 registers (vgpr_count is the unified VGPR + AGPR total on gfx950), 20 KB of LDS and NO scratch - the budget of two
    wavefronts per SIMD / eight per CU."""
    rows = [r for r in _kernel_rows(lib) if r["name"].startswith("mpmpc_reduced_kernel")]
    assert len(rows) >= 8, [r["name"] for r in rows]
    # ... and so must its twin for weightings with a terminal cost on the time state (BASELINE config 3)
    tt = [r for r in _kernel_rows(lib) if r["name"].startswith("mpmpc_reduced_t_kernel")]
    assert sorted(r["name"] for r in tt) == ["mpmpc_reduced_t_kernel<64, 16>", "mpmpc_reduced_t_kernel<64, 32>"], tt
    tail = [r for r in _kernel_rows(lib) if r["name"].startswith("mpmpc_reduced_tail_kernel")]
    assert sorted(r["name"] for r in tail) == ["mpmpc_reduced_tail_kernel<32, 16>", "mpmpc_reduced_tail_kernel<64, 16>",
                                               "mpmpc_reduced_tail_kernel<64, 32>"], tail
    for r in tail:
        assert r["vgpr"] <= 256 and r["agpr"] == 0 and r["lds"] <= 20 * 1024, r
        assert r["scratch"] <= (8 if "<64, 16>" in r["name"] else 0), r
    for r in rows + tt:
        assert r["scratch"] == 0, r
        assert r["vgpr"] <= 256 and r["agpr"] == 0, r
        assert r["lds"] <= 20 * 1024, r


def test_pair_layout_kernel_fits_one_wave_per_simd(lib):
    """mpmpc_reduced_pair_kernel (two stages per lane, csrc/lane_pair.hpp): twice the per-lane state of the one-stage kernels -
    one wavefront per SIMD, four per CU: at most 512 registers, 40 KB of LDS (80 slots), and the three dwords of scratch it
    was measured with (a regression fails)."""
    rows = {r["name"]: r for r in _kernel_rows(lib) if r["name"].startswith(("mpmpc_reduced_pair_kernel", "mpmpc_reduced_t_pair_kernel", "mpmpc_reduced_tail_pair_kernel", "mpmpc_reduced_pair_block_kernel",
                                                                            "mpmpc_reduced_t_pair_block_kernel", "mpmpc_reduced_tail_pair_block_kernel"))}
    assert sorted(rows) == sorted(["mpmpc_reduced_pair_kernel<16>", "mpmpc_reduced_pair_kernel<64>", "mpmpc_reduced_t_pair_kernel<64>",
                                   "mpmpc_reduced_tail_pair_kernel<64>", "mpmpc_reduced_pair_block_kernel", "mpmpc_reduced_t_pair_block_kernel",
                                   "mpmpc_reduced_tail_pair_block_kernel"]), rows
    for name, r in rows.items():
        # <16>: four instances per wavefront (horizons 16 .. 31, behind set_packing(16)); <64>: one instance of 65 .. 128 stages per
        # wavefront (the default at horizons 64 .. 127, and its twin for a terminal cost on the time state) - no scratch there
        # (the workgroup kernel of horizons 128 .. 255: its cold slots are dynamic LDS - 80 000 B, two workgroups per CU)
        # (... and the terminal-time solver on that workgroup, the one kernel of the family that spills: 164 B, measured when listed)
        known = {"mpmpc_reduced_pair_kernel<16>": 12, "mpmpc_reduced_t_pair_block_kernel": 164}
        assert r["vgpr"] <= 512 and r["lds"] <= 40 * 1024 and r["scratch"] <= known.get(name, 0), r


def test_no_batch_path_solve_kernel_has_scratch(lib):
    """private_segment_fixed_size must be 0 for every solve kernel a batch launch can pick: the reduced-native kernels, and
    the one-instance-per-wave general kernels that take their tail / the configurations the reduction does not apply to.
    KNOWN_SCRATCH lists the instantiations that still spill - none of them is reachable from BASELINE.json's
    configurations (N >= 32 with bounded e_psi / t or full weight matrices; horizons above 63) - with the bytes measured when they were
    listed: the test fails if one of them grows or a new one appears."""
    KNOWN_SCRATCH = {
        "mpmpc_solve_kernel<64, 32, false, 0>": 80, "mpmpc_solve_kernel<64, 32, true, 0>": 96,
        # (VAR 1 = full weights: since round 5 Q and R may have off-diagonal entries too - dense blocks on every lane)
        "mpmpc_solve_kernel<64, 32, false, 1>": 228, "mpmpc_solve_kernel<64, 32, true, 1>": 236,
        # horizons above 63 (round 5): the general solver on a workgroup of 2 / 4 wavefronts, 512 registers per lane
        # (VAR 2, the reduced polish of the reference's own weights - the tail of the reduced-native workgroup kernel, which has
        #  none - has none at 128 lanes and 28 B at 256, where the eight-row cyclic reduction crosses the wavefronts)
        "mpmpc_solve_block_kernel<256, 2>": 28,
        "mpmpc_solve_block_kernel<128, 0>": 128, "mpmpc_solve_block_kernel<128, 1>": 256,
        "mpmpc_solve_block_kernel<256, 0>": 268, "mpmpc_solve_block_kernel<256, 1>": 396,
        # the one-instance-per-wave form of the reduced-native tail kernel (mpmpc_set_tail_kernel(h, 2); the default form,
        # <32, 16>, has none): ONE dword (a lane mask the compiler keeps as 0 / 1 in a VGPR), stored once before
        # and read once inside each attempt of a tail instance - 10 % of a config-4 batch; its two-waves-per-SIMD budget
        # (256 registers, 20 KB of LDS) is asserted in test_reduced_native_kernels_fit_two_waves_per_simd
        "mpmpc_reduced_tail_kernel<64, 16>": 8,
    }       # (the shipped values, profiles/r4/kernel_resources.txt: a regression of a single dword fails)
    rows = [r for r in _kernel_rows(lib) if "solve_kernel" in r["name"] or "solve_block_kernel" in r["name"] or "reduced_kernel" in r["name"] or "reduced_t_kernel" in r["name"]
            or "reduced_block_kernel" in r["name"]
            or "reduced_tail_kernel" in r["name"]]
    assert rows
    bad = {r["name"]: r["scratch"] for r in rows if r["scratch"] > KNOWN_SCRATCH.get(r["name"], 0)}
    assert not bad, bad
    # the kernels of BASELINE configs 2-5 (N = 30 reduced-native + tail, N = 50 time-optimal) by name
    by = {r["name"]: r for r in rows}
    for name in ("mpmpc_reduced_kernel<64, 16, false>", "mpmpc_reduced_kernel<32, 16, false>", "mpmpc_solve_kernel<64, 16, false, 2>",
                 "mpmpc_reduced_t_kernel<64, 32>", "mpmpc_solve_kernel<64, 32, false, 3>"):
        assert by[name]["scratch"] == 0, by[name]
