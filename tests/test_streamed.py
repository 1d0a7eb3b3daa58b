"""StreamedBatches (multi-purpose-mpc_amd/streamed.py): several host-buffer calls in flight on one device, results in submission
order.  CPU: the ordering logic with emulation-backed stand-ins for the handles; GPU: real handles against mpmpc_solve."""
import numpy as np
import pytest

import mpmpc
import mpmpc_testlib as T
import scenarios
import streamed


class _EmuStagedHandle:
    """the staging / staged_begin / staged_end surface of mpmpc.Handle on the CPU emulation: begin remembers, end solves"""

    def __init__(self, emu, cfg, settings, track):
        self.emu, self.cfg, self.st, self.track = emu, cfg, settings, track
        self.v, self.pending, self.calls = None, None, 0

    def set_path(self, *a):
        pass

    def staging(self, B):
        N = self.cfg.N
        n, m = 5 * N + 3, 8 * N + 6
        self.v = dict(wp_id=np.zeros(B, np.int32), x0=np.zeros((B, 3)), cc_prev=np.zeros((B, 2 * N)), lb=np.zeros((B, N)), ub=np.zeros((B, N)),
                      z=np.zeros((B, n)), u0=np.zeros((B, 2)), status=np.zeros(B, np.int32), iters=np.zeros((B, 2), np.int32),
                      resid=np.zeros((B, 2)), y=np.zeros((B, m)))
        return self.v

    def staged_begin(self, B, with_rows=True, want_z=True, want_y=False):
        assert self.pending is None, "one call in flight per handle"
        self.pending = (B, want_z, want_y)
        self.calls += 1

    def staged_end(self):
        B, want_z, want_y = self.pending
        self.pending = None
        v = self.v
        qp = self.emu.assemble(self.cfg, self.track, (v["wp_id"], v["x0"], v["cc_prev"], v["lb"], v["ub"]))
        sol, _ = self.emu.solve_launch(self.cfg, self.st, qp, G=64)
        v["z"][:] = sol.z; v["u0"][:] = sol.u0; v["status"][:] = sol.status; v["iters"][:] = sol.iters; v["resid"][:] = sol.resid; v["y"][:] = sol.y


def _batches(track, n, B):
    out = []
    for i in range(n):
        sc = scenarios.make(4 if i % 2 else 2, track, B=B)
        r = np.random.default_rng(100 + i).permutation(B)
        out.append((sc.wp_id[r], sc.x0[r], sc.cc_prev[r], sc.lb[r], sc.ub[r]))
    return out, sc


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_streamed_batches_come_back_in_order(depth, emu, track):
    B = 6
    batches, sc = _batches(track, 7, B)
    cfg = T.stock_config(sc.N, sc.weights, max_batch=B)
    st = mpmpc.default_settings()
    hs = [_EmuStagedHandle(emu, cfg, st, track) for _ in range(depth)]
    sb = streamed.StreamedBatches(handles=hs)
    got = list(sb.map(batches, want_y=True))
    assert len(got) == len(batches) and sum(h.calls for h in hs) == len(batches)
    assert max(h.calls for h in hs) - min(h.calls for h in hs) <= 1          # round robin
    for b, s in zip(batches, got):
        qp = emu.assemble(cfg, track, b)
        ref, _ = emu.solve_launch(cfg, st, qp, G=64)
        assert np.array_equal(s.status, ref.status) and np.array_equal(s.z, ref.z) and np.array_equal(s.u0, ref.u0)
        assert np.array_equal(s.y, ref.y) and np.array_equal(s.iters, ref.iters)
    # submit / drain by hand: nothing comes back before the ring is full, everything after a drain
    sb = streamed.StreamedBatches(handles=[_EmuStagedHandle(emu, cfg, st, track) for _ in range(depth)])
    early = [sb.submit(*b, want_z=False) for b in batches[:depth]]
    assert all(e is None for e in early)
    first = sb.submit(*batches[depth])
    assert first is not None and first.z is None            # (the first batch was submitted with want_z = False)
    rest = sb.drain()
    assert len(rest) == depth and sb.drain() == []
    with pytest.raises(ValueError):
        streamed.StreamedBatches(handles=[])


def test_abandoned_stream_leaves_nothing_in_flight(emu, track):
    """a consumer that breaks out of map() early, or a submit that fails, must not leave batches in flight - the next
    stream would hand their stale Solutions out as the results of its own first batches."""
    B, depth = 4, 3
    batches, sc = _batches(track, 8, B)
    cfg = T.stock_config(sc.N, sc.weights, max_batch=B)
    st = mpmpc.default_settings()
    hs = [_EmuStagedHandle(emu, cfg, st, track) for _ in range(depth)]
    sb = streamed.StreamedBatches(handles=hs)
    ref = [emu.solve_launch(cfg, st, emu.assemble(cfg, track, b), G=64)[0] for b in batches]
    for k, s in enumerate(sb.map(batches)):          # (1) early break: batches 1 .. depth are in flight at this point
        assert np.array_equal(s.z, ref[k].z)
        if k == 1:
            break
    assert not sb._order and all(b is None for b in sb._busy) and all(h.pending is None for h in hs)
    got = list(sb.map(batches[4:]))                  # the next stream: its own results, from its first batch on
    assert len(got) == 4 and all(np.array_equal(g.z, r.z) and np.array_equal(g.status, r.status) for g, r in zip(got, ref[4:]))
    # (2) a generator dropped without being exhausted or closed explicitly
    g = sb.map(batches)
    next(g)
    del g
    import gc
    gc.collect()
    assert not sb._order and all(h.pending is None for h in hs)
    # (3) a batch that cannot be started: the batches before it are still delivered in order, the error surfaces, and the
    # ring is clean afterwards
    bad = list(batches[:5])
    bad[4] = (bad[4][0], np.zeros((B + 1, 3)), bad[4][2], bad[4][3], bad[4][4])       # x0 of the wrong shape
    seen = []
    with pytest.raises(ValueError):
        for s in sb.map(bad):
            seen.append(s)
    assert len(seen) == 2 and all(np.array_equal(s.z, r.z) for s, r in zip(seen, ref))     # batches 0 and 1 came back, in order
    assert not sb._order and all(b is None for b in sb._busy) and all(h.pending is None for h in hs)
    # ... submit() by hand: the Solution collected to make room rides on the exception
    for b in batches[:depth]:
        assert sb.submit(*b) is None
    with pytest.raises(ValueError) as ei:
        sb.submit(*bad[4])
    assert ei.value.done is not None and np.array_equal(ei.value.done.z, ref[0].z)
    rest = sb.drain()
    assert len(rest) == depth - 1 and np.array_equal(rest[0].z, ref[1].z)
    assert sb.submit(*batches[3]) is None            # the failed batch took no slot: the ring goes on where it was
    assert np.array_equal(sb.drain()[0].z, ref[3].z)


def test_discard_drains_every_slot_when_one_of_them_fails(emu, track):
    """a staged_end that raises inside discard() must not leave the other slots marked busy - all are drained, the
    first error surfaces once, and the ring starts again at its first handle."""
    B, depth = 4, 3
    batches, sc = _batches(track, 6, B)
    cfg = T.stock_config(sc.N, sc.weights, max_batch=B)
    st = mpmpc.default_settings()
    hs = [_EmuStagedHandle(emu, cfg, st, track) for _ in range(depth)]
    sb = streamed.StreamedBatches(handles=hs)
    for b in batches[:depth]:
        assert sb.submit(*b) is None
    good_end = hs[1].staged_end

    def failing_end():
        hs[1].pending = None
        raise mpmpc.MpmpcError("device error (injected)")
    hs[1].staged_end = failing_end
    with pytest.raises(mpmpc.MpmpcError):
        sb.discard()
    assert not sb._order and all(b is None for b in sb._busy) and all(h.pending is None for h in hs) and sb._next == 0
    hs[1].staged_end = good_end
    ref = [emu.solve_launch(cfg, st, emu.assemble(cfg, track, b), G=64)[0] for b in batches]
    got = list(sb.map(batches))
    assert len(got) == len(batches) and all(np.array_equal(g.z, r.z) for g, r in zip(got, ref))


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [1, 3])
def test_streamed_batches_on_device_match_the_single_call(depth, track):
    B = 512
    batches, sc = _batches(track, 8, B)
    cfg = T.stock_config(sc.N, sc.weights, max_batch=B)
    sb = streamed.StreamedBatches(cfg, mpmpc.default_settings(), depth=depth)
    sb.set_path(track.kappa, track.v_ref, track.ds_next)
    # (views instead of copies: a result stays valid until `depth` more batches have been submitted)
    sv = streamed.StreamedBatches(cfg, mpmpc.default_settings(), depth=depth, copy=False)
    sv.set_path(track.kappa, track.v_ref, track.ds_next)
    n_seen = 0
    for b, s in zip(batches, sv.map(batches)):
        assert s.z.base is not None and np.all(np.isfinite(s.u0))
        n_seen += 1
    assert n_seen == len(batches)
    sv.close()
    ref_h = mpmpc.Handle(cfg, mpmpc.default_settings())
    ref_h.set_path(track.kappa, track.v_ref, track.ds_next)
    got = list(sb.map(batches, want_y=True))
    assert len(got) == len(batches)
    for b, s in zip(batches, got):
        ref = ref_h.solve(*b, want_y=True)
        assert np.array_equal(s.status, ref.status) and np.array_equal(s.iters, ref.iters)
        assert np.array_equal(s.z, ref.z) and np.array_equal(s.u0, ref.u0) and np.array_equal(s.y, ref.y) and np.array_equal(s.resid, ref.resid)
    assert any((s.status == mpmpc.PRIMAL_INFEASIBLE).any() for s in got)      # the obstacle batches bring their tails along
    sb.close()


@pytest.mark.gpu
def test_batch_mpc_stream_is_batch_mpc_call_by_call():
    """BatchMPC.get_control_stream: the class with the reference's constructor arguments, three calls in flight, against
    get_control_batch on the same batches (corridor from the controller's table, and corridor rows coming with the batch)."""
    import test_host_mpc as H
    from MPC import BatchMPC
    from scipy import sparse
    m, rp, car = H.build_world()
    Q, R, QN = sparse.diags([1.0, 0.0, 0.0]), sparse.diags([0.5, 0.0]), sparse.diags([1.0, 0.0, 0.0])
    ic = {'umin': np.array([0.0, -np.tan(0.66) / car.length]), 'umax': np.array([1.0, np.tan(0.66) / car.length])}
    scn = {'xmin': np.array([-np.inf] * 3), 'xmax': np.array([np.inf] * 3)}
    tr = scenarios.sim_track()
    B = 300
    bm = BatchMPC(car, 30, Q, R, QN, scn, ic, 4.0, max_batch=B, corridor=(tr.ub_obstacles, tr.lb_obstacles))
    batches = []
    for i in range(5):
        sc = scenarios.make(4, tr, B=B)
        r = np.random.default_rng(7 + i).permutation(B)
        batches.append((sc.wp_id[r], sc.x0[r], sc.cc_prev[r]) if i % 2 else (sc.wp_id[r], sc.x0[r], sc.cc_prev[r], sc.lb[r], sc.ub[r]))
    want = [bm.get_control_batch(*b) for b in batches]
    got = list(bm.get_control_stream(batches, depth=3))
    assert len(got) == len(want)
    for (u, plan, status, _), (u2, plan2, status2) in zip(want, got):
        assert np.array_equal(status, status2) and np.array_equal(u, u2) and np.array_equal(plan, plan2)
    u3 = [g[0] for g in bm.get_control_stream(batches, depth=3, want_plan=False)]
    assert all(np.array_equal(a[0], b) for a, b in zip(want, u3))


@pytest.mark.gpu
def test_streamed_batches_of_changing_size(track):
    """The staging blocks of a handle are laid out for the batch size of the call: a stream whose batches change size re-lays
    them out between calls (never under a call in flight) and still returns every batch's results, in order."""
    sizes = [256, 100, 256, 17, 300, 1, 256]
    full = scenarios.make(4, track, B=300)
    cfg = T.stock_config(full.N, full.weights, max_batch=300)
    batches = []
    for i, B in enumerate(sizes):
        r = np.random.default_rng(50 + i).permutation(300)[:B]
        batches.append((full.wp_id[r], full.x0[r], full.cc_prev[r], full.lb[r], full.ub[r]))
    sb = streamed.StreamedBatches(cfg, mpmpc.default_settings(), depth=3)
    sb.set_path(track.kappa, track.v_ref, track.ds_next)
    ref_h = mpmpc.Handle(cfg, mpmpc.default_settings())
    ref_h.set_path(track.kappa, track.v_ref, track.ds_next)
    got = list(sb.map(batches))
    assert [g.status.size for g in got] == sizes
    for b, s in zip(batches, got):
        ref = ref_h.solve(*b)
        assert np.array_equal(s.status, ref.status) and np.array_equal(s.z, ref.z) and np.array_equal(s.u0, ref.u0)
    sb.close()


@pytest.mark.gpu
def test_batch_mpc_stream_ring_follows_corridor_and_settings():
    """the further handles of get_control_stream must carry what the controller's own handle carries - a corridor
    table built on the DEVICE (corridor="device"), a table rebuilt after the map changed, changed settings - and a ring of
    another depth closes the handles of the old one."""
    import test_host_mpc as H
    from MPC import BatchMPC
    from map import Obstacle
    from scipy import sparse
    m, rp, car = H.build_world(obstacles=False)
    Q, R, QN = sparse.diags([1.0, 0.0, 0.0]), sparse.diags([0.5, 0.0]), sparse.diags([1.0, 0.0, 0.0])
    ic = {'umin': np.array([0.0, -np.tan(0.66) / car.length]), 'umax': np.array([1.0, np.tan(0.66) / car.length])}
    scn = {'xmin': np.array([-np.inf] * 3), 'xmax': np.array([np.inf] * 3)}
    tr = scenarios.sim_track()
    B = 200
    bm = BatchMPC(car, 30, Q, R, QN, scn, ic, 4.0, max_batch=B, corridor="device")
    batches = []
    for i in range(6):
        sc = scenarios.make(4, tr, B=B)
        r = np.random.default_rng(70 + i).permutation(B)
        batches.append((sc.wp_id[r], sc.x0[r], sc.cc_prev[r]))            # no corridor rows: every handle needs the table

    def same(want, got):
        assert len(got) == len(want)
        for (u, plan, status, _), (u2, plan2, status2) in zip(want, got):
            assert np.array_equal(status, status2) and np.array_equal(u, u2) and np.array_equal(plan, plan2)

    want_free = [bm.get_control_batch(*b) for b in batches]
    same(want_free, list(bm.get_control_stream(batches, depth=3)))          # (the 2nd / 3rd handle used to raise MPMPC_E_STATE here)
    # the map changes: the rebuilt table must reach the whole ring
    m.add_obstacles([Obstacle(cx=c[0], cy=c[1], radius=c[2]) for c in
                     ((0.0, 0.0, 0.05), (-0.8, -0.5, 0.08), (-0.3, -1.0, 0.08), (0.73, -0.9, 0.07), (1.2, 0.0, 0.08))])
    bm.update_corridor_from_map()
    want_obs = [bm.get_control_batch(*b) for b in batches]
    assert any(not np.array_equal(a[2], b[2]) or not np.array_equal(a[0], b[0]) for a, b in zip(want_free, want_obs))
    same(want_obs, list(bm.get_control_stream(batches, depth=3)))
    # settings through the controller reach every handle
    bm.set_settings(mpmpc.default_settings(phase1_accept=0))
    want_strict = [bm.get_control_batch(*b) for b in batches]
    assert not any((w[2] == mpmpc.SOLVED_INACCURATE).any() for w in want_strict)
    same(want_strict, list(bm.get_control_stream(batches, depth=3)))
    # another depth: the old ring's handles are closed, the new ring is complete
    old = list(bm._ring)
    same(want_strict, list(bm.get_control_stream(batches, depth=2)))
    assert len(old) == 2 and all(not h._h for h in old) and len(bm._ring) == 1
    bm.close()
