"""ShardedBatchMPC / ShardedHandles (multi-purpose-mpc_amd/sharded.py): one process, one handle per device, the batch cut into
contiguous shards, every shard started before any is collected (the loop it serves: src/simulation.py:134-140).
CPU: a world of emulation backends.  GPU: two and three handles on device 0 against one."""
import numpy as np
import pytest

import mpmpc
import mpmpc_testlib as T
import scenarios
import sharded


@pytest.mark.parametrize("world,B", [(2, 7), (3, 10), (4, 3)])
def test_sharded_handles_concatenate_in_instance_order_cpu(world, B, emu, track):
    """ragged shards (B not a multiple of the world, a world larger than the batch): the sharded result is the single
    backend's, bit for bit"""
    sc = scenarios.make(4, track, B=B, N=10)
    cfg = T.stock_config(sc.N, sc.weights)
    mk = lambda: T.EmuBackend(cfg, mpmpc.default_settings(), emu)
    one = mk()
    one.set_path(track.kappa, track.v_ref, track.ds_next)
    ref = one.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    sh = sharded.ShardedHandles([mk() for _ in range(world)])
    sh.set_path(track.kappa, track.v_ref, track.ds_next)
    got = sh.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    for name in ("z", "u0", "status", "iters", "resid", "y"):
        assert np.array_equal(getattr(got, name), getattr(ref, name)), name
    assert [hi - lo for lo, hi in sh.bounds(B)] == [B // world + (1 if r < B % world else 0) for r in range(world)]


@pytest.mark.gpu
@pytest.mark.parametrize("cfgid,B,world", [(4, 2050, 2), (2, 1000, 3), (3, 300, 2)])
def test_sharded_handles_on_one_device_match_one_handle(cfgid, B, world, track):
    """`world` handles on device 0 (what a node with that many GPUs runs, one per device) against one handle solving the
    whole batch - with per-instance corridor rows and from the corridor table."""
    sc = scenarios.make(cfgid, track, B=B)
    table = "obstacles" if sc.obstacles else "free"
    ub, lb = (track.ub_obstacles, track.lb_obstacles) if sc.obstacles else (track.ub_free, track.lb_free)

    def handle(n):
        h = mpmpc.Handle(T.stock_config(sc.N, sc.weights, max_batch=n), mpmpc.default_settings())
        h.set_path(track.kappa, track.v_ref, track.ds_next)
        h.set_corridor(ub, lb)
        return h

    one = handle(B)
    ref = one.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    ref_tab = one.solve(sc.wp_id, sc.x0, sc.cc_prev, want_y=False)
    one.close()
    sh = sharded.ShardedHandles([handle(-(-B // world)) for _ in range(world)])

    def same(got, ref, y=True):
        # statuses exactly; the numbers to rounding level: a shard's instances share their packed waves with other partners
        # than in the whole batch, and the refinement loops of a wave run until its slowest instance is done
        # (tests/test_gpu_parity.py::test_every_lane_packing_gives_the_same_answers)
        assert np.array_equal(got.status, ref.status) and np.array_equal(got.iters[:, 0], ref.iters[:, 0])
        ok = ref.status == 1
        assert np.max(np.abs(got.u0[ok] - ref.u0[ok])) <= 1e-9 and np.max(np.abs(got.z[ok] - ref.z[ok])) <= 1e-9
        if y:
            assert np.max(np.abs(got.y[ok] - ref.y[ok])) <= 1e-7

    for _ in range(2):                                   # (twice: the handles' double-buffered slots swap between calls)
        same(sh.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True), ref)
    got = sh.solve(sc.wp_id, sc.x0, sc.cc_prev)
    same(got, ref_tab, y=False)
    assert got.y is None
    sh.close()
    del table


@pytest.mark.gpu
@pytest.mark.parametrize("N,weights", [(70, "stock"), (150, "stock"), (70, "time_optimal")])
def test_sharded_handles_at_long_horizons(N, weights, track, emu):
    """several handles of one process at horizons above 63 - the kernels there need a dynamic-LDS attribute that is
    set once per function AND DEVICE (it used to be once per process).  Two handles on device 0 (the box has one GPU; the
    per-device bookkeeping is what an 8-GPU node exercises) against one handle: the pair kernel (N = 70, stock weights), the
    256-lane workgroup kernels (N = 150) and the general workgroup kernel (time-optimal weights)."""
    B = 96
    tw = T.wide_track(track, emu, N)
    sc = scenarios.make(4, tw, B=B, N=N)
    sc.weights = weights

    def handle(n):
        h = mpmpc.Handle(T.stock_config(N, weights, max_batch=n), mpmpc.default_settings())
        h.set_path(track.kappa, track.v_ref, track.ds_next)
        return h
    one = handle(B)
    ref = one.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    one.close()
    sh = sharded.ShardedHandles([handle(B // 2), handle(B // 2)])
    got = sh.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    sh.close()
    assert np.array_equal(got.status, ref.status) and np.array_equal(got.iters, ref.iters)
    ok = ref.status == 1
    assert ok.sum() >= B // 2 and np.array_equal(got.u0[ok], ref.u0[ok]) and np.array_equal(got.z[ok], ref.z[ok])


@pytest.mark.gpu
def test_sharded_batch_mpc_is_batch_mpc():
    """the class with the reference's constructor arguments, two handles on device 0, against BatchMPC"""
    import test_host_mpc as H
    from MPC import BatchMPC
    from scipy import sparse
    m, rp, car = H.build_world()
    Q, R, QN = sparse.diags([1.0, 0.0, 0.0]), sparse.diags([0.5, 0.0]), sparse.diags([1.0, 0.0, 0.0])
    ic = {'umin': np.array([0.0, -np.tan(0.66) / car.length]), 'umax': np.array([1.0, np.tan(0.66) / car.length])}
    scn = {'xmin': np.array([-np.inf] * 3), 'xmax': np.array([np.inf] * 3)}
    tr = scenarios.sim_track()
    sc = scenarios.make(4, tr, B=600)
    corridor = (tr.ub_obstacles, tr.lb_obstacles)
    bm = BatchMPC(car, 30, Q, R, QN, scn, ic, 4.0, max_batch=sc.B, corridor=corridor)
    u, plan, status, _ = bm.get_control_batch(sc.wp_id, sc.x0, sc.cc_prev)
    sm = sharded.ShardedBatchMPC(car, 30, Q, R, QN, scn, ic, 4.0, max_batch=sc.B, devices=[0, 0], corridor=corridor)
    u2, plan2, status2, _ = sm.get_control_batch(sc.wp_id, sc.x0, sc.cc_prev)
    ok = status == 1
    assert np.array_equal(status, status2) and np.max(np.abs(u[ok] - u2[ok])) <= 1e-9 and np.max(np.abs(plan[ok] - plan2[ok])) <= 1e-9
    sm.close()


@pytest.mark.gpu
def test_non_diagonal_stage_weights_are_accepted_by_every_product_class():
    """Round 5: `Q` and `R` may be any symmetric positive semidefinite matrices, like in the reference
    (src/MPC.py:150 puts the whole matrices into P; its cost vector uses only their diagonals, src/MPC.py:153-155 - reproduced).
    MPC, BatchMPC and ShardedBatchMPC take them and agree with each other; what is still refused - loudly, before anything is
    launched - is a matrix that is not symmetric or not positive semidefinite (ValueError from the host classes, MPMPC_E_ARG
    from the C side for a caller that fills mpmpc_config itself)."""
    import test_host_mpc as H
    from MPC import MPC, BatchMPC
    from scipy import sparse
    m, rp, car = H.build_world()
    ic = {'umin': np.array([0.0, -np.tan(0.66) / car.length]), 'umax': np.array([1.0, np.tan(0.66) / car.length])}
    scn = {'xmin': np.array([-np.inf] * 3), 'xmax': np.array([np.inf] * 3)}
    Q, R, QN = sparse.diags([1.0, 0.0, 0.0]), sparse.diags([0.5, 0.0]), sparse.diags([1.0, 0.0, 0.0])
    Qf = np.array([[1.0, 0.1, 0.0], [0.1, 0.5, 0.0], [0.0, 0.0, 0.0]])
    Rf = np.array([[0.5, 0.05], [0.05, 0.1]])
    QNf = np.array([[1.0, 0.1, 0.0], [0.1, 0.5, 0.0], [0.0, 0.0, 0.2]])
    tr = scenarios.sim_track()
    sc = scenarios.make(4, tr, B=64, N=10)
    outs = []
    for q, r in ((Qf, R), (Q, Rf), (sparse.csc_matrix(Qf), sparse.csc_matrix(Rf))):
        bm = BatchMPC(car, 10, q, r, QNf, scn, ic, 4.0, max_batch=sc.B)
        u, plan, status, _ = bm.get_control_batch(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
        sm = sharded.ShardedBatchMPC(car, 10, q, r, QNf, scn, ic, 4.0, max_batch=sc.B, devices=[0, 0])
        u2, plan2, status2, _ = sm.get_control_batch(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
        assert np.array_equal(status, status2) and np.array_equal(u, u2) and (status == 1).mean() > 0.5
        outs.append(u[status == 1])
        bm.close()
        sm.close()
    assert np.max(np.abs(outs[0] - outs[1])) > 1e-6            # different weights, different optima
    # the single-car class: one step of the reference's loop
    mpc = MPC(car, 10, Qf, Rf, QNf, scn, ic, 4.0)
    car.s = 0.3
    u = mpc.get_control()
    assert u.shape == (2,) and np.all(np.isfinite(u)) and mpc.last_status in (1, 2, -3)
    with pytest.raises(ValueError):
        BatchMPC(car, 10, np.array([[1.0, 0.1, 0.0], [0.2, 0.5, 0.0], [0.0, 0.0, 0.0]]), R, QN, scn, ic, 4.0, max_batch=4)      # not symmetric
    for bad_q, bad_r in ((np.array([[1.0, 2.0, 0.0], [2.0, 0.5, 0.0], [0.0, 0.0, 0.0]]), R), (Q, np.array([[0.5, 1.0], [1.0, 0.1]]))):
        with pytest.raises(ValueError):                                                                                         # indefinite
            BatchMPC(car, 10, bad_q, bad_r, QN, scn, ic, 4.0, max_batch=4)
    cfg = T.stock_config(10, max_batch=4)
    cfg.QN_offdiag[0] = 5.0                                                              # indefinite: 1 x 0 - 25 < 0
    with pytest.raises(mpmpc.MpmpcError):
        mpmpc.Handle(cfg, mpmpc.default_settings())


@pytest.mark.gpu
def test_eight_handles_with_ragged_shards_reproduce_one_handle_at_65536():
    """the first 8-GPU run's data path, on one device - eight handles (device 0), ragged contiguous shards
    of a 65 536 + 5 instance config-5 batch through ShardedHandles (upload + resident launch of every shard STARTED before any
    is collected) against ONE handle solving the whole batch: bit for bit.  The single handle's batch is above the 64 MiB
    staging limit, so its upload goes through the page-locked bounce buffer (item 7b): same bits again."""
    tr = scenarios.sim_track()
    B = 65536 + 5
    sc = scenarios.make(5, tr, B=B)
    world = 8
    per = -(-B // world)
    hs = []
    for r in range(world):
        h = mpmpc.Handle(T.stock_config(sc.N, sc.weights, max_batch=per))
        h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
        hs.append(h)
    sh = sharded.ShardedHandles(hs)
    sizes = [hi - lo for lo, hi in sh.bounds(B)]
    assert sum(sizes) == B and max(sizes) - min(sizes) == 1          # ragged
    got = sh.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    sh.close()
    one_h = mpmpc.Handle(T.stock_config(sc.N, sc.weights, max_batch=B))
    one_h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
    with pytest.raises(mpmpc.MpmpcError):
        one_h.staging(16)                                              # (no staging blocks at this size: the bounce path it is)
    one = one_h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    # ... and the resident path after such an upload
    one_h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    one_h.solve_resident(B)
    two = one_h.download(B)
    one_h.close()
    for a in (got, two):
        assert np.array_equal(a.status, one.status) and np.array_equal(a.iters, one.iters)
        assert np.array_equal(a.u0, one.u0) and np.array_equal(a.z, one.z) and np.array_equal(a.resid, one.resid)
    assert (one.status == 1).mean() > 0.85 and (one.status == -3).sum() > 3000
