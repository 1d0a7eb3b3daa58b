"""N>1 path on CPU: world size 2 over gloo.  Each rank solves its contiguous shard (kernels run in
the CPU emulation here, on the GPU box the same sharding code drives libmpmpc.so), the only
collectives are the MAX of the elapsed time and the optional gather of the controls."""
import os
import sys

import numpy as np
import pytest

import mpmpc
import mpmpc_testlib as T
import scenarios
import sharding

sys.path.insert(0, T.ROOT)
import bench_dist  # noqa: E402


def test_shard_bounds_cover_batch():
    for total in (1, 7, 1024, 65536 + 3):
        for world in (1, 2, 3, 8):
            cuts = [sharding.shard_bounds(total, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(10, 2, 2)


def _worker(rank, world, port, total, N, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = scenarios.sim_track()
    sc = scenarios.make(4, tr, B=total, N=N)
    wp, x0, cc, lb, ub = sharding.shard([sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub], world, rank)
    cfg = T.stock_config(N, sc.weights)
    backend = T.EmuBackend(cfg, mpmpc.default_settings())
    backend.set_path(tr.kappa, tr.v_ref, tr.ds_next)
    sol = backend.solve(wp, x0, cc, lb, ub)
    dt = bench_dist.max_over_ranks(dist, 0.25 * (rank + 1))
    u_all, s_all = bench_dist.gather_controls(dist, sol.u0, sol.status, total)
    # the two bookkeeping calls of bench.py's multi-rank line, end to end: per-rank device ordinal + status counts, and the
    # all-gathered result buffer checked on rank 0 against ONE process solving the whole batch
    def solve_whole():
        one = T.EmuBackend(cfg, mpmpc.default_settings())
        one.set_path(tr.kappa, tr.v_ref, tr.ds_next)
        return one.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    check = bench_dist.gather_check(dist, rank, sol.u0, sol.status, total, solve_whole)
    ranks = bench_dist.rank_reports(dist, 40 + rank, sol.status)
    import json
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump({"check": check, "ranks": ranks}, f)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), u=u_all, s=s_all, dt=dt, n_local=wp.size)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_match_single_process(tmp_path):
    import torch.multiprocessing as mp
    total, N, world = 11, 10, 2          # odd batch: shards of 6 and 5
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, total, N, str(tmp_path)), nprocs=world, join=True)
    tr = scenarios.sim_track()
    sc = scenarios.make(4, tr, B=total, N=N)
    cfg = T.stock_config(N, sc.weights)
    ref = T.EmuBackend(cfg, mpmpc.default_settings())
    ref.set_path(tr.kappa, tr.v_ref, tr.ds_next)
    full = ref.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    got = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    assert [int(g["n_local"]) for g in got] == [6, 5]
    for g in got:                         # every rank holds the whole gathered result
        assert np.array_equal(g["s"], full.status)
        ok = full.status > 0
        assert np.array_equal(g["u"][ok], full.u0[ok])
        assert float(g["dt"]) == 0.5      # MAX over ranks of (0.25, 0.5)
    # bench.py's gather_check / ranks entries (bench_dist.gather_check, rank_reports) as the two ranks produced them
    import json
    rep = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(world)]
    assert rep[1]["check"] == {} and rep[0]["check"]["status_equal"] is True and rep[0]["check"]["instances"] == total
    assert rep[0]["check"]["max_abs_u_diff"] == 0.0
    for r in rep:
        assert [x["device"] for x in r["ranks"]] == [40, 41] and [x["rank"] for x in r["ranks"]] == [0, 1]
        assert sum(x["solved"] + x["solved_inaccurate"] + x["infeasible"] + x["other"] for x in r["ranks"]) == total
        assert sum(x["solved"] for x in r["ranks"]) == int((full.status == 1).sum())
    # without a process group: no gather, one report
    assert bench_dist.gather_check(None, 0, full.u0, full.status, total, None) is None
    assert bench_dist.rank_reports(None, 3, full.status)[0]["device"] == 3


def _run_bench(args, env_extra=None, timeout=300):
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


@pytest.mark.skipif(mpmpc.device_count() > 0, reason="checks the behaviour of a box without GPUs")
def test_bench_gpus_flag_cannot_lie_without_gpus():
    """`bench.py --gpus 2` starts two ranks itself; on a box without GPUs they fail, and so does the command -
    it can no longer print an n_gpus:1 line when asked for more."""
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"])
    assert r.returncode != 0
    assert '"n_gpus"' not in r.stdout
    assert "2-rank run failed" in r.stderr


def test_bench_rejects_a_world_size_that_differs_from_gpus():
    r = _run_bench(["--gpus", "4", "--steps", "1", "--warmup", "0", "--no-cpu"],
                   {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout


@pytest.mark.skipif(mpmpc.device_count() > 0, reason="the rehearsal is for boxes without GPUs")
def test_bench_launcher_rehearsal_with_eight_ranks():
    """`bench.py --gpus 8` end to end on the CPU box - the launcher (this process spawns torch.distributed.run
    with 8 ranks as a child), GPU_MAX_HW_QUEUES=8 in every rank's environment, LOCAL_RANK -> device ordinal of the rank's handle,
    the contiguous shards of the 8 x B batch, the barriers / MAX reductions of the timed loop and the gather check against one
    process solving the whole batch - over gloo, the kernels emulated (--dry-run: value is null, nothing is measured)."""
    import json
    r = _run_bench(["--gpus", "8", "--dry-run", "--config", "5", "--batch", "3", "--steps", "2", "--warmup", "1", "--repeats", "2",
                    "--prewarm", "1", "--no-cpu"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                     # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["value"] is None and d["n_gpus"] == 8 and d["world_size"] == 8
    assert d["gpu_max_hw_queues"] == "8" and d["launches_in_flight"] == 4
    assert [x["rank"] for x in d["ranks"]] == list(range(8)) and [x["device"] for x in d["ranks"]] == list(range(8))
    assert sum(x["solved"] + x["solved_inaccurate"] + x["infeasible"] + x["other"] for x in d["ranks"]) == 24
    assert d["gather_check"]["instances"] == 24 and d["gather_check"]["status_equal"] is True and d["gather_check"]["max_abs_u_diff"] == 0.0
    assert len(d["ms_per_step_by_rank_emulated"]) == 8 and d["config"]["batch_per_gpu"] == 3


@pytest.mark.skipif(mpmpc.device_count() > 0, reason="the rehearsal is for boxes without GPUs")
def test_bench_single_process_rehearsal_with_eight_handles():
    """the torch-free second path of the 8-GPU run - `bench.py --gpus 8 --single-process` - rehearsed with
    eight emulation handles: the line has the torchrun line's schema (ranks with device ordinals and status counts, the
    one-launch-in-flight point, the gather check) and every handle sits on a device of its own."""
    import json
    r = _run_bench(["--gpus", "8", "--single-process", "--dry-run", "--config", "5", "--batch", "5", "--steps", "2", "--warmup", "1",
                    "--repeats", "1"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["value"] is None and d["n_gpus"] == 8 and d["rccl_world_size"] == 0
    assert d["devices"] == list(range(8)) and [x["device"] for x in d["ranks"]] == list(range(8))
    assert sum(x["solved"] + x["solved_inaccurate"] + x["infeasible"] + x["other"] for x in d["ranks"]) == 40
    assert d["gather_check"] == {"same_status": True, "same_u0": True, "instances": 40}
    assert "value_one_launch_in_flight" in d and "one_launch_in_flight" in d and d["config"]["batch_per_gpu"] == 5
