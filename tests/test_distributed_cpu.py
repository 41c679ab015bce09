"""CPU tests (gloo, world 2 and 3) of the multi-GPU orchestration and of the rank launcher.

What runs here is the code a multi-GPU node runs, minus the GPU: the partition arithmetic is the engine's
(dsi_scatter_plan / dsi_plane_range / dsi_argmax_keys_*), the orchestration classes are
`distributed.EngineTemporalFusion` / `EnginePipelinedTemporalFusion`, the ranks are started by
`launch.spawn_ranks` (what `bench.py --gpus N` calls when no launcher started them) and joined by `launch.Dist`;
the voxel arithmetic comes from the CPU oracle through stand-in grids, and the one collective goes through the
host-staged test transport instead of RCCL.  Reference: process2.cpp:211-242 (temporal fusion), SURVEY 8(e)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "distributed_cpu_worker.py")


def run_ranks(world, job, out_dir, extra=()):
    """`world` ranks of tests/distributed_cpu_worker.py through launch.spawn_ranks, in a child interpreter (so that
    the ranks' stderr is a real file).  Returns (status, rank-0 stdout)."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from dvs_mcemvs_amd import launch\n"
            "sys.exit(launch.spawn_ranks(%d, [sys.executable, %r, %r, %r] + %r, n_devices=%d, timeout=300))\n"
            % (ROOT, world, WORKER, job, str(out_dir), list(extra), world))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    return r.returncode, r.stdout, r.stderr


@pytest.mark.parametrize("mode", [0, 1])
def test_two_rank_temporal_fusion_equals_single_process(tmp_path, mode):
    from dvs_mcemvs_amd import distributed as dd, synthetic as syn
    from oracle import oracle as orc
    from oracle_pipeline import OracleMapper

    rc, out, err = run_ranks(2, "temporal", tmp_path, [str(mode)])
    assert rc == 0, err[-3000:]
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    # the launcher's aggregation: rank 0 prints ONE line; units summed over ranks, time = max over ranks
    assert line["world"] == 2 and line["units_all_ranks"] == 4 and line["spawned"] is True
    assert line["elapsed_max"] >= max(line["elapsed_each"]) - 1e-12
    # single-process reference: all 4 slices in order (process2.cpp:98-242)
    rig = syn.stereo_rig(9000, width=40, height=30, duration=0.3, seed=17)
    x, y, ts = rig["events"][0]
    acc = np.zeros((8, 30, 40), np.float32)
    for a, b in dd.subinterval_bounds(x.shape[0], 4):
        m = OracleMapper(rig["cam"], dimZ=8, min_depth=4.0, max_depth=100.0)
        assert m.evaluateDSI((x[a:b], y[a:b], ts[a:b]), rig["trajectories"][0], rig["T_rv_w"])
        acc = orc.accumulate(acc, m.dsi, mode)
    ref = orc.finalize(acc, mode, 4)
    r0 = np.load(tmp_path / "fused_rank0.npy")
    r1 = np.load(tmp_path / "fused_rank1.npy")
    assert np.array_equal(r0, r1)                   # every rank holds the same fused DSI
    # summation order differs (slices 0,2 | 1,3 vs 0,1,2,3): equal to rounding
    assert np.allclose(r0, ref, rtol=1e-6, atol=1e-6)
    assert r0.any()


def test_partitioning():
    from dvs_mcemvs_amd import distributed as dd
    assert dd.subinterval_bounds(10, 4) == [(0, 2), (2, 4), (4, 6), (6, 8)]  # remainder dropped
    assert dd.slices_of_rank(8, 8, 3) == [3]
    assert dd.slices_of_rank(8, 2, 1) == [1, 3, 5, 7]
    owned = sorted(k for r in range(3) for k in dd.slices_of_rank(7, 3, r))
    assert owned == list(range(7))


def test_pipelined_temporal_fusion_rounds(tmp_path):
    """Five fusion rounds through the double-buffered EnginePipelinedTemporalFusion (stand-in grids, program
    order on the CPU): every round equals the single-process harmonic fusion of that round's slices."""
    from oracle import oracle as orc

    rc, out, err = run_ranks(2, "pipelined", tmp_path)
    assert rc == 0, err[-3000:]
    r0 = np.load(tmp_path / "pipe_rank0.npy")
    r1 = np.load(tmp_path / "pipe_rank1.npy")
    assert np.array_equal(r0, r1)
    rng = np.random.default_rng(5)
    for k in range(5):
        slices = rng.uniform(0, 3, (2, 4, 6, 5)).astype(np.float32)
        acc = np.zeros((4, 6, 5), np.float32)
        for s in slices:
            acc = orc.accumulate(acc, s, 1)
        assert np.allclose(r0[k], orc.finalize(acc, 1, 2), rtol=1e-6, atol=1e-7)


def test_plane_ranges_and_argmax_keys(built):
    from dvs_mcemvs_amd import distributed as dd
    from oracle import oracle as orc
    assert dd.plane_ranges(256, 8) == [(32 * r, 32) for r in range(8)]
    assert dd.plane_ranges(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    rng = np.random.default_rng(9)
    dsi = rng.integers(0, 4, (37, 9, 11)).astype(np.float32)    # many ties -> "first maximum wins" matters
    dsi[:, 0, 0] = 0.0                                            # all-zero column -> index 0
    dsi[5, 1, 1] = dsi[30, 1, 1] = 99.0                           # equal maxima in different shards
    conf, idx = orc.collapse_max_z(dsi)
    keys = None
    for b, c in dd.plane_ranges(37, 5):
        cl, il = orc.collapse_max_z(dsi[b:b + c])
        k = dd.pack_argmax_keys(cl, il, b)
        keys = k if keys is None else np.maximum(keys, k)
    c2, i2 = dd.unpack_argmax_keys(keys)
    assert np.array_equal(c2, conf) and np.array_equal(i2, idx) and i2[1, 1] == 5 and i2[0, 0] == 0


def test_plane_sharded_argmax_equals_single_process(tmp_path):
    """configs[4]-style sharding (one big DSI, planes split over ranks, voxel-wise camera fusion local,
    ONE all-reduce(MAX) of packed (confidence, index) keys) gives the unsharded depth-map inputs."""
    from dvs_mcemvs_amd import synthetic as syn
    from oracle import oracle as orc
    from oracle_pipeline import OracleMapper

    rc, out, err = run_ranks(3, "planes", tmp_path)
    assert rc == 0, err[-3000:]
    rig = syn.stereo_rig(6000, width=40, height=30, duration=0.3, seed=23)
    fused = None
    for cam in range(2):
        m = OracleMapper(rig["cam"], dimZ=13, min_depth=4.0, max_depth=100.0)
        assert m.evaluateDSI(rig["events"][cam], rig["trajectories"][cam], rig["T_rv_w"])
        fused = m.dsi.copy() if fused is None else orc.fuse2(fused, m.dsi, 3)
    conf, idx = orc.collapse_max_z(fused)
    for r in range(3):
        z = np.load(tmp_path / ("plane_rank%d.npz" % r))
        assert np.array_equal(z["conf"], conf) and np.array_equal(z["idx"], idx)
    assert conf.max() > 0


@pytest.mark.parametrize("world,nz", [(2, 21), (3, 8), (3, 2)])
def test_reduce_scattered_depth_map_on_stand_in_grids(tmp_path, world, nz):
    """The reduce-scatter form of the temporal fusion with the ENGINE's partition (dsi_scatter_plan): owned planes
    + all-reduced remainder planes, finalize and arg-max of what a rank owns, MAX of the packed keys -- equals
    all-reduce + finalize + arg-max.  dimZ = 21 over 2 ranks and 8 over 3 leave remainder planes; 2 over 3 gives
    q = 0 (every plane is a remainder plane)."""
    from oracle import oracle as orc
    rc, out, err = run_ranks(world, "scattered", tmp_path, [str(nz)])
    assert rc == 0, err[-3000:]
    rng = np.random.default_rng(31)
    slices = rng.uniform(0, 3, (world, nz, 6, 7)).astype(np.float32)
    slices[:, :, 0, 0] = 0.0
    acc = np.zeros((nz, 6, 7), np.float32)
    for s in slices:
        acc = orc.accumulate(acc, s, 1)
    conf, idx = orc.collapse_max_z(orc.finalize(acc, 1, world))
    for r in range(world):
        z = np.load(tmp_path / ("scattered_rank%d.npz" % r))
        assert np.array_equal(z["idx"], idx), r
        assert np.allclose(z["conf"], conf, rtol=1e-6)


@pytest.mark.parametrize("world,nz", [(3, 8), (2, 5)])
def test_pipelined_temporal_fusion_with_the_reduce_scatter_collective(tmp_path, world, nz):
    """EnginePipelinedTemporalFusion(scattered=...) over four rounds -- the class `bench.py --temporal-collective
    reduce_scatter` drives -- equals the single-process temporal HM + arg-max of every round on every rank.  World 3
    with dimZ 8: two owned planes per rank plus two tail planes (process2.cpp:211-242 across ranks)."""
    from oracle import oracle as orc
    rc, out, err = run_ranks(world, "pipelined_scattered", tmp_path, [str(nz)])
    assert rc == 0, err[-3000:]
    rng = np.random.default_rng(77)
    rounds = [rng.uniform(0, 3, (world, nz, 6, 5)).astype(np.float32) for _ in range(4)]
    got = [np.load(tmp_path / ("pipe_scattered_rank%d.npz" % r)) for r in range(world)]
    for k, slices in enumerate(rounds):
        slices[:, :, 0, 0] = 0.0
        acc = np.zeros((nz, 6, 5), np.float32)
        for s in slices:
            acc = orc.accumulate(acc, s, 1)
        conf, idx = orc.collapse_max_z(orc.finalize(acc, 1, world))
        for r in range(world):
            assert np.array_equal(got[r]["idx"][k], idx), (k, r)
            assert np.allclose(got[r]["conf"][k], conf, rtol=1e-6), (k, r)
        assert idx[0, 0] == 0


def test_launcher_refuses_a_smaller_job_and_reports_a_failed_rank(tmp_path):
    """`bench.py --gpus N` on a node with fewer than N devices must fail, not run fewer ranks; a rank that dies
    takes the job down with its status."""
    from dvs_mcemvs_amd import launch
    with open(tmp_path / "err.txt", "w+") as err, open(tmp_path / "out.txt", "w+") as out:
        assert launch.spawn_ranks(2, [sys.executable, "-c", "print('never')"], n_devices=1, out=out, err=err) == 2
        assert launch.spawn_ranks(0, [sys.executable, "-c", "print('never')"], out=out, err=err) == 2
        err.seek(0)
        msg = err.read()
        assert "refusing to run a smaller job" in msg and "1 GPU device" in msg
        out.seek(0)
        assert out.read() == ""
        code = "import os, sys, time; r = int(os.environ['RANK']); print('rank', r); time.sleep(0 if r == 1 else 30); sys.exit(7 if r == 1 else 0)"
        t0 = __import__("time").time()
        assert launch.spawn_ranks(3, [sys.executable, "-c", code], n_devices=3, out=out, err=err) == 7
        assert __import__("time").time() - t0 < 20          # the sleeping ranks were stopped, not waited for
    # bench.py itself: a launcher's WORLD_SIZE that contradicts --gpus is an error, not a note
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode != 0 and "contradicts WORLD_SIZE" in r.stderr


def test_bench_refuses_more_ranks_than_devices(built):
    """On this box (no GPU; on the 1-GPU lease the GPU test repeats it with --gpus 2) `python bench.py --gpus 2`
    exits non-zero with a message that names the device count, and prints no JSON line."""
    import dvs_mcemvs_amd as d
    n_dev = d.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(2, n_dev + 1)), "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 2, r.stderr[-2000:]
    assert "refusing to run a smaller job" in r.stderr and ("%d GPU device" % n_dev) in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_launch_env_defaults_yield_to_the_environment_and_are_reported():
    """launch.py's two environment defaults are guesses until an 8-GPU node has run -- the caller's
    environment wins, DSI_LAUNCH_NO_ENV_DEFAULTS=1 sets neither, and environment_report says what a rank runs with."""
    from dvs_mcemvs_amd import launch
    env = launch.rank_environment({}, 1, 4, 29999)
    assert env["RANK"] == "1" and env["WORLD_SIZE"] == "4" and env["MASTER_ADDR"] == "127.0.0.1"
    assert env["NCCL_SOCKET_IFNAME"] == "lo" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    rep = launch.environment_report(env)
    assert rep["NCCL_SOCKET_IFNAME"] == {"value": "lo", "is_launch_default": True} and not rep["defaults_disabled"]
    env = launch.rank_environment({"NCCL_SOCKET_IFNAME": "eth0"}, 0, 2, 29999)
    assert env["NCCL_SOCKET_IFNAME"] == "eth0"
    assert launch.environment_report(env)["NCCL_SOCKET_IFNAME"] == {"value": "eth0", "is_launch_default": False}
    env = launch.rank_environment({launch.NO_DEFAULTS_ENV: "1"}, 0, 2, 29999)
    assert "NCCL_SOCKET_IFNAME" not in env and "HSA_ENABLE_IPC_MODE_LEGACY" not in env
    rep = launch.environment_report(env)
    assert rep["defaults_disabled"] and rep["HSA_ENABLE_IPC_MODE_LEGACY"]["value"] is None
