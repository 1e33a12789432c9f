"""The N > 1 plumbing on CPU: two gloo ranks run the functions bench.py itself uses for its control plane and for
its host-staged exchange (robigo_luculenta_amd/distributed.py) -- the communicator-id broadcast, the sum of the
ranks' plot buffers onto rank 0, max/sum aggregation of timings.  The device kernels need a GPU, so the ranks'
buffers come from the CPU oracle here (stream = rank, exactly the sharding bench.py uses); the GPU leg of the same
path is tests/test_gpu_multi.py."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
W, H, PATHS_PER_BATCH, BATCHES = 48, 27, 4096, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_buffer(rank):
    sys.path.insert(0, HERE)
    import _oracle as O
    objs, cam = O.demo_scene_desc()
    scene = O.Scene(objs, cam)
    xyz = np.zeros((W * H, 3), np.float32)
    for b in range(BATCHES):   # every rank renders the same batch indices with its own RNG stream
        ph, _ = scene.render(W, H, 1, rank, b * PATHS_PER_BATCH, PATHS_PER_BATCH)
        O.plot(W, H, ph, xyz)
    return xyz


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from robigo_luculenta_amd import distributed as D
    assert D.env_rank() == (rank, rank, world)
    assert D.init_control_plane(rank, world)
    # the 128-byte communicator id travels from rank 0 to everybody (rl_comm_unique_id -> rl_comm_init_rank)
    uid = bytes(range(128)) if rank == 0 else None
    assert D.broadcast_bytes(uid, 128, root=0) == bytes(range(128))
    # which GPU is whose: the ranks compare what their devices ARE (host / PCI bus id), whatever their numbers are
    assert D.all_gather_strings("host/0000:%02x:00.0" % (0xc1 + rank)) == ["host/0000:c1:00.0", "host/0000:c2:00.0"]
    assert D.one_gpu_per_rank(["h/a", "h/b"]) and not D.one_gpu_per_rank(["h/a", "h/a"])
    assert (D.pick_device(3, 8), D.pick_device(3, 1), D.pick_device(5, 4)) == (3, 0, 1)
    mine = _rank_buffer(rank)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), mine)
    summed = mine.copy()
    holds_sum = D.host_staged_reduce(summed, root=0)
    assert holds_sum == (rank == 0)
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), summed)
    elapsed, (rays, paths) = D.aggregate(1.0 + rank, [10.0 + rank, 100.0])
    assert elapsed == float(world) and rays == 10.0 * world + sum(range(world)) and paths == 100.0 * world
    D.shutdown()


def test_single_rank_helpers_are_no_ops():
    from robigo_luculenta_amd import distributed as D
    assert D.init_control_plane(0, 1) is False
    buf = np.ones(12, np.float32)
    assert D.host_staged_reduce(buf) is True and (buf == 1).all()
    assert D.aggregate(2.5, [3, 4]) == (2.5, [3.0, 4.0])


def test_two_rank_gloo_exchange_matches_sum_of_streams(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / ("rank%d.npy" % r)) for r in range(world)]
    reduced = np.load(tmp_path / "reduced.npy").reshape(-1, 3)
    assert reduced.tobytes() == (parts[0] + parts[1]).tobytes()   # two operands: the float sum has one order
    assert parts[0].any() and parts[1].any() and not np.array_equal(parts[0], parts[1])
    # disjoint RNG streams: two independent estimates of the same image
    assert abs(parts[0].sum() / parts[1].sum() - 1.0) < 0.25


def test_the_n_rank_bench_line_says_what_the_exchange_ran_on_and_what_it_cost():
    """a SCALE record must answer "did RCCL see N ranks" and "where did the time go" by itself.  bench.py's
    --dry-run sends made-up counters of two CPU ranks through the control plane and line assembly of a real run:
    value = SUM of the ranks' rays / MAX of their times, config.rccl = what every rank's communicator reports,
    exchange.ms_per_step = MAX over ranks, and --total-paths turns the line into a strong-scaling one."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run", "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--launches-per-step", "2", "--total-paths", str(2 * 2 * 64 * 1000)],
                         capture_output=True, timeout=300, env=dict(os.environ, MASTER_PORT=str(_free_port())))
    assert run.returncode == 0, run.stderr.decode()[-2000:]
    line = json.loads([l for l in run.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["unit"] == "Mrays/s"
    assert line["config"]["total_paths_per_step"] == 2 * 2 * 64 * 1000 and line["config"]["paths_per_launch"] == 64 * 1000
    paths = 64 * 1000 * 2 * 3                                    # per rank over the 3 steps
    assert line["value"] == pytest.approx(paths * (3.0 + 4.0) / 1.25 / 1e6)   # rank r: (3 + r) rays per path, 1 + r / 4 seconds
    rccl = line["config"]["rccl"]
    assert rccl["world"] == 2 and rccl["ranks_agree"] is True and "backend_used" in rccl and "version" in rccl
    ex = line["exchange"]
    assert ex["ms_per_step"] == 3.0 and ex["per_step"] == 1.0 and 0 < ex["share_of_step"] < 1   # rank r: 2 + r ms
    weak = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run", "--gpus", "2", "--steps", "1", "--warmup", "0"],
                          capture_output=True, timeout=300, env=dict(os.environ, MASTER_PORT=str(_free_port())))
    line = json.loads([l for l in weak.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "weak" and line["config"]["paths_per_launch"] == 1024 * 524288
    # the line answers ">= 90 % of linear?" by itself -- the one-GPU rate of the same run (rank 0 alone) and BOTH
    # scaling modes against it, the headline's mode marked
    sd = line["scaling_detail"]
    assert sd["n1_same_run"]["value"] > 0 and sd["weak"]["headline"] is True and sd["strong"]["headline"] is False
    assert sd["weak"]["value"] == pytest.approx(line["value"])
    for mode in ("weak", "strong"):
        assert sd[mode]["efficiency"] == pytest.approx(sd[mode]["value"] / (2 * sd["n1_same_run"]["value"]))
    # the made-up ranks: (3 + r) rays per path in (1 + r / 4) s against rank 0 alone at 3 rays per path in 1 s per step
    assert sd["weak"]["efficiency"] == pytest.approx((3.0 + 4.0) / 1.25 / (2 * 3.0))


def _worker8(rank, world, port):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from robigo_luculenta_amd import distributed as D
    assert D.env_rank() == (rank, rank, world) and D.init_control_plane(rank, world)
    assert D.broadcast_bytes(bytes(range(128)) if rank == 0 else None, 128, root=0) == bytes(range(128))
    ids = D.all_gather_strings("node/0000:%02x:00.0" % (0x05 + 0x10 * rank))
    assert len(ids) == world and len(set(ids)) == world and D.one_gpu_per_rank(ids)
    buf = np.full(3 * 64 * 36, float(rank + 1), np.float32)     # an XYZ buffer per rank: the sum onto rank 0 is 1 + 2 + ... + 8
    assert D.host_staged_reduce(buf, root=0) == (rank == 0)
    if rank == 0:
        assert (buf == sum(range(1, world + 1))).all()
    elapsed, (rays, paths) = D.aggregate(1.0 + 0.1 * rank, [float(rank), 1.0])
    assert elapsed == pytest.approx(1.0 + 0.1 * (world - 1)) and rays == sum(range(world)) and paths == world
    D.shutdown()


def test_eight_rank_control_plane_and_exchange():
    """the SCALE run is eight ranks and no eight-GPU node has been available to any round; everything of it that is
    not the device -- rendezvous, the communicator id's broadcast, one-GPU-per-rank detection, the host-staged sum onto rank 0,
    max / sum aggregation -- with EIGHT gloo ranks."""
    mp.spawn(_worker8, args=(8, _free_port()), nprocs=8, join=True)


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_the_eight_rank_bench_line_assembles_and_names_its_scaling_mode(mode):
    """bench.py --dry-run --gpus 8: the line an 8-GPU SCALE run prints, from made-up counters through the real control plane.  value =
    SUM of the ranks' rays / MAX of their times; scaling_detail carries both modes with efficiency = value / (8 x the same-run one-GPU
    rate); config.rccl.ranks_agree; and `metric` says which mode `value` is."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    extra = ["--total-paths", str(8 * 2 * 64 * 1000)] if mode == "strong" else []
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run", "--gpus", "8", "--steps", "2", "--warmup", "1",
                          "--launches-per-step", "2"] + extra, capture_output=True, timeout=600, env=dict(os.environ, MASTER_PORT=str(_free_port())))
    assert run.returncode == 0, run.stderr.decode()[-2000:]
    line = json.loads([l for l in run.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == mode and ("8 GPUs, %s scaling" % mode.upper()) in line["metric"]
    per_rank_paths = (64 * 1000 if mode == "strong" else 1024 * 524288) * 2 * 2          # paths per launch x launches x steps
    assert line["config"]["paths_per_launch"] * 2 * 2 == per_rank_paths
    assert line["config"]["total_paths_per_step"] == per_rank_paths * 8 // 2
    # rank r: (3 + r) rays per path in 1 + r / 4 seconds
    assert line["value"] == pytest.approx(per_rank_paths * sum(3.0 + r for r in range(8)) / (1.0 + 7 / 4) / 1e6)
    rccl = line["config"]["rccl"]
    assert rccl["world"] == 8 and rccl["ranks_agree"] is True
    sd = line["scaling_detail"]
    other = "strong" if mode == "weak" else "weak"
    assert sd[mode]["headline"] is True and sd[other]["headline"] is False and sd[mode]["value"] == pytest.approx(line["value"])
    for m in ("weak", "strong"):
        assert sd[m]["efficiency"] == pytest.approx(sd[m]["value"] / (8 * sd["n1_same_run"]["value"]))
    assert line["exchange"]["ms_per_step"] == 2.0 + 7 and line["exchange"]["per_step"] == 1.0


def test_app_rank_plan_for_eight_gpus_and_for_shared_ones():
    """rl_app_run's rank bookkeeping (csrc/rl_app.cpp plan_ranks) without a GPU.  devices = [0..7]: every rank leads its
    own device and is its own communicator rank, rank 0 = root; a device listed twice: the second rank is added onto the first
    (leader) and has no communicator of its own; one device: no communicator at all.  Rank r's RNG stream is config.stream + r."""
    sys.path.insert(0, os.path.dirname(HERE))
    import robigo_luculenta_amd as R
    assert R.app_rank_plan(list(range(8))) == (list(range(8)), list(range(8)), list(range(8)), 8)
    assert R.app_rank_plan([4, 5, 6, 7]) == ([4, 5, 6, 7], [0, 1, 2, 3], [0, 1, 2, 3], 4)      # rank 0's device is comm rank 0 whatever its number
    assert R.app_rank_plan([0, 0, 1, 1, 0]) == ([0, 0, 1, 1, 0], [0, 0, 2, 2, 0], [0, -1, 1, -1, -1], 2)
    assert R.app_rank_plan([3, 3]) == ([3, 3], [0, 0], [-1, -1], 0)                             # two streams on one GPU: no RCCL
    assert R.app_rank_plan(None, device=2) == ([2], [0], [-1], 0)
    assert R.app_rank_plan([0, 1, 2, 3, 4, 5, 6, 7, 0]) [2] == [0, 1, 2, 3, 4, 5, 6, 7, -1]

