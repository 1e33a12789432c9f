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
