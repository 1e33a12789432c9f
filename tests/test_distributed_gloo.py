"""The N > 1 path on CPU: two gloo ranks shard the batches, render with stream = rank (here with the
CPU oracle standing in for the device kernels, which need a GPU) and sum-reduce their XYZ plot
buffers to rank 0 exactly as bench.py / the App do over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
W, H, PATHS_PER_BATCH, TOTAL_BATCHES = 48, 27, 4096, 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_buffer(rank, lo, hi):
    sys.path.insert(0, HERE)
    import _oracle as O
    objs, cam = O.demo_scene_desc()
    scene = O.Scene(objs, cam)
    xyz = np.zeros((W * H, 3), np.float32)
    for b in range(lo, hi):
        ph, _ = scene.render(W, H, 1, rank, b * PATHS_PER_BATCH, PATHS_PER_BATCH)
        O.plot(W, H, ph, xyz)
    return xyz


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from robigo_luculenta_amd.distributed import reduce_plot_buffer, shard_batches
    lo, hi = shard_batches(TOTAL_BATCHES, rank, world)
    mine = _rank_buffer(rank, lo, hi)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), mine)
    t = torch.from_numpy(mine.reshape(-1).copy())
    holds_sum = reduce_plot_buffer(t, root=0)
    assert holds_sum == (rank == 0)
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_batches_partition():
    from robigo_luculenta_amd.distributed import shard_batches
    for total in (0, 1, 5, 8, 259200):
        for world in (1, 2, 3, 8):
            spans = [shard_batches(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_batches(4, 2, 2)


def test_two_rank_gloo_reduce_matches_sum_of_shards(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / ("rank%d.npy" % r)) for r in range(world)]
    reduced = np.load(tmp_path / "reduced.npy").reshape(-1, 3)
    assert reduced.tobytes() == (parts[0] + parts[1]).tobytes()   # two operands: float addition is exact-order-free
    assert parts[0].any() and parts[1].any() and not np.array_equal(parts[0], parts[1])
    # disjoint RNG streams: the two ranks' first batches are different samples of the same image
    assert abs(parts[0].sum() / parts[1].sum() - 3 / 2) < 0.25     # rank 0 rendered 3 batches, rank 1 rendered 2
