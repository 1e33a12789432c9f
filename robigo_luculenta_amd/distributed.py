"""The multi-rank plumbing of bench.py.  The hot path shards by SAMPLES: rank g of G renders the
full frame with RNG stream g, and the only exchange is the sum of the ranks' XYZ plot buffers onto rank 0 when a
plot unit is gathered (gather_unit.rs:49-64 then runs on rank 0).

The exchange itself is the library's: rl_plot_unit_reduce / rl_gather_unit_allreduce = one ncclReduce over xGMI on
the plot unit's stream (include/robigo_luculenta.h).  torch.distributed is used here for the CONTROL plane only --
handing rank 0's 128-byte communicator id to the other ranks, barriers, max / sum of the ranks' timings -- over
`gloo`, which needs no GPU and lets tests/test_distributed_gloo.py run these very functions with two CPU ranks.

`host_staged_reduce` is the data plane of `bench.py --dist-backend gloo`: the same sum staged through host memory,
for boxes where the ranks have to share one GPU (RCCL admits one rank per device)."""
import os

import numpy as np

BATCH = 1024 * 512  # trace_unit.rs:67


def env_rank():
    """(rank, local_rank, world) as torch.distributed.run / the driver export them."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_control_plane(rank, world):
    """gloo process group over 127.0.0.1 (or whatever MASTER_ADDR / MASTER_PORT say).  No-op for one rank."""
    if world <= 1:
        return False
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return True


def shutdown():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def broadcast_bytes(payload, n_bytes, root=0):
    """Hands `payload` (bytes on `root`, ignored elsewhere) to every rank: the ncclUniqueId of rl_comm_unique_id."""
    import torch
    import torch.distributed as dist
    buf = torch.zeros(n_bytes, dtype=torch.uint8)
    if dist.get_rank() == root:
        assert len(payload) == n_bytes
        buf.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
    dist.broadcast(buf, src=root)
    return bytes(buf.numpy().tobytes())


def host_staged_reduce(xyz_host, root=0):
    """Sums every rank's plot buffer (a float32 numpy array, modified in place on `root`) over gloo.  Returns True
    on the rank that now holds the sum and must accumulate it; the others only clear (plot_unit.rs:98-102)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return True
    t = torch.from_numpy(xyz_host.reshape(-1))
    dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)
    return dist.get_rank() == root


def aggregate(elapsed_s, counters):
    """bench.py's timing contract: MAX over ranks of the timed region, SUM over ranks of the work counters."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s), [float(c) for c in counters]
    tmax = torch.tensor([float(elapsed_s)], dtype=torch.float64)
    tsum = torch.tensor([float(c) for c in counters], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    return float(tmax[0]), [float(v) for v in tsum]


def make_comm(R, rank, world, device):
    """One RCCL rank of the library (rl_comm_init_rank) for this process; the id travels over the control plane.
    If rank 0 cannot obtain an id (RCCL missing), every rank raises together instead of waiting for a peer."""
    payload, error = bytes(129), None
    if rank == 0:
        try:
            payload = b"\x01" + R.Comm.unique_id()
        except Exception as e:  # noqa: BLE001
            error = e
    payload = broadcast_bytes(payload, 129, root=0)
    if payload[0] != 1:
        raise error if error is not None else RuntimeError("rank 0 could not create an RCCL communicator id")
    return R.Comm(payload[1:], world, rank, device)


def all_gather_strings(text, max_bytes=64):
    """Every rank's `text` (one short string each), in rank order."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [text]
    raw = text.encode()[:max_bytes]
    mine = torch.zeros(max_bytes, dtype=torch.uint8)
    mine[:len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    parts = [torch.zeros(max_bytes, dtype=torch.uint8) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [bytes(p.numpy().tobytes()).rstrip(b"\0").decode() for p in parts]


def pick_device(local_rank, n_visible):
    """The visible device a rank uses: its local rank where the launcher leaves every GPU visible, device
    (local_rank mod n) where it hands every rank a mask of its own (then n is usually 1)."""
    return local_rank if local_rank < n_visible else local_rank % max(n_visible, 1)


def one_gpu_per_rank(bus_ids):
    """RCCL admits one rank per GPU: true iff the ranks' (host, PCI bus id) pairs are all different."""
    return len(set(bus_ids)) == len(bus_ids)
