"""Sample-parallel multi-GPU plan: the hot path shards by SAMPLES.  Rank g of G renders
the full frame with RNG stream g and its own contiguous run of path indices; the only exchange is a
sum of the per-rank XYZ plot buffers at GatherUnit time (gather_unit.rs:49-64 on the root).

These helpers are backend-agnostic torch.distributed code: `nccl` (= RCCL over xGMI) on GPUs,
`gloo` in the CPU tests."""
import torch.distributed as dist

BATCH = 1024 * 512  # trace_unit.rs:67


def shard_batches(total_batches, rank, world_size):
    """Batches [lo, hi) of a job of `total_batches` that `rank` renders (contiguous, sizes differ by
    at most one).  Every rank uses stream = rank, so path indices may overlap between ranks without
    reusing random numbers."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(total_batches, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_plot_buffer(xyz, root=0, group=None):
    """GatherUnit-time exchange: sums every rank's plot buffer (a torch tensor of 3*W*H floats) onto
    `root` in place.  Returns True on the rank that now holds the sum and must accumulate it; the
    other ranks only clear their buffer (PlotUnit::clear, plot_unit.rs:98-102)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(xyz, dst=root, op=dist.ReduceOp.SUM, group=group)
        return dist.get_rank(group) == root
    return True
