"""
Multi-GPU plumbing for the sharded iterative solver (one process per GPU).

The collectives themselves (RCCL all-gather / all-reduce over xGMI) are issued inside
libgdml_hip.so on the context's compute stream (csrc/comm.hip); this module only distributes the
RCCL unique id through whatever host channel ``torch.distributed`` already provides (any backend,
including gloo) and mirrors the library's shard arithmetic for callers and tests.
The reference has no distributed code at all.
"""


def shard_range(rank, world, n_points):
    """Contiguous shard of training points owned by `rank`: same rule as csrc/comm.hip::shard_points.
    Returns (first, last_exclusive, points_per_rank)."""
    per = (n_points + world - 1) // world
    a = min(per * rank, n_points)
    b = min(a + per, n_points)
    return a, b, per


def init_comm_from_torch_distributed(ctx, group=None):
    """Create the RCCL communicator of `ctx` using an initialised torch.distributed process group to
    ship rank 0's unique id.  Returns (rank, world)."""
    import torch.distributed as dist

    from . import _lib

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    payload = [_lib.Context.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(payload, src=0, group=group)
    ctx.comm_init(payload[0], rank, world)
    return rank, world
