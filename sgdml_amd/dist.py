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


def pick_backend(device, group=None):
    """'rccl' when every rank of the group sits on its own physical GPU, else 'host'.  Physical identity is the PCI
    bus id of the rank's device, exchanged through the group, so the answer is right both when every rank sees all
    GPUs and picks device LOCAL_RANK and when the launcher hands each rank a private one-device view
    (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank); ranks that share a GPU get the host-staged collectives
    because RCCL refuses duplicate devices."""
    import torch.distributed as dist

    from . import _lib

    ids = [None] * dist.get_world_size(group)
    dist.all_gather_object(ids, _lib.device_pci_bus_id(device), group=group)
    return 'rccl' if all(i is not None for i in ids) and len(set(ids)) == len(ids) else 'host'


def init_comm_from_torch_distributed(ctx, group=None, backend='rccl'):
    """Create the communicator of `ctx` from an initialised torch.distributed process group (any backend,
    it only ships the id / carries the host-staged collectives).  Returns (rank, world).

    backend='rccl': RCCL over xGMI inside the library; rank 0's unique id is broadcast through the group.
    backend='host': the library stages its two collectives through pinned host memory and the group performs
        them (gloo): for ranks that share a GPU (RCCL refuses duplicate devices) and for tests of the sharded
        code path on a one-GPU box."""
    import torch
    import torch.distributed as dist

    from . import _lib

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if backend == 'rccl':
        payload = [_lib.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(payload, src=0, group=group)
        ctx.comm_init(payload[0], rank, world)
    elif backend == 'host':
        def allreduce(buf):
            dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.SUM, group=group)

        def allgather(buf, chunk):
            t = torch.from_numpy(buf)
            mine = t[rank * chunk:(rank + 1) * chunk].clone()
            dist.all_gather(list(t.split(chunk)), mine, group=group)

        ctx.comm_init_host(rank, world, allreduce, allgather)
    else:
        raise ValueError("backend must be 'rccl' or 'host'")
    ctx._bcast = lambda arr, src=0: broadcast_array(arr, src=src, group=group)
    return rank, world


def broadcast_array(arr, src=0, group=None):
    """Broadcast a NumPy array (shape and dtype included) from rank `src`; every rank gets a copy."""
    import torch.distributed as dist

    payload = [arr if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(payload, src=src, group=group)
    return payload[0]
