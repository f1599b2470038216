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

    import socket

    ids = [None] * dist.get_world_size(group)
    bus = _lib.device_pci_bus_id(device)
    # bus ids repeat from node to node: a GPU is (host, bus id)
    dist.all_gather_object(ids, None if bus is None else (socket.gethostname(), bus), group=group)
    return 'rccl' if all(i is not None for i in ids) and len(set(ids)) == len(ids) else 'host'


def probe_rccl(device, group=None, timeout=150):
    """Does RCCL come up and finish its collectives across the ranks of `group` on this node?  Every rank runs
    sgdml_amd._rccl_probe (a toy sharded solve through the library's RCCL path) in a CHILD process under `timeout`
    seconds -- a hang inside ncclCommInitRank or a collective cannot be interrupted from within the process that
    made the call, a child can be killed -- and the ranks compare the replicated result.  Returns (ok, detail) with
    the same answer on every rank."""
    import json
    import os
    import subprocess
    import sys

    import torch.distributed as dist

    from . import _lib

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    payload = [None]
    if rank == 0:
        try:
            payload = [_lib.Context.comm_unique_id()]
        except Exception as e:
            payload = [repr(e)]
    dist.broadcast_object_list(payload, src=0, group=group)
    mine = {'ok': False}
    if isinstance(payload[0], bytes):
        cmd = [sys.executable, '-m', 'sgdml_amd._rccl_probe', str(device), str(rank), str(world), payload[0].hex()]
        try:
            r = subprocess.run(cmd, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=timeout,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)  # on timeout run() kills exactly this child
            lines = [ln for ln in r.stdout.decode(errors='replace').splitlines() if ln.startswith('{')]
            mine = json.loads(lines[-1]) if r.returncode == 0 and lines else \
                {'ok': False, 'error': 'exit {}: {}'.format(r.returncode, r.stderr.decode(errors='replace')[-300:])}
        except subprocess.TimeoutExpired:
            mine = {'ok': False, 'error': 'no answer within {} s'.format(timeout)}
        except Exception as e:
            mine = {'ok': False, 'error': repr(e)}
    else:
        mine = {'ok': False, 'error': 'unique id: {}'.format(payload[0])}
    every = [None] * world
    dist.all_gather_object(every, mine, group=group)
    ok = all(e.get('ok') for e in every)
    if ok:
        c = [e['checksum'] for e in every]
        ok = max(c) - min(c) <= 1e-9 * max(1.0, max(abs(v) for v in c))
        if not ok:
            return False, {'error': 'replicated solution differs across ranks', 'checksums': c}
        return True, {'collectives': every[0]['collectives'], 'checksum': c[0]}
    return False, {'errors': {r: e.get('error', 'not ok') for r, e in enumerate(every) if not e.get('ok')}}


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
    # the rank as the group knows it: gdml_comm_info answers 0 of 1 while the communicator is parked (gdml_comm_suspend), so
    # anything that must happen on ONE rank (checkpoint writers) is gated on this, not on comm_info()
    ctx._dist_rank, ctx._dist_world = rank, world

    def all_min(value):
        t = torch.tensor([float(value)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return float(t[0])

    ctx._all_min = all_min
    return rank, world


def broadcast_array(arr, src=0, group=None):
    """Broadcast a NumPy array (shape and dtype included) from rank `src`; every rank gets a copy."""
    import torch.distributed as dist

    payload = [arr if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(payload, src=src, group=group)
    return payload[0]
