"""
Multi-GPU plumbing for the sharded solvers (one process per GPU).

The collectives themselves (RCCL all-gather / all-reduce over xGMI) are issued inside libgdml_hip.so on the context's
compute stream (csrc/comm.hip).  What the host has to do around them -- ship the RCCL unique id, rank 0's random draws and a
few decisions to every rank; for ranks that share a GPU, carry the two host-staged collectives -- goes through a HOST GROUP:

  * sgdml_amd.hostchannel.HostChannel: a few TCP sockets, standard library only, rendezvous from RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT.  The default: no PyTorch anywhere in the stack (BASELINE.json north_star).
  * an initialised torch.distributed process group (any backend), for callers that already have one: pass it as `group`.

The reference has no distributed code at all.
"""
import numpy as np


def shard_range(rank, world, n_points):
    """Contiguous shard of training points owned by `rank`: same rule as csrc/comm.hip::shard_points.
    Returns (first, last_exclusive, points_per_rank)."""
    per = (n_points + world - 1) // world
    a = min(per * rank, n_points)
    b = min(a + per, n_points)
    return a, b, per


def sharded_vector_positions(world, n_points, dim_i, use_E_cstr):
    """Position of every entry of a reference-order vector (forces of all points, then -- with energy constraints -- the
    n_points energy entries) inside the replicated device vectors of the sharded solvers: same rule as
    csrc/common.h::VecLayout::pos.  Returns (positions, chunk, n_pad).  Forces only: the identity, padded at the end.  With
    energy constraints the order is rank-major, rank r's chunk = [force entries of its points | their energy entries | padding]
    (a rank's local rows stay one contiguous run).  The order never crosses the C ABI; this restatement serves tests and tools."""
    per = (n_points + world - 1) // world
    n_ff = n_points * dim_i
    if not use_E_cstr:
        return np.arange(n_ff), per * dim_i, per * dim_i * world
    chunk = per * (dim_i + 1)
    pos = np.empty(n_ff + n_points, dtype=np.int64)
    pt = np.arange(n_points)
    r = pt // per
    cnt = np.minimum(per, n_points - r * per)
    pos[:n_ff] = (r[:, None] * chunk + (pt - r * per)[:, None] * dim_i + np.arange(dim_i)[None]).ravel()
    pos[n_ff:] = r * chunk + cnt * dim_i + (pt - r * per)
    return pos, chunk, chunk * world


class _TorchGroup(object):
    """Adapter: a torch.distributed process group with the HostChannel interface (CPU tensors; nothing on the GPU path)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self._dist, self._group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def allgather_obj(self, obj):
        every = [None] * self.world
        self._dist.all_gather_object(every, obj, group=self._group)
        return every

    def bcast_obj(self, obj, src=0):
        payload = [obj if self.rank == src else None]
        self._dist.broadcast_object_list(payload, src=src, group=self._group)
        return payload[0]

    def barrier(self):
        self._dist.barrier(group=self._group)

    def all_max(self, value):
        return max(self.allgather_obj(float(value)))

    def all_min(self, value):
        return min(self.allgather_obj(float(value)))

    def allreduce_sum(self, buf):
        import torch

        self._dist.all_reduce(torch.from_numpy(buf), op=self._dist.ReduceOp.SUM, group=self._group)

    def allgather(self, buf, chunk):
        import torch

        t = torch.from_numpy(buf)
        mine = t[self.rank * chunk:(self.rank + 1) * chunk].clone()
        self._dist.all_gather(list(t.split(chunk)), mine, group=self._group)

    def close(self):
        pass


_default_channel = None


def host_group(group=None):
    """The host group to use: `group` itself when it already is one (HostChannel or anything with its interface), an adapter
    around a torch.distributed group when one is passed (or when the caller has initialised torch.distributed: its default
    group), otherwise the process-wide HostChannel built from the launcher's environment (created on first use; torch is
    never imported by this module on its own)."""
    global _default_channel
    if group is not None:
        if hasattr(group, 'allgather_obj'):
            return group
        return _TorchGroup(group)
    import sys

    tdist = sys.modules.get('torch.distributed')  # a caller who initialised torch.distributed keeps using its default group
    if tdist is not None and tdist.is_available() and tdist.is_initialized():
        return _TorchGroup(None)
    if _default_channel is None:
        from .hostchannel import HostChannel

        _default_channel = HostChannel()
    return _default_channel


def pick_backend(device, group=None):
    """'rccl' when every rank of the group sits on its own physical GPU, else 'host'.  Physical identity is the PCI
    bus id of the rank's device, exchanged through the group, so the answer is right both when every rank sees all
    GPUs and picks device LOCAL_RANK and when the launcher hands each rank a private one-device view
    (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank); ranks that share a GPU get the host-staged collectives
    because RCCL refuses duplicate devices."""
    import socket

    from . import _lib

    hg = host_group(group)
    bus = _lib.device_pci_bus_id(device)
    # bus ids repeat from node to node: a GPU is (host, bus id)
    ids = hg.allgather_obj(None if bus is None else (socket.gethostname(), bus))
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

    from . import _lib

    hg = host_group(group)
    rank, world = hg.rank, hg.world
    payload = None
    if rank == 0:
        try:
            payload = _lib.Context.comm_unique_id()
        except Exception as e:
            payload = repr(e)
    payload = hg.bcast_obj(payload, src=0)
    mine = {'ok': False}
    if isinstance(payload, bytes):
        cmd = [sys.executable, '-m', 'sgdml_amd._rccl_probe', str(device), str(rank), str(world), payload.hex()]
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
        mine = {'ok': False, 'error': 'unique id: {}'.format(payload)}
    every = hg.allgather_obj(mine)
    ok = all(e.get('ok') for e in every)
    if ok:
        c = [e['checksum'] for e in every]
        ok = max(c) - min(c) <= 1e-9 * max(1.0, max(abs(v) for v in c))
        if not ok:
            return False, {'error': 'replicated solution differs across ranks', 'checksums': c}
        return True, {'collectives': every[0]['collectives'], 'checksum': c[0]}
    return False, {'errors': {r: e.get('error', 'not ok') for r, e in enumerate(every) if not e.get('ok')}}


def init_comm(ctx, group=None, backend='rccl'):
    """Create the communicator of `ctx` over the ranks of a host group (see host_group).  Returns (rank, world).

    backend='rccl': RCCL over xGMI inside the library; rank 0's unique id is broadcast through the group.
    backend='host': the library stages its two collectives through pinned host memory and the group performs them: for
        ranks that share a GPU (RCCL refuses duplicate devices) and for tests of the sharded code path on a one-GPU box."""
    from . import _lib

    hg = host_group(group)
    rank, world = hg.rank, hg.world
    if backend == 'rccl':
        uid = hg.bcast_obj(_lib.Context.comm_unique_id() if rank == 0 else None, src=0)
        ctx.comm_init(uid, rank, world)
    elif backend == 'host':
        ctx.comm_init_host(rank, world, hg.allreduce_sum, hg.allgather)
    else:
        raise ValueError("backend must be 'rccl' or 'host'")
    ctx._host_group = hg
    ctx._bcast = lambda arr, src=0: hg.bcast_obj(arr if hg.rank == src else None, src=src)
    # the rank as the group knows it: gdml_comm_info answers 0 of 1 while the communicator is parked (gdml_comm_suspend), so
    # anything that must happen on ONE rank (checkpoint writers) is gated on this, not on comm_info()
    ctx._dist_rank, ctx._dist_world = rank, world
    ctx._all_min = hg.all_min
    return rank, world


def init_comm_from_torch_distributed(ctx, group=None, backend='rccl'):
    """Rounds 1-4 name of init_comm for callers that have initialised torch.distributed: uses `group` (None = the default
    process group) as the host group."""
    import torch.distributed as dist

    if group is None and not dist.is_initialized():
        raise RuntimeError('torch.distributed is not initialised (use sgdml_amd.dist.init_comm for the PyTorch-free channel)')
    return init_comm(ctx, group=_TorchGroup(group), backend=backend)


def broadcast_array(arr, src=0, group=None):
    """Broadcast a NumPy array (shape and dtype included) from rank `src`; every rank gets a copy."""
    hg = host_group(group)
    return hg.bcast_obj(arr if hg.rank == src else None, src=src)
