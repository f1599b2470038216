"""
RCCL probe: `python -m sgdml_amd._rccl_probe <device> <rank> <world> <unique-id hex>`.

One short sharded solve (K_nm rows, Nystroem factor, a few PCG iterations on a 9-atom toy system) through the
library's RCCL communicator, run in a process of its own so that a launcher can find out, under a time limit and
before it commits its real ranks, whether RCCL initialises and completes its all-gather / all-reduce on this
node (bench.py does: the answer decides between `rccl` and the host-staged collectives).  Prints one JSON line
{"ok": true, "checksum": ..., "collectives": ...}; the replicated solution must agree across ranks, which the
caller checks by comparing checksums.  No reference counterpart (the reference is single-process).
"""
import json
import sys

import numpy as np


def run(device, rank, world, unique_id):
    from . import _lib

    n_atoms, n_train, k = 9, 8 * world, 4
    rs = np.random.RandomState(11)
    R = rs.randn(n_train, n_atoms, 3) * 0.3 + np.arange(n_atoms)[None, :, None] * 1.3
    y = rs.randn(n_train * 3 * n_atoms)
    ctx = _lib.Context(device)
    try:
        ctx.comm_init(unique_id, rank, world)
        tp = np.arange(n_atoms * (n_atoms - 1) // 2, dtype=np.int64)[None]
        xd, gd = ctx.desc_from_R(R.reshape(n_train, -1), n_atoms)
        ctx.train_upload(xd, gd, tp)
        ctx.predict_upload_model(xd, np.zeros_like(xd), tp, 10.0, None)
        N3 = 3 * n_atoms
        idx = (np.arange(0, n_train, n_train // k)[:k, None] * N3 + np.arange(N3)[None]).ravel().astype(np.int64)
        ctx.assemble_K(10.0, False, idx=idx, alloc_extra_rows=idx.size)
        ctx.nystroem_factor(1e-8, idx)
        x, info, iters, resid = ctx.pcg(1e-8, False, y, rtol=0.0, maxiter=5)
        calls, nbytes = ctx.comm_stats()
        ok = bool(np.all(np.isfinite(x))) and iters == 5
        return {'ok': ok, 'checksum': float(np.dot(x, np.cos(np.arange(x.size)))), 'resid': float(resid),
                'collectives': int(calls), 'collective_bytes': float(nbytes)}
    finally:
        ctx.close()


if __name__ == '__main__':
    dev, rank, world = (int(v) for v in sys.argv[1:4])
    print(json.dumps(run(dev, rank, world, bytes.fromhex(sys.argv[4]))))
