"""Worker of tests/test_hostchannel_cpu.py: one rank of a HostChannel job (plain subprocesses, no launcher, no torch).
Exercises every operation the multi-GPU path uses and writes what it saw."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    from sgdml_amd.dist import broadcast_array, host_group, shard_range

    hg = host_group()
    r, w = hg.rank, hg.world
    res = {'rank': r, 'world': w, 'torch_loaded': 'torch' in sys.modules}
    res['gathered'] = hg.allgather_obj({'rank': r, 'blob': bytes([r]) * 128})[-1]['blob'].hex()[:8]
    res['bcast0'] = hg.bcast_obj('id-from-0' if r == 0 else None, src=0)
    res['bcast_last'] = hg.bcast_obj(('from', r) if r == w - 1 else None, src=w - 1)
    arr = broadcast_array(np.arange(5.0) * (r + 1), src=0)
    res['arr'] = arr.tolist()
    res['max'], res['min'] = hg.all_max(10.0 * r), hg.all_min(10.0 * r)
    # host-staged collectives on float64 buffers, in place (the contract of gdml_comm_init_host)
    buf = np.full(1000, float(r + 1))
    hg.allreduce_sum(buf)
    res['allreduce'] = [float(buf[0]), float(buf[-1])]
    chunk = 7
    g = np.zeros(chunk * w)
    g[r * chunk:(r + 1) * chunk] = r + 0.5
    hg.allgather(g, chunk)
    res['allgather'] = g[::chunk].tolist()
    big = np.random.RandomState(r).standard_normal(300000)  # 2.4 MB: several TCP segments
    ref = sum(np.random.RandomState(k).standard_normal(300000) for k in range(w))
    hg.allreduce_sum(big)
    res['big_ok'] = bool(np.array_equal(big, ref))
    res['shard'] = list(shard_range(r, w, 10))
    hg.barrier()
    with open(out_path + '.%d' % r, 'w') as f:
        json.dump(res, f)
    hg.close()


if __name__ == '__main__':
    main()
