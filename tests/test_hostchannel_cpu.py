"""CPU tests of sgdml_amd.hostchannel: the PyTorch-free host group of the multi-GPU path (rendezvous, object and buffer
collectives) with three plain subprocesses -- no launcher, no torch import anywhere in the workers."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(world, tmp_path, extra_env=None):
    out = str(tmp_path / 'chan.json')
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   OMP_NUM_THREADS='1')
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', '_channel_worker.py'), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    for p in procs:
        so, se = p.communicate(timeout=120)
        assert p.returncode == 0, se.decode()[-2000:]
    return [json.load(open(out + '.%d' % r)) for r in range(world)]


def test_host_channel_three_ranks(tmp_path):
    world = 3
    res = _run(world, tmp_path)
    for r, x in enumerate(res):
        assert (x['rank'], x['world']) == (r, world) and not x['torch_loaded']
        assert x['gathered'] == '02020202' and x['bcast0'] == 'id-from-0' and x['bcast_last'] == ['from', world - 1]
        assert x['arr'] == list(np.arange(5.0))
        assert (x['max'], x['min']) == (20.0, 0.0)
        assert x['allreduce'] == [6.0, 6.0] and x['allgather'] == [0.5, 1.5, 2.5] and x['big_ok']
    assert [x['shard'] for x in res] == [[0, 4, 4], [4, 8, 4], [8, 10, 4]]


def test_host_channel_skips_a_port_that_is_taken(tmp_path):
    """Rank 0 moves to the next port when the derived one is in use, and the other ranks find it there."""
    from sgdml_amd.hostchannel import channel_port

    blocker = socket.socket()
    blocker.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 0)
    try:
        # occupy the FIRST port the channel would try (an unrelated listener that never answers with the magic word)
        port = _free_port()
        blocker.bind(('127.0.0.1', port))
        blocker.listen(1)
        res = _run(2, tmp_path, extra_env={'GDML_CHANNEL_PORT': str(port)})
        assert [x['allreduce'] for x in res] == [[3.0, 3.0]] * 2
        assert channel_port(29500) != 29500  # never the launcher's own store port
    finally:
        blocker.close()


def test_single_rank_channel_is_trivial():
    from sgdml_amd.hostchannel import HostChannel

    c = HostChannel(rank=0, world=1)
    buf = np.arange(4.0)
    c.allreduce_sum(buf)
    c.allgather(buf, 4)
    assert c.allgather_obj('x') == ['x'] and c.bcast_obj(3) == 3 and c.all_max(2) == 2 and list(buf) == [0, 1, 2, 3]
    c.barrier()
