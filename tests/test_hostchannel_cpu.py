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


def test_message_codec_round_trip_and_refusals():
    """Nothing that arrives is unpickled: the tagged encoding carries what the ranks exchange and nothing else."""
    import pytest

    from sgdml_amd.hostchannel import decode, encode

    obj = {'rank': 3, 'id': bytes(range(128)), 'vals': [1.5, -2, None, True, ('a', 7)], 'big': 1 << 80,
           'arr': np.arange(12.0).reshape(3, 4), 'idx': np.array([[1, 2]], dtype=np.int64), 'empty': np.zeros((0, 3)), 's': 'σ'}
    back = decode(encode(obj))
    assert back['rank'] == 3 and back['id'] == obj['id'] and back['vals'] == obj['vals'] and back['big'] == 1 << 80
    assert back['arr'].dtype == np.float64 and np.array_equal(back['arr'], obj['arr']) and back['arr'].flags.writeable
    assert back['idx'].dtype == np.int64 and back['empty'].shape == (0, 3) and back['s'] == 'σ'
    assert isinstance(back['vals'][4], tuple)
    with pytest.raises(TypeError):
        encode(object())                      # arbitrary objects are not sent ...
    with pytest.raises(TypeError):
        encode(np.array([object()]))
    import pickle
    with pytest.raises(ValueError):
        decode(pickle.dumps({'x': 1}))        # ... and a pickle is not a message
    with pytest.raises(ValueError):
        decode(encode([1, 2]) + b'x')
    with pytest.raises((ValueError, struct_error())):
        decode(encode('abcdef')[:-2])


def struct_error():
    import struct

    return struct.error


def test_handshake_refuses_strangers_and_is_not_held_up_by_them():
    """Rank 0's accept loop against: a connection that says nothing (it may not hold up the real ranks beyond the 2 s
    handshake timeout), the right magic with a bad MAC, a claimed rank out of range, and a rank that holds ANOTHER job's
    token (never admitted, and it refuses rank 0 in turn) -- then the real rank 1 connects and the collectives work."""
    import threading
    import time

    from sgdml_amd import hostchannel as hc

    port = _free_port()
    token, foreign = hc.new_token(), hc.new_token()
    root = {}

    def serve():
        root['c'] = hc.HostChannel(rank=0, world=2, addr='127.0.0.1', port=port, timeout=60, token=token)

    th = threading.Thread(target=serve)
    th.start()
    time.sleep(0.3)
    strangers = [socket.create_connection(('127.0.0.1', port)) for _ in range(3)]
    strangers[1].sendall(hc._MAGIC + (1).to_bytes(4, 'little') + os.urandom(16))   # right magic, cannot answer the challenge
    strangers[2].sendall(hc._MAGIC + (7).to_bytes(4, 'little') + os.urandom(16))   # rank out of range
    try:
        try:
            hc.HostChannel(rank=1, world=2, addr='127.0.0.1', port=port, timeout=1.0, token=foreign)
            raise AssertionError('a rank with a foreign token was admitted')
        except TimeoutError:
            pass
        assert 'c' not in root  # rank 0 is still waiting for ITS rank 1
        t0 = time.time()
        peer = hc.HostChannel(rank=1, world=2, addr='127.0.0.1', port=port, timeout=30, token=token)
        th.join(timeout=30)
        assert 'c' in root and time.time() - t0 < 10.0
        box = {}
        t2 = threading.Thread(target=lambda: box.setdefault('r', root['c'].allgather_obj({'r': 0, 'a': np.arange(3.0)})))
        t2.start()
        mine = peer.allgather_obj({'r': 1, 'a': np.ones(2)})
        t2.join(timeout=30)
        assert [x['r'] for x in mine] == [0, 1] and np.array_equal(box['r'][1]['a'], np.ones(2))
        # a second connection that claims an already registered rank is refused as well (rank 0 no longer listens here; the
        # duplicate check is exercised through the accept loop of a 3-rank root)
        peer.close()
        root['c'].close()
    finally:
        for s_ in strangers:
            s_.close()


def test_duplicate_rank_is_refused():
    import threading
    import time

    from sgdml_amd import hostchannel as hc

    port = _free_port()
    token = hc.new_token()
    root = {}
    th = threading.Thread(target=lambda: root.setdefault('c', hc.HostChannel(rank=0, world=3, addr='127.0.0.1', port=port, timeout=60, token=token)))
    th.start()
    time.sleep(0.3)
    first = hc.HostChannel(rank=1, world=3, addr='127.0.0.1', port=port, timeout=30, token=token)
    try:
        hc.HostChannel(rank=1, world=3, addr='127.0.0.1', port=port, timeout=1.0, token=token)  # same rank again
        raise AssertionError('a duplicate rank was admitted')
    except TimeoutError:
        pass
    second = hc.HostChannel(rank=2, world=3, addr='127.0.0.1', port=port, timeout=30, token=token)
    th.join(timeout=30)
    assert sorted(root['c']._peers) == [1, 2]
    for c in (first, second, root['c']):
        c.close()
