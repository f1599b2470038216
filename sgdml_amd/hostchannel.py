"""
Host-side channel between the ranks of a one-process-per-GPU job -- standard library only.

What the multi-GPU path needs from the host is small: ship the 128-byte RCCL unique id, rank 0's random draws and a few
decisions (solver choice, memory-model k, checkpoint clock) to every rank, a barrier and a max for timings, and -- only for
ranks that SHARE a GPU, where RCCL refuses to run -- the two host-staged collectives of csrc/comm.hip.  The data path
(all-reduce / all-gather of vectors and m x m blocks over xGMI) is RCCL inside the library.  Rounds 1-4 borrowed
torch.distributed's gloo group for this; `north_star` asks for a PyTorch-free stack, and none of it needs more than a few
sockets: rank 0 listens, the other ranks connect (star), every operation is gather-to-root + scatter.

Rendezvous from the launcher's environment (torch.distributed.run, mpirun wrappers, a shell loop -- anything that sets them):
RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT.  The channel does NOT use MASTER_PORT itself (torch.distributed.run keeps its own
store there); it derives a port from it (GDML_CHANNEL_PORT overrides), probes a short sequence of ports on collision and
checks a magic word on connect, so a foreign service on a port is skipped rather than talked to.
The reference has no distributed code at all.
"""
import os
import pickle
import socket
import struct
import time

import numpy as np

_MAGIC = b'GDMLCHN1'
_PORT_TRIES = 16


def _send(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError('host channel: peer closed the connection')
        got += k
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    return _recv_exact(sock, n)


def channel_port(master_port):
    """First port the channel tries for a launcher whose MASTER_PORT is `master_port`."""
    env = os.environ.get('GDML_CHANNEL_PORT')
    if env:
        return int(env)
    return 20000 + (int(master_port) * 7 + 13) % 20000


class HostChannel(object):
    """Star-topology channel: rank 0 holds one socket per peer, every other rank one socket to rank 0."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0):
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        port = channel_port(os.environ.get('MASTER_PORT', '29500')) if port is None else int(port)
        self._peers = {}  # rank 0: rank -> socket
        self._root = None  # other ranks: socket to rank 0
        self._listener = None
        if self.world <= 1:
            return
        if self.rank == 0:
            last = None
            for k in range(_PORT_TRIES):
                ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    ls.bind((addr if addr not in ('localhost',) else '127.0.0.1', port + k))
                    ls.listen(self.world)
                    self._listener = ls
                    break
                except OSError as e:
                    last = e
                    ls.close()
            if self._listener is None:
                raise OSError('host channel: no free port in [{}, {}): {}'.format(port, port + _PORT_TRIES, last))
            self._listener.settimeout(timeout)
            while len(self._peers) < self.world - 1:
                conn, _ = self._listener.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                try:
                    hello = _recv_exact(conn, len(_MAGIC) + 4)
                except (ConnectionError, socket.timeout, OSError):
                    conn.close()
                    continue
                if hello[:len(_MAGIC)] != _MAGIC:
                    conn.close()
                    continue
                (r,) = struct.unpack('<I', hello[len(_MAGIC):])
                conn.sendall(_MAGIC)
                conn.settimeout(None)
                self._peers[r] = conn
        else:
            deadline = time.time() + timeout
            while self._root is None:
                for k in range(_PORT_TRIES):
                    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    s.settimeout(2.0)
                    try:
                        s.connect((addr, port + k))
                        s.sendall(_MAGIC + struct.pack('<I', self.rank))
                        if _recv_exact(s, len(_MAGIC)) == _MAGIC:
                            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            s.settimeout(None)
                            self._root = s
                            break
                    except (OSError, ConnectionError):
                        pass
                    s.close()
                if self._root is None:
                    if time.time() > deadline:
                        raise TimeoutError('host channel: rank 0 not reachable at {}:{}..{}'.format(addr, port, port + _PORT_TRIES - 1))
                    time.sleep(0.05)

    # ------------------------------------------------------------------ object collectives

    def allgather_obj(self, obj):
        """List of every rank's object, in rank order, on every rank."""
        if self.world <= 1:
            return [obj]
        if self.rank == 0:
            every = [obj] + [None] * (self.world - 1)
            for r, s in self._peers.items():
                every[r] = pickle.loads(_recv(s))
            blob = pickle.dumps(every, protocol=4)
            for s in self._peers.values():
                _send(s, blob)
            return every
        _send(self._root, pickle.dumps(obj, protocol=4))
        return pickle.loads(_recv(self._root))

    def bcast_obj(self, obj, src=0):
        if self.world <= 1:
            return obj
        if src != 0:  # rare: route through the root
            return self.allgather_obj(obj if self.rank == src else None)[src]
        if self.rank == 0:
            blob = pickle.dumps(obj, protocol=4)
            for s in self._peers.values():
                _send(s, blob)
            return obj
        return pickle.loads(_recv(self._root))

    def barrier(self):
        self.allgather_obj(None)

    def all_max(self, value):
        return max(self.allgather_obj(float(value)))

    def all_min(self, value):
        return min(self.allgather_obj(float(value)))

    # ------------------------------------------------------------------ host-staged collectives (float64 buffers, in place)

    def allreduce_sum(self, buf):
        """buf <- sum over ranks, summed in rank order on rank 0 (the same bits on every rank)."""
        if self.world <= 1:
            return
        if self.rank == 0:
            acc = np.array(buf, dtype=np.float64, copy=True)
            parts = {r: np.frombuffer(_recv(s), dtype=np.float64) for r, s in self._peers.items()}
            for r in sorted(parts):
                acc += parts[r]
            buf[:] = acc
            blob = acc.tobytes()
            for s in self._peers.values():
                _send(s, blob)
        else:
            _send(self._root, np.ascontiguousarray(buf, dtype=np.float64).tobytes())
            buf[:] = np.frombuffer(_recv(self._root), dtype=np.float64)

    def allgather(self, buf, chunk):
        """buf holds world * chunk doubles; rank r's contribution sits at [r chunk, (r + 1) chunk)."""
        if self.world <= 1:
            return
        mine = slice(self.rank * chunk, (self.rank + 1) * chunk)
        if self.rank == 0:
            for r, s in self._peers.items():
                buf[r * chunk:(r + 1) * chunk] = np.frombuffer(_recv(s), dtype=np.float64)
            blob = np.ascontiguousarray(buf, dtype=np.float64).tobytes()
            for s in self._peers.values():
                _send(s, blob)
        else:
            _send(self._root, np.ascontiguousarray(buf[mine], dtype=np.float64).tobytes())
            buf[:] = np.frombuffer(_recv(self._root), dtype=np.float64)

    def close(self):
        for s in list(self._peers.values()) + [self._root, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._root, self._listener = {}, None, None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
