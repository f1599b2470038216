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
store there); it derives a port from it (GDML_CHANNEL_PORT overrides) and probes a short sequence of ports on collision.

Who may talk on that port (round 6): every connection is authenticated in both directions by an HMAC-SHA256
challenge / response over a per-job token -- GDML_CHANNEL_TOKEN when the launcher sets one (bench.py's own launcher and the
test launchers draw 32 random bytes per job with `new_token`); otherwise a digest of (uid, MASTER_ADDR, MASTER_PORT,
TORCHELASTIC_RUN_ID), which keeps foreign services and other jobs out but is only as secret as those values: set the
token for anything that is not a single trusted node.  A connection that does not complete the handshake within 2 s is
dropped (it cannot hold up the accept loop); a claimed rank outside 1 .. world-1 or already registered is refused.
Nothing received is ever unpickled: messages are a small tagged encoding of None / bool / int / float / str / bytes /
tuple / list / dict / ndarray (`encode` / `decode`).
The reference has no distributed code at all.
"""
import hashlib
import hmac
import os
import socket
import struct
import time

import numpy as np

_MAGIC = b'GDMLCHN2'
_PORT_TRIES = 16
_HANDSHAKE_TIMEOUT = 2.0
_MAX_MSG = 1 << 36


def job_token():
    """Per-job secret of the handshake (bytes)."""
    env = os.environ.get('GDML_CHANNEL_TOKEN')
    if env:
        return env.encode()
    ident = '%d:%s:%s:%s' % (os.getuid(), os.environ.get('MASTER_ADDR', '127.0.0.1'), os.environ.get('MASTER_PORT', '29500'),
                             os.environ.get('TORCHELASTIC_RUN_ID', ''))
    return hashlib.sha256(ident.encode()).digest()


def new_token():
    """A fresh random token for a launcher to put into its children's GDML_CHANNEL_TOKEN."""
    return os.urandom(32).hex()


def _mac(token, *parts):
    return hmac.new(token, b'|'.join(parts), hashlib.sha256).digest()


# ---- tagged encoding: what the ranks exchange is ids, floats, small dicts and arrays


def _enc(obj, out):
    if obj is None:
        out.append(b'N')
    elif isinstance(obj, (bool, np.bool_)):
        out.append(b'T' if obj else b'F')
    elif isinstance(obj, (int, np.integer)):
        b = str(int(obj)).encode()
        out.append(b'i' + struct.pack('<I', len(b)) + b)
    elif isinstance(obj, (float, np.floating)):
        out.append(b'd' + struct.pack('<d', float(obj)))
    elif isinstance(obj, str):
        b = obj.encode('utf-8')
        out.append(b's' + struct.pack('<Q', len(b)) + b)
    elif isinstance(obj, (bytes, bytearray, memoryview)):
        b = bytes(obj)
        out.append(b'b' + struct.pack('<Q', len(b)) + b)
    elif isinstance(obj, (tuple, list)):
        out.append((b't' if isinstance(obj, tuple) else b'l') + struct.pack('<Q', len(obj)))
        for x in obj:
            _enc(x, out)
    elif isinstance(obj, dict):
        out.append(b'm' + struct.pack('<Q', len(obj)))
        for k, v in obj.items():
            _enc(k, out)
            _enc(v, out)
    elif isinstance(obj, np.ndarray):
        if obj.dtype.hasobject:
            raise TypeError('host channel: object arrays are not sent')
        a = np.ascontiguousarray(obj)
        dt = a.dtype.str.encode()
        out.append(b'a' + struct.pack('<B', len(dt)) + dt + struct.pack('<B', a.ndim) + struct.pack('<%dq' % a.ndim, *a.shape))
        out.append(a.tobytes())
    else:
        raise TypeError('host channel: cannot send a %s' % type(obj).__name__)


def encode(obj):
    out = []
    _enc(obj, out)
    return b''.join(out)


def _dec(buf, pos):
    tag = bytes(buf[pos:pos + 1])
    pos += 1
    if tag == b'N':
        return None, pos
    if tag in (b'T', b'F'):
        return tag == b'T', pos
    if tag == b'i':
        (n,) = struct.unpack_from('<I', buf, pos)
        return int(bytes(buf[pos + 4:pos + 4 + n]).decode()), pos + 4 + n
    if tag == b'd':
        return struct.unpack_from('<d', buf, pos)[0], pos + 8
    if tag in (b's', b'b'):
        (n,) = struct.unpack_from('<Q', buf, pos)
        raw = bytes(buf[pos + 8:pos + 8 + n])
        if len(raw) != n:
            raise ValueError('host channel: truncated message')
        return (raw.decode('utf-8') if tag == b's' else raw), pos + 8 + n
    if tag in (b't', b'l'):
        (n,) = struct.unpack_from('<Q', buf, pos)
        pos += 8
        if n > len(buf):
            raise ValueError('host channel: bad sequence length')
        items = []
        for _ in range(n):
            x, pos = _dec(buf, pos)
            items.append(x)
        return (tuple(items) if tag == b't' else items), pos
    if tag == b'm':
        (n,) = struct.unpack_from('<Q', buf, pos)
        pos += 8
        if n > len(buf):
            raise ValueError('host channel: bad mapping length')
        d = {}
        for _ in range(n):
            k, pos = _dec(buf, pos)
            v, pos = _dec(buf, pos)
            d[k] = v
        return d, pos
    if tag == b'a':
        ln = buf[pos]
        dt = np.dtype(bytes(buf[pos + 1:pos + 1 + ln]).decode())
        if dt.hasobject:
            raise ValueError('host channel: object arrays are not accepted')
        pos += 1 + ln
        nd = buf[pos]
        shape = struct.unpack_from('<%dq' % nd, buf, pos + 1)
        pos += 1 + 8 * nd
        if any(x < 0 for x in shape):
            raise ValueError('host channel: bad array shape')
        nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        raw = bytes(buf[pos:pos + nbytes])
        if len(raw) != nbytes:
            raise ValueError('host channel: truncated array')
        return np.frombuffer(raw, dtype=dt).reshape(shape).copy(), pos + nbytes
    raise ValueError('host channel: unknown tag %r' % tag)


def decode(blob):
    obj, pos = _dec(memoryview(blob), 0)
    if pos != len(blob):
        raise ValueError('host channel: trailing bytes in a message')
    return obj


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
    if n > _MAX_MSG:
        raise ConnectionError('host channel: message length %d refused' % n)
    return _recv_exact(sock, n)


def channel_port(master_port):
    """First port the channel tries for a launcher whose MASTER_PORT is `master_port`."""
    env = os.environ.get('GDML_CHANNEL_PORT')
    if env:
        return int(env)
    return 20000 + (int(master_port) * 7 + 13) % 20000


class HostChannel(object):
    """Star-topology channel: rank 0 holds one socket per peer, every other rank one socket to rank 0."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0, token=None):
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        port = channel_port(os.environ.get('MASTER_PORT', '29500')) if port is None else int(port)
        self._peers = {}  # rank 0: rank -> socket
        self._root = None  # other ranks: socket to rank 0
        self._listener = None
        if self.world <= 1:
            return
        token = job_token() if token is None else (token.encode() if isinstance(token, str) else bytes(token))
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
            deadline = time.time() + timeout
            while len(self._peers) < self.world - 1:
                left = deadline - time.time()
                if left <= 0:
                    raise TimeoutError('host channel: %d of %d ranks connected within %.0f s'
                                       % (len(self._peers) + 1, self.world, timeout))
                self._listener.settimeout(left)
                try:
                    conn, _ = self._listener.accept()
                except socket.timeout:
                    continue
                # handshake under a SHORT timeout:
                #   peer: magic, claimed rank, nonce_p   ->   root: nonce_r, MAC(root | rank | nonce_p)   ->   peer: MAC(peer | rank | nonce_r)
                conn.settimeout(_HANDSHAKE_TIMEOUT)
                try:
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    hello = _recv_exact(conn, len(_MAGIC) + 4 + 16)
                    if hello[:len(_MAGIC)] != _MAGIC:
                        raise ConnectionError('magic')
                    (r,) = struct.unpack('<I', hello[len(_MAGIC):len(_MAGIC) + 4])
                    peer_nonce = hello[len(_MAGIC) + 4:]
                    if not (1 <= r < self.world) or r in self._peers:
                        raise ConnectionError('rank')
                    nonce = os.urandom(16)
                    rb = struct.pack('<I', r)
                    conn.sendall(nonce + _mac(token, b'root', rb, peer_nonce))
                    reply = _recv_exact(conn, 32)
                    if not hmac.compare_digest(reply, _mac(token, b'peer', rb, nonce)):
                        raise ConnectionError('token')
                    conn.sendall(_MAGIC)
                except (ConnectionError, socket.timeout, OSError, struct.error):
                    conn.close()
                    continue
                conn.settimeout(None)
                self._peers[r] = conn
        else:
            deadline = time.time() + timeout
            while self._root is None:
                for k in range(_PORT_TRIES):
                    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    s.settimeout(_HANDSHAKE_TIMEOUT)
                    try:
                        s.connect((addr, port + k))
                        nonce = os.urandom(16)
                        rb = struct.pack('<I', self.rank)
                        s.sendall(_MAGIC + rb + nonce)
                        answer = _recv_exact(s, 16 + 32)
                        if hmac.compare_digest(answer[16:], _mac(token, b'root', rb, nonce)):  # it IS this job's rank 0
                            s.sendall(_mac(token, b'peer', rb, answer[:16]))
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
                every[r] = decode(_recv(s))
            blob = encode(every)
            for s in self._peers.values():
                _send(s, blob)
            return every
        _send(self._root, encode(obj))
        return decode(_recv(self._root))

    def bcast_obj(self, obj, src=0):
        if self.world <= 1:
            return obj
        if src != 0:  # rare: route through the root
            return self.allgather_obj(obj if self.rank == src else None)[src]
        if self.rank == 0:
            blob = encode(obj)
            for s in self._peers.values():
                _send(s, blob)
            return obj
        return decode(_recv(self._root))

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
