"""ctypes binding of libgdml_hip.so (include/gdml_hip.h).  No compute happens in Python."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GDML_HIP_LIB', os.path.join(_HERE, 'libgdml_hip.so'))  # override: A/B builds

GDML_OK = 0
ERR_INVALID, ERR_HIP, ERR_OOM, ERR_STATE, ERR_NOT_PD, ERR_UNSUPPORTED, ERR_COMM = -1, -2, -3, -4, -5, -6, -7
COLS_ALL, COLS_POINTS, COLS_INDEX = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_vp = C.c_void_p
PCG_CB = C.CFUNCTYPE(C.c_int, C.c_int64, C.c_double, _vp)
HOST_COLL_CB = C.CFUNCTYPE(C.c_int, _dp, C.c_int64, _vp)  # gdml_host_allreduce / gdml_host_allgather

# name -> (restype, argtypes).  Must list every symbol declared in include/gdml_hip.h.
SIGNATURES = {
    'gdml_abi_version': (C.c_int, []),
    'gdml_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'gdml_device_pci_bus_id': (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    'gdml_ctx_create': (C.c_int, [C.c_int, C.POINTER(_vp)]),
    'gdml_ctx_destroy': (C.c_int, [_vp]),
    'gdml_last_error': (C.c_char_p, [_vp]),
    'gdml_sync': (C.c_int, [_vp]),
    'gdml_mem_info': (C.c_int, [_vp, _ip, _ip, _ip]),
    'gdml_mem_reserve': (C.c_int, [_vp, C.c_int64, _ip]),
    'gdml_phase_ms': (C.c_int, [_vp, C.c_char_p, _dp, _ip]),
    'gdml_profile': (C.c_int, [_vp, C.c_int]),
    'gdml_kernel_stat': (C.c_int, [_vp, C.c_char_p, _dp, _ip, _dp]),
    'gdml_desc_from_R': (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp, _vp]),
    'gdml_sym_eig_absv': (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp]),
    'gdml_perm_match': (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp, C.c_int64, C.POINTER(C.c_int64)]),
    'gdml_train_upload': (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int, _vp, C.c_int]),
    'gdml_assemble_K': (C.c_int, [_vp, C.c_double, C.c_int, C.c_int, C.c_int64, C.c_int64, _vp, C.c_int64,
                                  C.c_int64, _vp, C.c_int64]),
    'gdml_assemble_A': (C.c_int, [_vp, C.c_double, C.c_double, C.c_int, C.c_int64]),
    'gdml_K_shape': (C.c_int, [_vp, _ip, _ip, _ip]),
    'gdml_chol_set_rhs': (C.c_int, [_vp, _vp, C.c_int64]),
    'gdml_chol_factor': (C.c_int, [_vp, C.c_double, C.POINTER(C.c_int)]),
    'gdml_lu_solve': (C.c_int, [_vp, C.c_double, _vp, C.c_int64, _vp, C.POINTER(C.c_int)]),
    'gdml_chol_solve': (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp]),
    'gdml_predict_upload_model': (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int, _vp, C.c_int, C.c_double, _vp]),
    'gdml_set_alphas': (C.c_int, [_vp, _vp, _vp]),
    'gdml_predict': (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, _vp, _vp]),
    'gdml_predict_dev': (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, _vp, _vp]),
    'gdml_kernel_matvec': (C.c_int, [_vp, C.c_double, C.c_int, _vp, C.c_int64, _vp]),
    'gdml_predict_errors': (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, C.c_double, C.c_double, _vp, _vp, _vp]),
    'gdml_nystroem_factor': (C.c_int, [_vp, C.c_double, _vp, C.c_int64, _vp, _vp, C.POINTER(C.c_int)]),
    'gdml_nystroem_lev_scores': (C.c_int, [_vp, _vp]),
    'gdml_precon_apply': (C.c_int, [_vp, C.c_double, _vp, C.c_int64, _vp]),
    'gdml_pcg': (C.c_int, [_vp, C.c_double, C.c_int, _vp, _vp, C.c_int64, C.c_double, C.c_int64, C.c_int,
                           PCG_CB, C.c_int64, _vp, _vp, _ip, _dp, C.POINTER(C.c_int)]),
    'gdml_pcg_x': (C.c_int, [_vp, _vp]),
    'gdml_dist_chol_solve': (C.c_int, [_vp, C.c_double, C.c_double, _vp, C.c_int64, _vp, C.POINTER(C.c_int)]),
    'gdml_comm_unique_id': (C.c_int, [_vp]),
    'gdml_comm_init': (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    'gdml_comm_info': (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'gdml_comm_suspend': (C.c_int, [_vp, C.c_int]),
    'gdml_comm_init_host': (C.c_int, [_vp, C.c_int, C.c_int, HOST_COLL_CB, HOST_COLL_CB, _vp]),
    'gdml_comm_stats': (C.c_int, [_vp, _ip, _dp]),
    'gdml_set_option': (C.c_int, [_vp, C.c_char_p, C.c_double]),
    'gdml_get_option': (C.c_int, [_vp, C.c_char_p, _dp, C.POINTER(C.c_int)]),
    'gdml_dev_alloc': (C.c_int, [_vp, C.c_int64, C.POINTER(_vp)]),
    'gdml_dev_free': (C.c_int, [_vp, _vp]),
    'gdml_memcpy_h2d': (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    'gdml_memcpy_d2h': (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    'gdml_K_dev': (C.c_int, [_vp, C.POINTER(_vp), _ip]),
}

_lib = None


class GDMLHipError(RuntimeError):
    pass


def load():
    """Load libgdml_hip.so (built by __graft_entry__.build() / make -C sgdml_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libgdml_hip.so not found at {} -- build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` or `make -C sgdml_amd/csrc`. There is no CPU fallback.'.format(LIB_PATH)
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # raises AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def device_count():
    """Number of visible HIP devices (0 without a GPU)."""
    n = C.c_int(0)
    load().gdml_device_count(C.byref(n))
    return n.value


def device_pci_bus_id(device):
    """PCI bus id of visible device `device` (its physical identity), or None if the runtime cannot tell."""
    buf = C.create_string_buffer(64)
    if load().gdml_device_pci_bus_id(int(device), buf, 64) != 0:
        return None
    return buf.value.decode() or None


def preflight(attempts=3, timeout=180):
    """First touch of the GPU in a CHILD process: create a context and run one tiny kernel.  On a freshly leased box the
    very first kernel launch of a process has aborted inside the HIP runtime twice in ~40 runs (no message, before any of
    our code ran on the device); an abort cannot be caught, so callers that must not die with it (the test session, the
    benchmark, smoke) let a child take that first touch, and retry it.  Returns the number of attempts used (0: no GPU
    visible, nothing to do); raises RuntimeError when every attempt failed."""
    import subprocess
    import sys

    if device_count() < 1:
        return 0
    code = ('import numpy as np; from sgdml_amd import _lib; c = _lib.Context(0); '
            'c.desc_from_R(np.arange(18.0).reshape(2, 9) ** 1.5, 3); c.close()')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    err = ''
    for k in range(1, attempts + 1):
        try:
            p = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True, timeout=timeout)
            if p.returncode == 0:
                return k
            err = (p.stderr or '')[-400:]
        except subprocess.TimeoutExpired:
            err = 'timeout'
    raise RuntimeError('GPU preflight failed %d times: %s' % (attempts, err))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def i64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int64)


def _digest(a):
    """Content fingerprint of a contiguous array (xxhash when present, else md5)."""
    try:
        import xxhash

        return xxhash.xxh64(memoryview(a).cast('B')).hexdigest()
    except ImportError:  # pragma: no cover
        import hashlib

        return hashlib.md5(memoryview(a).cast('B')).hexdigest()


def tril_perms_from_lin(tril_perms_lin, dim_d):
    """Undo the linearisation of sgdml/train.py:903-904: returns the (P,D) int64 table."""
    lin = np.asarray(tril_perms_lin, dtype=np.int64).ravel()
    n_perms = lin.size // dim_d
    return np.ascontiguousarray(lin.reshape(dim_d, n_perms).T - np.arange(n_perms)[:, None] * dim_d)


class Context(object):
    """One gdml_ctx (one GPU, own streams and device buffers)."""

    max_query_batch = 1 << 17  # queries per gdml_predict call (bounds the device work buffers)

    def __init__(self, device=None):
        self._h = None
        lib = load()
        if device is None:
            device = int(os.environ.get('LOCAL_RANK', '0'))
            n = C.c_int(0)
            lib.gdml_device_count(C.byref(n))
            if n.value > 0:
                device %= n.value
        h = _vp()
        rc = lib.gdml_ctx_create(int(device), C.byref(h))
        if rc != GDML_OK:
            msg = lib.gdml_last_error(None).decode()
            raise GDMLHipError('cannot create GPU context (device {}): {}'.format(device, msg))
        self._lib = lib
        self._h = h
        self.device = device

    def close(self):
        if self._h is not None:
            self._lib.gdml_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- error mapping (SURVEY.md section 8b "Error conventions")
    def _check(self, rc):
        if rc == GDML_OK:
            return
        msg = self._lib.gdml_last_error(self._h).decode()
        cause = getattr(self, '_coll_exc', None)
        if cause is not None:  # a host-collective callback failed: that exception is the real reason of this error code
            self._coll_exc = None
            raise GDMLHipError('[{}] {}'.format(rc, msg)) from cause
        if rc == ERR_INVALID:
            raise ValueError(msg)
        if rc == ERR_OOM:
            raise MemoryError(msg)
        if rc == ERR_NOT_PD:
            raise np.linalg.LinAlgError(msg)
        if rc == ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        raise GDMLHipError('[{}] {}'.format(rc, msg))

    def sync(self):
        self._check(self._lib.gdml_sync(self._h))

    # -- multi-GPU (one process per GPU, RCCL bound inside the library)
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL unique id (call on one rank, ship to the others by any host channel)."""
        lib = load()
        buf = C.create_string_buffer(128)
        rc = lib.gdml_comm_unique_id(buf)
        if rc != GDML_OK:
            raise GDMLHipError('gdml_comm_unique_id failed: ' + lib.gdml_last_error(None).decode())
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        """unique_id=None creates a 'virtual rank' (shard arithmetic, no collectives; tests only)."""
        self._check(self._lib.gdml_comm_init(self._h, unique_id, int(rank), int(world)))

    def comm_init_host(self, rank, world, allreduce, allgather):
        """Host-staged collectives (gdml_comm_init_host).  allreduce(buf) / allgather(buf, chunk) receive a
        float64 NumPy view of the pinned staging buffer and must complete the collective in place."""
        def _ar(p, count, _user):
            try:
                allreduce(np.ctypeslib.as_array(p, shape=(count,)))
                return 0
            except BaseException as e:  # never let an exception cross the C frame
                self._coll_exc = e
                return 1

        def _ag(p, chunk, _user):
            try:
                allgather(np.ctypeslib.as_array(p, shape=(chunk * world,)), int(chunk))
                return 0
            except BaseException as e:
                self._coll_exc = e
                return 1

        self._coll_cbs = (HOST_COLL_CB(_ar), HOST_COLL_CB(_ag))  # keep the thunks alive
        self._check(self._lib.gdml_comm_init_host(self._h, int(rank), int(world), self._coll_cbs[0],
                                                  self._coll_cbs[1], None))

    def comm_suspended(self):
        """Context manager: inside it this context behaves like a single GPU without a communicator
        (gdml_comm_suspend) -- for solves every rank performs redundantly."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            self._check(self._lib.gdml_comm_suspend(self._h, 1))
            try:
                yield self
            finally:
                self._check(self._lib.gdml_comm_suspend(self._h, 0))

        return _cm()

    def comm_stats(self):
        """(collectives issued, payload bytes handed to them by this rank)."""
        n, b = C.c_int64(), C.c_double()
        self._check(self._lib.gdml_comm_stats(self._h, C.byref(n), C.byref(b)))
        return n.value, b.value

    def set_option(self, key, value):
        """Tuning / ablation option of this context (keys: include/gdml_hip.h)."""
        self._check(self._lib.gdml_set_option(self._h, key.encode(), float(value)))

    def get_option(self, key):
        """Value of an option, or None when it was never set (built-in default applies)."""
        v, is_set = C.c_double(), C.c_int()
        self._check(self._lib.gdml_get_option(self._h, key.encode(), C.byref(v), C.byref(is_set)))
        return v.value if is_set.value else None

    def comm_info(self):
        r, w = C.c_int(), C.c_int()
        self._check(self._lib.gdml_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def mem_info(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.gdml_mem_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def mem_reserve(self, n_bytes=None):
        """Reserve the process-level device arena (gdml_mem_reserve): one block the large matrices of every context on
        this device are carved from, kept until the process ends.  n_bytes=None: 85 % of the memory that is free right
        now; 0 releases it.  Returns the bytes held."""
        if n_bytes is None:
            n_bytes = int(0.85 * self.mem_info()[1]) // 2**30 * 2**30
        got = C.c_int64()
        self._check(self._lib.gdml_mem_reserve(self._h, int(n_bytes), C.byref(got)))
        return got.value

    def phase_ms(self, name):
        ms, n = C.c_double(), C.c_int64()
        self._check(self._lib.gdml_phase_ms(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile(self, enable=True):
        self._check(self._lib.gdml_profile(self._h, int(bool(enable))))

    def kernel_stat(self, name):
        """(summed ms, launches, summed algorithmic work) of a hot kernel since profile(True)."""
        ms, n, w = C.c_double(), C.c_int64(), C.c_double()
        self._check(self._lib.gdml_kernel_stat(self._h, name.encode(), C.byref(ms), C.byref(n), C.byref(w)))
        return ms.value, n.value, w.value

    def desc_from_R(self, R, n_atoms, lat_and_inv=None):
        R = f64(R).reshape(-1, 3 * n_atoms)
        M = R.shape[0]
        D = n_atoms * (n_atoms - 1) // 2
        xd = np.empty((M, D))
        gd = np.empty((M, D, 3))
        lat = lat_inv = None
        if lat_and_inv is not None:
            lat, lat_inv = f64(lat_and_inv[0]), f64(lat_and_inv[1])
        self._check(self._lib.gdml_desc_from_R(self._h, _ptr(R), M, n_atoms, _ptr(lat), _ptr(lat_inv),
                                               _ptr(xd), _ptr(gd)))
        return xd, gd

    def sym_eig_absv(self, adj):
        """|eigenvectors| (M,N,N) of symmetric matrices, columns by decreasing eigenvalue (batched Jacobi on the device)."""
        adj = f64(adj)
        M, N = adj.shape[0], adj.shape[1]
        out = np.empty((M, N, N))
        self._check(self._lib.gdml_sym_eig_absv(self._h, _ptr(adj), M, N, _ptr(out)))
        return out

    def perm_match(self, absv, adj, species):
        """Pairwise atom matching of the symmetry search on the device (gdml_perm_match).  absv = None: the eigenvectors of the
        distance matrices are computed on the device too (batched Jacobi).  Returns (cost (M,M) with the entries i < j set,
        ij (n,2), perms (n,N)): the kept assignments."""
        adj = f64(adj)
        M, N = adj.shape[0], adj.shape[1]
        if adj.shape != (M, N, N):
            raise ValueError('adj must be (M,N,N)')
        if absv is not None:
            absv = f64(absv)
            if absv.shape != (M, N, N):
                raise ValueError('absv must be (M,N,N)')
        sp = np.ascontiguousarray(species, dtype=np.int32)
        cost = np.zeros((M, M))
        cap = max(1, min(M * (M - 1) // 2, 4 * M))  # a tree's worth of kept pairs and then some; repeated with full room if short
        while True:
            ij = np.empty((cap, 2), dtype=np.int32)
            pm = np.empty((cap, N), dtype=np.int32)
            n = C.c_int64(0)
            self._check(self._lib.gdml_perm_match(self._h, _ptr(absv), _ptr(adj), _ptr(sp), M, N, _ptr(cost), _ptr(ij),
                                                  _ptr(pm), cap, C.byref(n)))
            if n.value <= cap:
                return cost, ij[:n.value], pm[:n.value]
            cap = M * (M - 1) // 2

    def train_upload(self, R_desc, R_d_desc, tril_perms):
        R_desc, R_d_desc, tril_perms = f64(R_desc), f64(R_d_desc), i64(tril_perms)
        M, D = R_desc.shape
        n_atoms = int((1 + np.sqrt(8 * D + 1)) / 2)
        if R_d_desc.shape != (M, D, 3):
            raise ValueError('R_d_desc must be the compressed (M,D,3) Jacobian')
        # the training set (descriptors, Jacobians, permutations and the dense tables derived from them on
        # the device) is sigma-independent: a hyper-parameter sweep over the same points uploads it once
        fp = (R_desc.shape, tril_perms.shape, _digest(R_desc), _digest(R_d_desc), _digest(tril_perms))
        if getattr(self, '_train_fp', None) == fp:
            return
        self._train_fp = None
        self._check(self._lib.gdml_train_upload(self._h, _ptr(R_desc), _ptr(R_d_desc), M, n_atoms,
                                                _ptr(tril_perms), tril_perms.shape[0]))
        self._train_fp = fp
        self.n_train, self.n_atoms = M, n_atoms

    def assemble_K(self, sig, use_E_cstr=False, points=None, idx=None, alloc_extra_rows=0, to_host=False,
                   for_cholesky=None):
        """Returns the host copy (rows+extra, cols) if to_host else None (matrix stays on the GPU).
        for_cholesky=lam: all columns, device only, assembled as A = -K + lam I for chol_factor (gdml_assemble_A)."""
        self._last_use_E = bool(use_E_cstr)
        if for_cholesky is not None:
            if points is not None or idx is not None or to_host:
                raise ValueError('for_cholesky assembles the full device-resident system matrix')
            self._check(self._lib.gdml_assemble_A(self._h, float(sig), float(for_cholesky), int(bool(use_E_cstr)),
                                                  int(alloc_extra_rows)))
            return None
        kind, a, b, ip, n_idx = COLS_ALL, 0, 0, None, 0
        if points is not None:
            kind, (a, b) = COLS_POINTS, points
        elif idx is not None:
            idx = i64(idx)
            kind, ip, n_idx = COLS_INDEX, _ptr(idx), idx.size
        if not to_host:
            self._check(self._lib.gdml_assemble_K(self._h, float(sig), int(bool(use_E_cstr)), kind, a, b, ip,
                                                  n_idx, int(alloc_extra_rows), None, 0))
            return None
        # two-step: the shape is only known to the library -> query after a device-only assembly
        n_rows = self.n_train * 3 * self.n_atoms + (self.n_train if use_E_cstr else 0)
        if kind == COLS_ALL:
            n_cols = n_rows
        elif kind == COLS_POINTS:
            n_cols = (min(b, self.n_train) - min(a, self.n_train)) * 3 * self.n_atoms + \
                     (max(b, self.n_train) - max(a, self.n_train))
        else:
            n_cols = n_idx
        K = np.empty((n_rows + alloc_extra_rows, n_cols))
        self._check(self._lib.gdml_assemble_K(self._h, float(sig), int(bool(use_E_cstr)), kind, a, b, ip,
                                              n_idx, int(alloc_extra_rows), _ptr(K), n_cols))
        return K

    def resident_K_bytes(self):
        """Bytes of the kernel-matrix buffer the context keeps for reuse (0 if none)."""
        try:
            rows, cols, extra = self.K_shape()
        except GDMLHipError:
            return 0
        ld = (cols + 15) // 16 * 16
        return (rows + extra) * ld * 8

    def K_shape(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.gdml_K_shape(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def K_to_host(self):
        """Copy of the resident matrix / factor (tests only)."""
        rows, cols, extra = self.K_shape()
        p, ld = _vp(), C.c_int64()
        self._check(self._lib.gdml_K_dev(self._h, C.byref(p), C.byref(ld)))
        buf = np.empty((rows + extra, ld.value))
        self._check(self._lib.gdml_memcpy_d2h(self._h, _ptr(buf), p, buf.nbytes))
        return buf[:, :cols]

    def chol_factor(self, lam):
        info = C.c_int(0)
        self._check(self._lib.gdml_chol_factor(self._h, float(lam), C.byref(info)))
        return info.value

    def lu_solve(self, lam, y):
        """alphas = -(A^-1 y), A = -K + lam I, by LU with partial pivoting (the full K of assemble_K must be
        resident and is consumed).  Raises numpy.linalg.LinAlgError for an exactly singular matrix."""
        y = f64(y).ravel()
        out = np.empty_like(y)
        info = C.c_int(0)
        self._check(self._lib.gdml_lu_solve(self._h, float(lam), _ptr(y), y.size, _ptr(out), C.byref(info)))
        return out

    def dist_chol_solve(self, sig, lam, y):
        """Distributed analytic solve over the ranks of this context's communicator (gdml_dist_chol_solve):
        alphas = -(A^-1 y), A = -K + lam I, on every rank."""
        y = f64(y).ravel()
        out = np.empty_like(y)
        info = C.c_int(0)
        self._check(self._lib.gdml_dist_chol_solve(self._h, float(sig), float(lam), _ptr(y), y.size, _ptr(out),
                                                   C.byref(info)))
        return out

    def chol_set_rhs(self, y):
        """Hands y over before chol_factor (K must have been assembled with alloc_extra_rows >= 1): the
        factorisation then performs the forward substitution and chol_solve(None) only the backward one."""
        y = f64(y).ravel()
        self._check(self._lib.gdml_chol_set_rhs(self._h, _ptr(y), y.size))

    def chol_solve(self, y=None, n_refine=0):
        if y is None:
            n = self.K_shape()[0]
            out = np.empty(n)
            self._check(self._lib.gdml_chol_solve(self._h, None, n, int(n_refine), _ptr(out)))
            return out
        y = f64(y).ravel()
        out = np.empty_like(y)
        self._check(self._lib.gdml_chol_solve(self._h, _ptr(y), y.size, int(n_refine), _ptr(out)))
        return out

    def predict_upload_model(self, R_desc, R_d_desc_alpha, tril_perms, sig, alphas_E=None):
        R_desc, R_d_desc_alpha, tril_perms = f64(R_desc), f64(R_d_desc_alpha), i64(tril_perms)
        alphas_E = f64(alphas_E)
        M, D = R_desc.shape
        n_atoms = int((1 + np.sqrt(8 * D + 1)) / 2)
        self._check(self._lib.gdml_predict_upload_model(self._h, _ptr(R_desc), _ptr(R_d_desc_alpha), M, n_atoms,
                                                        _ptr(tril_perms), tril_perms.shape[0], float(sig),
                                                        _ptr(alphas_E)))
        self.model_n_train, self.model_n_atoms = M, n_atoms

    def set_alphas(self, alphas_F, alphas_E=None):
        alphas_F, alphas_E = f64(alphas_F).ravel(), f64(alphas_E)
        self._check(self._lib.gdml_set_alphas(self._h, _ptr(alphas_F), _ptr(alphas_E)))

    def predict(self, R=None, lat_and_inv=None, return_E=True):
        n_atoms = self.model_n_atoms
        if R is None:
            B = self.n_train
        else:
            R = f64(R).reshape(-1, 3 * n_atoms)
            B = R.shape[0]
        E = np.empty(B) if return_E else None
        F = np.empty((B, 3 * n_atoms))
        lat = lat_inv = None
        if lat_and_inv is not None:
            lat, lat_inv = f64(lat_and_inv[0]), f64(lat_and_inv[1])
        if R is None or B <= self.max_query_batch:
            self._check(self._lib.gdml_predict(self._h, _ptr(R), B, _ptr(lat), _ptr(lat_inv), _ptr(E), _ptr(F)))
            return E, F
        # very large batches go through in slices so that the device work buffers stay bounded (the reference's
        # GPU path re-batches instead of failing as well, torchtools.py:349-387)
        for b0 in range(0, B, self.max_query_batch):
            b1 = min(B, b0 + self.max_query_batch)
            Ec = None if E is None else E[b0:b1]
            self._check(self._lib.gdml_predict(self._h, _ptr(R[b0:b1]), b1 - b0, _ptr(lat), _ptr(lat_inv), _ptr(Ec),
                                               _ptr(F[b0:b1])))
        return E, F

    def predict_errors(self, R, F_ref, E_ref=None, std=1.0, c=0.0, lat_and_inv=None):
        """Eight error sums of a labelled batch, evaluated on the GPU (see gdml_predict_errors)."""
        n_atoms = self.model_n_atoms
        R = f64(R).reshape(-1, 3 * n_atoms)
        F_ref = f64(F_ref).reshape(R.shape[0], 3 * n_atoms)
        E_ref = None if E_ref is None else f64(E_ref).ravel()
        lat = lat_inv = None
        if lat_and_inv is not None:
            lat, lat_inv = f64(lat_and_inv[0]), f64(lat_and_inv[1])
        out = np.empty(8)
        self._check(self._lib.gdml_predict_errors(self._h, _ptr(R), R.shape[0], _ptr(lat), _ptr(lat_inv), float(std),
                                                  float(c), _ptr(E_ref), _ptr(F_ref), _ptr(out)))
        return out

    def kernel_matvec(self, lam, use_E_cstr, v):
        v = f64(v).ravel()
        out = np.empty_like(v)
        self._check(self._lib.gdml_kernel_matvec(self._h, float(lam), int(bool(use_E_cstr)), _ptr(v), v.size,
                                                 _ptr(out)))
        return out

    def _n_lev(self):
        n_rows, _, _ = self.K_shape()  # rows held by this rank
        _, world = self.comm_info()
        if world <= 1:
            return n_rows
        # sharded: the scores of ALL rows come back, in the reference order (forces, then energy constraints)
        return self.n_train * 3 * self.n_atoms + (self.n_train if getattr(self, '_last_use_E', False) else 0)

    def nystroem_factor(self, lam, idx, want_factor=False, want_lev=True):
        """(leverage scores or None, factor or None, info).  info is a bit field (gdml_nystroem_factor): bit 0 = the second
        Cholesky failed and the QR-equivalent branch ran (iterative.py:313-324); bit 1 = the preconditioner is applied
        matrix-free (option pcg.precon_form); info >> 8 = jitter escalations of the first Cholesky (iterative.py:442-463).
        want_lev=False: the scores are computed on demand by nystroem_lev_scores()."""
        idx = i64(idx)
        n_rows, _, _ = self.K_shape()
        lev = np.empty(self._n_lev()) if want_lev else None
        fac = np.empty((idx.size, n_rows)) if want_factor else None
        info = C.c_int(0)
        self._check(self._lib.gdml_nystroem_factor(self._h, float(lam), _ptr(idx), idx.size, _ptr(lev),
                                                   _ptr(fac), C.byref(info)))
        return lev, fac, info.value

    def nystroem_lev_scores(self):
        """Leverage scores of the resident Nystroem factor (gdml_nystroem_lev_scores): valid until the next assembly."""
        lev = np.empty(self._n_lev())
        self._check(self._lib.gdml_nystroem_lev_scores(self._h, _ptr(lev)))
        return lev

    def precon_apply(self, lam, v):
        v = f64(v).ravel()
        out = np.empty_like(v)
        self._check(self._lib.gdml_precon_apply(self._h, float(lam), _ptr(v), v.size, _ptr(out)))
        return out

    def pcg(self, lam, use_E_cstr, y, x0=None, rtol=1e-4, maxiter=1000, use_precon=True, callback=None,
            cb_every=1):
        """Preconditioned CG on the device (gdml_pcg).  callback(it, resid, fetch_x) is called every cb_every iterations;
        fetch_x() copies the iterate of THAT iteration to the host (the only way it crosses PCIe before the solve ends);
        a true return value stops the solve with that iterate.  Returns (x, info, iterations, residual norm)."""
        y = f64(y).ravel()
        n = y.size
        x0 = f64(x0)
        x = np.empty(n)
        iters, resid, info = C.c_int64(), C.c_double(), C.c_int()
        exc = []

        def _fetch_x():
            xk = np.empty(n)
            self._check(self._lib.gdml_pcg_x(self._h, _ptr(xk)))
            return xk

        def _cb(it, res, _user):
            try:
                return int(bool(callback(int(it), float(res), _fetch_x)))
            except BaseException as e:  # propagate after the C call returns
                exc.append(e)
                return 1

        cb = PCG_CB(_cb) if callback is not None else C.cast(None, PCG_CB)
        rc = self._lib.gdml_pcg(self._h, float(lam), int(bool(use_E_cstr)), _ptr(y), _ptr(x0), n, float(rtol),
                                int(maxiter), int(bool(use_precon)), cb, int(cb_every if callback else 0), None,
                                _ptr(x), C.byref(iters), C.byref(resid), C.byref(info))
        if exc:
            raise exc[0]
        self._check(rc)
        return x, info.value, iters.value, resid.value
