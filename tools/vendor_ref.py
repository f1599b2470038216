#!/usr/bin/env python
"""Vendor reference points on the same box (measurement only; nothing of this is in the product path): rocBLAS dsyrk /
dgemm with the trailing update's shape (n x n, K = 1024) and rocSOLVER dpotrf at the benchmark order n = 63 000.
    python tools/vendor_ref.py [n]      -> one line per routine: ms, TFLOP/s, fraction of the 78.6 TFLOP/s fp64-MFMA peak"""
import ctypes as C
import sys
import time

import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 63000
K = 1024
hip = C.CDLL('libamdhip64.so')
blas = C.CDLL('librocblas.so')
sol = C.CDLL('librocsolver.so')
PEAK = 78.6


def chk(rc, what):
    if rc != 0:
        raise SystemExit('%s failed: %d' % (what, rc))


def dmalloc(nbytes):
    p = C.c_void_p()
    chk(hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)), 'hipMalloc')
    return p


hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipMemcpy2D.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
h = C.c_void_p()
chk(blas.rocblas_create_handle(C.byref(h)), 'rocblas_create_handle')
A = dmalloc(n * n * 8)          # n x n, column major, ld = n
P = dmalloc(n * K * 8)          # n x K panel
info = dmalloc(8)
ones = np.ones(n)
half = np.full(n * K, 1e-3)
chk(hip.hipMemcpy(P, half.ctypes.data_as(C.c_void_p), n * K * 8, 1), 'H2D')
one, mone = C.c_double(1.0), C.c_double(-1.0)
LOWER, NONE, TRANS = 122, 111, 112
blas.rocblas_dsyrk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_int,
                               C.POINTER(C.c_double), C.c_void_p, C.c_int]
blas.rocblas_dgemm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_int,
                               C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_int]
sol.rocsolver_dpotrf.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]


def reset_A():  # identity (positive definite; the flop count of potrf does not depend on the values)
    chk(hip.hipMemset(A, 0, n * n * 8), 'memset')
    chk(hip.hipMemcpy2D(A, (n + 1) * 8, ones.ctypes.data_as(C.c_void_p), 8, 8, n, 1), 'diag')
    hip.hipDeviceSynchronize()


def timed(label, flops, fn, reps=3):
    best = 1e30
    for _ in range(reps):
        hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        chk(fn(), label)
        hip.hipDeviceSynchronize()
        best = min(best, time.perf_counter() - t0)
    tf = flops / best / 1e12
    print('%-44s %9.2f ms  %6.2f TFLOP/s = %.3f of the fp64-MFMA peak' % (label, best * 1e3, tf, tf / PEAK), flush=True)


reset_A()
timed('rocblas_dsyrk lower, n=%d, K=%d' % (n, K), float(n) * n * K,
      lambda: blas.rocblas_dsyrk(h, LOWER, NONE, n, K, C.byref(mone), P, n, C.byref(one), A, n))
timed('rocblas_dgemm NT, %d x %d, K=%d (full square)' % (n, n, K), 2.0 * n * n * K,
      lambda: blas.rocblas_dgemm(h, NONE, TRANS, n, n, K, C.byref(mone), P, n, P, n, C.byref(one), A, n))
reset_A()
t0 = time.perf_counter()
chk(sol.rocsolver_dpotrf(h, LOWER, n, A, n, info), 'rocsolver_dpotrf')
hip.hipDeviceSynchronize()
dt = time.perf_counter() - t0
print('%-44s %9.2f ms  %6.2f TFLOP/s = %.3f of the fp64-MFMA peak  (first call)' % ('rocsolver_dpotrf lower, n=%d' % n, dt * 1e3,
                                                                                 n**3 / 3.0 / dt / 1e12, n**3 / 3.0 / dt / 1e12 / PEAK))
reset_A()
timed('rocsolver_dpotrf lower, n=%d' % n, n**3 / 3.0, lambda: (sol.rocsolver_dpotrf(h, LOWER, n, A, n, info)), reps=1)
