"""Round-5 GPU tests (real MI355X, through the C ABI): the forms of the Nystroem preconditioner (stored fp64 factor, fp32
factor + Gram correction, matrix-free) against each other and against the reference's PCG runs, the lazily evaluated
leverage scores, and the process-level device arena (gdml_mem_reserve)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import gdml_oracle as orc  # noqa: E402
from _pcg_compare import assert_same_convergence  # noqa: E402
from tests.test_oracle_golden import _lat, _model, cancel_floor  # noqa: E402

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=True))


def _tril_perms(g):
    from sgdml_amd import _lib

    return _lib.tril_perms_from_lin(g['tril_perms_lin'], g['R_desc'].shape[1])


@pytest.fixture
def ctx():
    from sgdml_amd import _lib

    c = _lib.Context(0)
    yield c
    c.close()


# ------------------------------------------------------------------ matrix-free preconditioner


def test_preconditioner_forms_agree(golden, ctx):
    """pcg.precon_form on every small fixture (energy constraints, a permutation group, PBC included): 0 = the reference's
    stored fp64 factor; 3 = the factor rounded to fp32 + the m x m Gram correction T0 (half the bytes per application);
    1 = matrix-free (two kernel mat-vecs + the m x m matrix Z = L_mm^-T L^-T).  A single application agrees with the
    reference's factor to 1e-6 / 1e-5 of its norm in all three (that this is NOT enough for form 1 inside PCG at
    lam = 1e-10 is recorded in profiles/r05_pcg_bisect.txt).  The leverage scores come out of gdml_nystroem_lev_scores on
    demand; the info bits report the form."""
    g = golden
    lam, sig, use_E = float(g['lam']), float(g['sig']), bool(g['use_E_cstr'])
    idx = g['col_idxs']
    m = len(idx)
    tp = _tril_perms(g)
    ctx.train_upload(g['R_desc'], g['R_d_desc'], tp)
    rng = np.random.default_rng(5)
    vs = [g['v'], rng.standard_normal(len(g['v']))]
    out = {}
    for form, bits in ((0, (0,)), (1, (2,)), (3, (4, 0))):
        ctx.set_option('pcg.precon_form', form)
        ctx.assemble_K(sig, use_E, idx=idx, alloc_extra_rows=m)
        lev, _, info = ctx.nystroem_factor(lam, idx, want_lev=False)
        assert lev is None and (info & 6) in bits and (info & 1) == 0, (form, info)
        if form == 3:
            # the fp32 form needs the rounded factor's Gram matrix well conditioned (pcg.f32_min_pivot); where the inducing
            # columns are not supported by the data the library keeps the reference's form and says so
            piv = ctx.get_option('pcg.f32_last_min_pivot')
            assert ((info & 6) == 4) == (piv is not None and piv >= 1e-7), (info, piv)
        pv = [ctx.precon_apply(lam, v) for v in vs]  # the matrix-free form builds its own operator model when none is resident
        out[form] = (pv, ctx.nystroem_lev_scores(), info)
    ref = [orc.precon_apply(g['L_inv_K_mn'], lam, v) for v in vs]
    for k in range(2):
        scale = np.abs(ref[k]).max()
        assert np.abs(out[0][0][k] - ref[k]).max() <= 1e-6 * scale
        assert np.abs(out[1][0][k] - out[0][0][k]).max() <= 1e-6 * scale, np.abs(out[1][0][k] - out[0][0][k]).max() / scale
        # where the fp32 form was accepted: 1e-5 of the operator.  One exception: lam = 1e-10 with a weak inducing direction
        # (n4_p6_pbc: smallest squared singular value 0.77) -- such a direction weighs lam / (sigma_i + lam) ~ 0.2 in the
        # operator, and there it is the REFERENCE's stored factor whose Gram matrix is off the exact I - lam L^-1 L^-T by the
        # rounding error of K_nm^T K_nm relative to lam (1e-13 / 1e-10).  Round 6: on n10_p2_pbc (lam = 1e-4, weakest direction
        # 0.06) this assertion found that round 5's T0 used L0 L0^T where the operator needs L0^T L0 (3 % off; csrc/cg.hip).
        piv = ctx.get_option('pcg.f32_last_min_pivot')
        if out[3][2] & 4:
            tol3 = 1e-5 if (piv >= 0.99 or lam >= 1e-6) else 1e-2
            assert np.abs(out[3][0][k] - out[0][0][k]).max() <= tol3 * scale, (np.abs(out[3][0][k] - out[0][0][k]).max() / scale, piv)
    lev_ref = np.einsum('ij,ij->j', g['L_inv_K_mn'], g['L_inv_K_mn'])
    np.testing.assert_allclose(out[0][1], lev_ref, rtol=0, atol=1e-6)
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(out[3][1], out[0][1], rtol=1e-9, atol=1e-12)  # the fp64 factor stays resident next to the fp32 copy
    # the alternative (QR-equivalent) branch of the second factorisation: Z is carried through its three solves; the fp32
    # form has no Cholesky factor to take the exact Gram from and falls back to the reference's form
    ctx.set_option('nys.force_qr', 1)
    for form, bits in ((1, 2), (3, 0)):
        ctx.set_option('pcg.precon_form', form)
        ctx.assemble_K(sig, use_E, idx=idx, alloc_extra_rows=m)
        _, _, info = ctx.nystroem_factor(lam, idx, want_lev=False)
        assert info & 1 and (info & 6) == bits
        pq = ctx.precon_apply(lam, vs[0])
        assert np.abs(pq - ref[0]).max() <= 1e-5 * np.abs(ref[0]).max()


@pytest.mark.parametrize('name,maxit', [('pcg_n9_m400', 5000), ('pcg_n12_p6_m200', 20000), ('cfg2_traj_m300', 5000)])
def test_fp32_preconditioner_on_the_reference_pcg_runs(name, maxit):
    """gdml_pcg with the fp32-stored, Gram-corrected preconditioner (pcg.precon_form = 3) on the three reference PCG fixtures
    (the inducing columns the reference drew).  It is not the reference's operator bit for bit -- it is the exact Woodbury
    inverse on the rounded factor's column space, where the stored fp64 factor carries the spectrum 1 - lam/(sigma + lam)
    only to ~1e-10 -- so the statement is: same first steps (1e-3: the residual moves by 1e-4 of its norm per step), converged to
    the same tolerance in NO MORE iterations than the reference (+10 %), true residual below tolerance, same predictions."""
    from sgdml_amd import _lib
    from sgdml_amd.utils.desc import Desc

    g = load(name)
    M, N = g['R_train'].shape[:2]
    sig, lam, y, idx = float(g['sig']), float(g['lam']), g['y'], g['inducing_pts_idxs']
    tp = orc.tril_perms_from_atom_perms(g['perms'])
    c = _lib.Context()
    try:
        xd, gd = c.desc_from_R(g['R_train'].reshape(M, -1), N)
        c.train_upload(xd, gd, tp)
        runs = {}
        for form in (0, 3):
            c.set_option('pcg.precon_form', form)
            c.assemble_K(sig, False, idx=idx, alloc_extra_rows=len(idx))
            _, _, info = c.nystroem_factor(lam, idx, want_lev=False)
            assert (info & 6) == (4 if form == 3 else 0)
            c.predict_upload_model(xd, np.zeros_like(xd), tp, sig, None)
            hist = []
            x, inf, iters, resid = c.pcg(lam, False, y, rtol=1e-4, maxiter=maxit,
                                         callback=lambda it, r, fetch_x: hist.append(r) or False)
            assert inf == 0
            runs[form] = (x, iters, np.array(hist))
        n_ref, ref = int(g['n_iters']), g['resid_hist']
        x, iters, ours = runs[3]
        assert iters <= n_ref + max(2, n_ref // 10), (iters, runs[0][1], n_ref)
        assert iters >= n_ref // 4, (iters, runs[0][1], n_ref)
        np.testing.assert_allclose(ours[:8], ref[:8], rtol=1e-3)
        r = c.kernel_matvec(lam, False, x) + y
        assert np.linalg.norm(r) <= 1.05e-4 * np.linalg.norm(y)
        d = Desc(N)
        F = []
        for coeffs in (g['alphas'], -x):
            c.predict_upload_model(xd, d.d_desc_dot_vec(gd, coeffs.reshape(M, -1)), tp, sig, None)
            F.append(c.predict(g['R_test'].reshape(len(g['R_test']), -1))[1])
        assert np.abs(F[1] - F[0]).max() <= 5e-3 * np.abs(F[0]).max()
    finally:
        c.close()


def test_fp32_form_through_the_dropin_solver():
    """GDMLTrain.train -> Iterative.solve with pcg.precon_form = 3: the reference's inducing columns under its seed,
    converged in no more iterations than the reference (+10 %), and the restart policy gets its leverage scores (computed
    on demand from the resident fp64 factor)."""
    from sgdml_amd.predict import GDMLPredict
    from sgdml_amd.solvers.iterative import Iterative
    from sgdml_amd.train import GDMLTrain

    g = load('pcg_n12_p6_m200')
    M, N = g['R_train'].shape[:2]
    task = {
        'type': 't', 'code_version': '1.0.3', 'dataset_name': np.array('synth'), 'dataset_theory': np.array('pair'),
        'z': g['z'], 'R_train': g['R_train'], 'F_train': g['F_train'], 'E_train': g['E_train'],
        'idxs_train': np.arange(M), 'md5_train': 'x', 'idxs_valid': np.arange(0), 'md5_valid': 'x',
        'sig': int(g['sig']), 'lam': float(g['lam']), 'use_E': True, 'use_E_cstr': False, 'use_sym': True,
        'perms': g['perms'],
    }
    seen = {}
    orig = Iterative._init_precon_operator

    def spy(self, *a, **kw):
        out = orig(self, *a, **kw)
        seen['form'] = self.precon_form
        seen['lev'] = self._lev_scores_now()
        return out

    Iterative._init_precon_operator = spy
    tr = GDMLTrain()
    try:
        tr._force_solver = 'cg'
        tr._force_n_inducing_pts = int(g['k'])
        tr._context().set_option('pcg.precon_form', 3)
        np.random.seed(int(g['seed']))
        model = tr.train(task)
    finally:
        Iterative._init_precon_operator = orig
        tr.__del__()
    assert seen['form'] == 'fp32' and tr._last_precon_form == 'fp32'
    assert seen['lev'].shape == (3 * N * M,) and np.all(seen['lev'] >= 0) and seen['lev'].max() <= 1.0 + 1e-6
    # the first preconditioner is built on the reference's inducing columns; the run may go through the reference's restart
    # policy (100 steps without net progress on this system's plateau -> 1.2 x more inducing points, iterative.py:755-801)
    k_final = len(model['inducing_pts_idxs']) // (3 * N)
    assert k_final in (int(g['k']), int(np.ceil(1.2 * int(g['k']))))
    if k_final == int(g['k']):
        assert np.array_equal(model['inducing_pts_idxs'], g['inducing_pts_idxs'])
    assert model['solver_resid'] <= model['solver_tol'] * model['norm_y_train']
    n_ref = int(g['n_iters'])
    assert int(model['solver_iters']) <= n_ref + max(2, n_ref // 10)
    E, F = GDMLPredict(model).predict(g['R_test'].reshape(len(g['R_test']), -1))
    assert np.abs(F - g['F_test']).max() <= 5e-3 * np.abs(g['F_test']).max()


def test_precon_form_is_chosen_by_size():
    """pcg.precon_form = 2 (default): the reference's stored fp64 factor for every system of the parity suite, the fp32 form
    once the factor reaches 1 GiB per rank (N = 21, M = 1500, k = 40: n = 94 500, m = 2520, 1.9 GB) -- where it must agree
    with the stored form's action to 1e-5."""
    from sgdml_amd import _lib

    c = _lib.Context()
    try:
        for N, M, k, want in ((9, 60, 4, 0), (21, 1500, 40, 4)):
            ds = orc.synth_dataset(N, M, seed=3, jitter=0.3)
            xd, gd = c.desc_from_R(ds['R'].reshape(M, -1), N)
            tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
            c.train_upload(xd, gd, tp)
            idx = np.sort(np.random.default_rng(1).choice(3 * N * M, 3 * N * k, replace=False))
            c.set_option('pcg.precon_form', 2)
            c.assemble_K(20.0, False, idx=idx, alloc_extra_rows=len(idx))
            _, _, info = c.nystroem_factor(1e-10, idx, want_lev=False)
            assert (info & 6) == want, (N, M, k, info)
            v = np.random.default_rng(2).standard_normal(3 * N * M)
            a = c.precon_apply(1e-10, v)
            c.set_option('pcg.precon_form', 3 if want == 0 else 0)
            c.assemble_K(20.0, False, idx=idx, alloc_extra_rows=len(idx))
            c.nystroem_factor(1e-10, idx, want_lev=False)
            b = c.precon_apply(1e-10, v)
            c.set_option('pcg.precon_form', 2)
            assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max()
    finally:
        c.close()


# ------------------------------------------------------------------ single-launch prediction


@pytest.mark.parametrize('B', [1, 2, 3, 5, 7])
def test_single_launch_prediction(golden, ctx, B):
    """Host batches of up to eight geometries take ONE launch (descriptors from the kernel arguments, contraction,
    last-workgroup reduction and back-projection, E / F through host-mapped memory): equal to the three-kernel
    path (option predict.fused = 0) to rounding and to the reference's predictions, on every small fixture -- permutation
    group, energy-constraint coefficients, the periodic cell -- with and without energies, repeated calls, every row split."""
    g = golden
    m = _model(g)
    tp = _tril_perms(g)
    ctx.predict_upload_model(np.ascontiguousarray(m['R_desc'].T), m['R_d_desc_alpha'], tp, float(g['sig']), m.get('alphas_E'))
    R = g['R_test'].reshape(len(g['R_test']), -1)[:B]
    fl = cancel_floor(g)
    ctx.set_option('predict.fused', 0)
    E0, F0 = ctx.predict(R, _lat(g))
    ctx.set_option('predict.fused', 1)
    for rows in (16, 2, 5, 64):
        ctx.set_option('predict.fused_rows', rows)
        for rep in range(3):
            E1, F1 = ctx.predict(R, _lat(g))
            assert np.abs(F1 - F0).max() <= 1e-12 * np.abs(F0).max() + fl / float(m['std'])
            # (the energy-constraint terms sum coefficients of size 1e9 in another order: 10 x the force floor)
            assert np.abs(E1 - E0).max() <= 1e-12 * max(1.0, np.abs(E0).max()) + 10 * fl * float(g['sig']) / float(m['std'])
    ctx.set_option('predict.fused_rows', 16)
    E1, F1 = ctx.predict(R, _lat(g))
    assert np.abs(F1 * m['std'] - g['F_test'][:B]).max() <= 1e-10 * np.abs(g['F_test']).max() + fl
    assert np.abs(E1 * m['std'] + m['c'] - g['E_test'][:B]).max() <= 1e-10 * max(1.0, np.abs(g['E_test']).max()) + fl * float(g['sig'])
    E2, F2 = ctx.predict(R, _lat(g), return_E=False)
    assert E2 is None and np.array_equal(F2, F1)
    ctx.set_option('predict.fused_spin', 0)  # completion through the stream instead of the polled sequence number
    E3, F3 = ctx.predict(R, _lat(g))
    ctx.set_option('predict.fused_spin', 1)
    assert np.array_equal(F3, F1) and np.array_equal(E3, E1)


def test_single_launch_prediction_aspirin_size():
    """The shape the latency path is quoted on (N = 21, D = 210, 1000 training points; also a 27-element group at N = 12):
    one to four queries against the oracle, 1e-10."""
    from sgdml_amd import _lib

    c = _lib.Context()
    try:
        for N, M, perms in ((21, 1000, None), (23, 300, None), (12, 150, 'g6')):
            ds = orc.synth_dataset(N, M + 4, seed=4, jitter=0.3)
            Rf = ds['R'].reshape(M + 4, -1)
            if perms is None:
                tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
            else:
                tp = orc.tril_perms_from_atom_perms(load('pcg_n12_p6_m200')['perms'])
            xo, go = orc.desc_from_R(Rf[:M])
            JA = np.random.default_rng(1).standard_normal(xo.shape)
            c.predict_upload_model(xo, JA, tp, 20.0, None)
            xq, gq = orc.desc_from_R(Rf[M:])
            Eo, Fo = orc.predict_from_desc(xq, gq, xo, JA, tp, 20.0)
            for B in (1, 2, 4):
                if B * 3 * N > 384:
                    continue
                E, F = c.predict(Rf[M:M + B])
                assert np.abs(F - Fo[:B]).max() <= 1e-10 * np.abs(Fo).max()
                assert np.abs(E - Eo[:B]).max() <= 1e-10 * np.abs(Eo).max()
    finally:
        c.close()


# ------------------------------------------------------------------ device arena


def test_mem_reserve_arena_lifecycle():
    """gdml_mem_reserve: reserve -> the large matrices of two contexts are carved from the block in turn (first fit; no
    growth of driver-side use) -> a third that fits no gap goes to hipMalloc -> handing a matrix back frees its gap -> the
    block survives gdml_ctx_destroy and serves the next context -> reserve(0) refuses while a buffer is carved and releases
    afterwards; gdml_mem_info counts the LARGEST contiguous gap of the arena as free (round 6: a carve needs one gap, so the
    sum of the idle pieces over-promised on a fragmented arena); K is the same matrix through the arena as through hipMalloc."""
    import ctypes as C

    from sgdml_amd import _lib

    GB = 1 << 30
    N, M = 21, 400  # n = 25 200: K = 5.08 GB (4.73 GiB), above the arena's 1 GiB threshold
    KB = 25200 * 25200 * 8
    ds = orc.synth_dataset(N, M, seed=2, jitter=0.3)
    tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
    rows = np.random.default_rng(0).choice(3 * N * M, 64, replace=False)

    def sample_K(c):
        K_rows, K_cols, _ = c.K_shape()
        p, ld = C.c_void_p(), C.c_int64()
        c._check(c._lib.gdml_K_dev(c._h, C.byref(p), C.byref(ld)))
        out = np.empty((len(rows), ld.value))
        for i, r in enumerate(rows):
            c._check(c._lib.gdml_memcpy_d2h(c._h, out[i].ctypes.data_as(C.c_void_p), C.c_void_p(p.value + int(r) * ld.value * 8),
                                            ld.value * 8))
        return out[:, :K_cols], p.value

    ctxs = []

    def new_ctx():
        c = _lib.Context(0)
        ctxs.append(c)
        return c

    try:
        a = new_ctx()
        xd, gd = a.desc_from_R(ds['R'].reshape(M, -1), N)
        a.train_upload(xd, gd, tp)
        a.assemble_K(20.0, False)  # plain hipMalloc
        K_plain, _ = sample_K(a)
        a.close()
        a = new_ctx()
        _, free0, total = a.mem_info()
        assert a.mem_reserve(12 * GB) == 12 * GB
        _, free1, _ = a.mem_info()
        assert abs(free1 - free0) <= GB // 4  # an idle arena is free memory as far as large buffers are concerned
        assert a.mem_reserve(12 * GB) == 12 * GB  # same size again: no-op
        a.train_upload(xd, gd, tp)
        a.assemble_K(20.0, False)
        K_a, p_a = sample_K(a)
        assert np.array_equal(K_a, K_plain)
        held_a, free2, _ = a.mem_info()
        assert held_a >= KB
        assert abs((free1 - free2) - KB) <= GB // 4, (free1, free2)  # exactly the carved block left the free count
        with pytest.raises(_lib.GDMLHipError):
            a.mem_reserve(0)  # in use
        # second context: the next gap of the same block
        b = new_ctx()
        b.train_upload(xd, gd, tp)
        b.assemble_K(20.0, False)
        K_b, p_b = sample_K(b)
        assert np.array_equal(K_b, K_plain)
        assert p_a <= p_b < p_a + 12 * GB and abs(p_b - p_a) >= KB
        _, free3, _ = b.mem_info()
        assert abs((free2 - free3) - KB) <= GB // 4
        # third context: 12 GB hold two 4.73 GiB matrices, not three -> hipMalloc (outside the block, driver memory drops)
        c = new_ctx()
        c.train_upload(xd, gd, tp)
        c.assemble_K(20.0, False)
        K_c, p_c = sample_K(c)
        assert np.array_equal(K_c, K_plain)
        assert not (min(p_a, p_b) <= p_c < min(p_a, p_b) + 12 * GB)
        _, free4, _ = c.mem_info()
        assert abs((free3 - free4) - KB) <= GB // 4
        # a goes away: its gap is idle again (the block stays with the process) and serves the next large request
        a.close()
        _, free5, _ = b.mem_info()
        # layout now: [gap of a's matrix | b's matrix | tail]: the largest gap is a's former block, not gap + tail
        tail = 12 * GB - 2 * KB
        assert abs((free5 - free4) - (KB - tail)) <= GB // 4, (free4, free5)
        d = new_ctx()
        d.train_upload(xd, gd, tp)
        d.assemble_K(20.0, False)
        _, p_d = sample_K(d)
        assert p_d == p_a
        for k in (b, c, d):
            k.close()
        e = new_ctx()
        assert e.mem_reserve(0) == 0
        _, free6, _ = e.mem_info()
        assert abs(free6 - free0) <= GB // 4
    finally:
        for k in ctxs:
            k.close()
        z = _lib.Context(0)
        try:
            z.mem_reserve(0)
        except Exception:
            pass
        z.close()
