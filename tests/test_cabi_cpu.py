"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol declared in
include/gdml_hip.h, the ctypes table matches the header, and the product path fails loudly
(no CPU fallback) when no GPU is present."""
import os
import re

import numpy as np
import pytest

from sgdml_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, 'include', 'gdml_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(gdml_[a-zA-Z0-9_]+)\s*\(', src)) - {'gdml_pcg_cb', 'gdml_host_allreduce', 'gdml_host_allgather'})


def test_header_and_ctypes_table_agree():
    assert _header_functions() == sorted(_lib.SIGNATURES.keys())


def test_library_exports_every_header_symbol():
    lib = _lib.load()
    for name in _header_functions():
        assert hasattr(lib, name), name
    assert lib.gdml_abi_version() == 4


def test_no_cpu_fallback_without_gpu():
    import ctypes as C

    lib = _lib.load()
    n = C.c_int(-1)
    lib.gdml_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip('GPU present')
    with pytest.raises(_lib.GDMLHipError):
        _lib.Context()
    from sgdml_amd.train import GDMLTrain

    t = GDMLTrain()
    try:
        with pytest.raises(_lib.GDMLHipError):
            t._context()
    finally:
        t.__del__()


def test_product_package_never_imports_oracle_or_reference():
    """The product (sgdml_amd/, bench.py outside its cpu_baseline leg) must reach neither the oracle nor the
    reference package: no `oracle` token at all, no `import sgdml` / `from sgdml` / `sgdml.` module access."""
    ref_import = re.compile(r'^\s*(import\s+sgdml(\s|\.|$)|from\s+sgdml(\s|\.))', re.M)
    ref_dyn = re.compile(r"import_module\(\s*['\"]sgdml(['\".])|__import__\(\s*['\"]sgdml(['\".])")
    pkg = os.path.join(ROOT, 'sgdml_amd')
    seen = 0
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h')):
                path = os.path.join(dp, f)
                txt = open(path).read()
                seen += 1
                assert 'oracle' not in txt.replace('no CPU fallback', ''), path
                if f.endswith('.py'):
                    assert not ref_import.search(txt), path
                    assert not ref_dyn.search(txt), path
    assert seen > 10


def test_options_env_is_the_only_environment_variable_the_library_reads():
    for f in os.listdir(os.path.join(ROOT, 'sgdml_amd', 'csrc')):
        if f.endswith(('.hip', '.h')):
            txt = open(os.path.join(ROOT, 'sgdml_amd', 'csrc', f)).read()
            for m in re.finditer(r'getenv\("([A-Z_]+)"\)', txt):
                assert m.group(1) == 'GDML_OPTIONS', (f, m.group(1))


def test_tril_perms_roundtrip():
    from oracle import gdml_oracle as orc

    perms = np.array([[0, 1, 2, 3, 4], [1, 0, 2, 3, 4], [0, 1, 3, 2, 4], [1, 0, 3, 2, 4]])
    tp = orc.tril_perms_from_atom_perms(perms)
    lin = orc.tril_perms_lin_from_tril_perms(tp)
    assert np.array_equal(_lib.tril_perms_from_lin(lin, tp.shape[1]), tp)


def test_desc_host_helpers_match_oracle():
    from oracle import gdml_oracle as orc
    from sgdml_amd.utils.desc import Desc

    rs = np.random.RandomState(0)
    N, M = 6, 4
    R = rs.normal(size=(M, 3 * N)) * 2
    xd, gd = orc.desc_from_R(R)
    d = Desc(N)
    v = rs.normal(size=(M, 3 * N))
    np.testing.assert_allclose(d.d_desc_dot_vec(gd, v), orc.d_desc_dot_vec(gd, v), rtol=1e-13)
    f = rs.normal(size=(M, d.dim))
    np.testing.assert_allclose(d.vec_dot_d_desc(gd, f), orc.vec_dot_d_desc(gd, f), rtol=1e-13, atol=1e-14)
    full = d.d_desc_from_comp(gd)
    np.testing.assert_allclose(full, orc.d_desc_from_comp(gd))
    np.testing.assert_allclose(d.d_desc_to_comp(full), gd)
    p = np.array([2, 0, 1, 3, 5, 4])
    assert np.array_equal(Desc.perm(p), orc.desc_perm(p))


def test_draw_strat_sample_properties():
    from sgdml_amd.train import GDMLTrain

    t = GDMLTrain()
    try:
        np.random.seed(3)
        T = np.random.normal(size=500)
        idx = t.draw_strat_sample(T, 50)
        assert len(idx) == 50 and len(set(idx.tolist())) == 50
        excl = idx[:10]
        idx2 = t.draw_strat_sample(T, 40, excl_idxs=excl)
        assert len(idx2) == 40 and not set(idx2.tolist()) & set(excl.tolist())
        assert np.array_equal(t.draw_strat_sample(T[:7], 7), np.arange(7))
        with pytest.raises(Exception):
            GDMLTrain()  # singleton, train.py:336-342
    finally:
        t.__del__()


def test_dataset_md5_matches_reference_recipe():
    import hashlib

    from sgdml_amd.utils import io

    rs = np.random.RandomState(1)
    ds = {'z': np.arange(3), 'R': rs.normal(size=(4, 3, 3)), 'E': rs.normal(size=4), 'F': rs.normal(size=(4, 3, 3))}
    h = hashlib.md5()
    for k in ['z', 'R', 'E', 'F']:
        h.update(hashlib.md5(ds[k].ravel()).digest())
    assert io.dataset_md5(ds) == h.hexdigest().encode('utf-8')


def test_ase_calculator_optional_dependency():
    """intf/ase_calc.py mirrors the reference: ASE is optional and its absence is an ImportError
    (reference ase_calc.py:25-31)."""
    import importlib
    import sys

    try:
        import ase  # noqa: F401
        pytest.skip('ASE is installed here')
    except ImportError:
        pass
    sys.modules.pop('sgdml_amd.intf.ase_calc', None)
    with pytest.raises(ImportError, match='Optional ASE dependency not found'):
        importlib.import_module('sgdml_amd.intf.ase_calc')


def _toy_dataset(n=40, n_atoms=4, with_E=True, seed=0):
    rs = np.random.RandomState(seed)
    ds = {
        'type': 'd', 'name': np.array('toy'), 'theory': np.array('none'), 'z': np.array([6, 1, 1, 8][:n_atoms]),
        'R': rs.normal(size=(n, n_atoms, 3)), 'F': rs.normal(size=(n, n_atoms, 3)),
    }
    if with_E:
        ds['E'] = rs.normal(size=n)
    return ds


def test_create_task_schema_and_sampling():
    """Host logic of GDMLTrain.create_task (reference train.py:370-524): schema keys, disjoint train/valid
    samples from one dataset, explicit permutations, singleton rule, error conventions.  No GPU involved."""
    from sgdml_amd.train import GDMLTrain

    ds = _toy_dataset()
    perms = np.array([[0, 1, 2, 3], [0, 2, 1, 3]])
    np.random.seed(0)
    tr = GDMLTrain()
    try:
        with pytest.raises(Exception):
            GDMLTrain()  # one instance per process (train.py:336-342)
        task = tr.create_task(ds, 10, ds, 12, sig=10, lam=1e-8, perms=perms)
        for key in ('type', 'code_version', 'dataset_name', 'dataset_theory', 'z', 'R_train', 'F_train', 'E_train',
                    'idxs_train', 'md5_train', 'idxs_valid', 'md5_valid', 'sig', 'lam', 'use_E', 'use_E_cstr',
                    'use_sym', 'perms'):
            assert key in task, key
        assert task['type'] == 't' and task['R_train'].shape == (10, 4, 3) and task['E_train'].shape == (10,)
        assert len(set(task['idxs_train'])) == 10 and len(set(task['idxs_valid'])) == 12
        assert not set(task['idxs_train']) & set(task['idxs_valid'])  # same dataset -> disjoint
        assert task['md5_train'] == task['md5_valid']
        assert np.array_equal(task['perms'], perms) and task['use_E_cstr'] is False
        # use_sym=False -> identity only
        t2 = tr.create_task(ds, 5, ds, 5, sig=10, use_sym=False)
        assert np.array_equal(t2['perms'], np.arange(4)[None])
        # energy constraints need energies; use_E=False drops them and forces use_E_cstr off
        t3 = tr.create_task(_toy_dataset(with_E=False), 5, _toy_dataset(with_E=False, seed=1), 5, sig=10,
                            use_E=False, use_E_cstr=True, use_sym=False)
        assert 'E_train' not in t3 and t3['use_E_cstr'] is False
        with pytest.raises(ValueError, match='No energy labels'):
            tr.create_task(_toy_dataset(with_E=False), 5, ds, 5, sig=10)
        with pytest.raises(ValueError, match='do not match the number of atoms'):
            tr.create_task(ds, 5, ds, 5, sig=10, perms=np.arange(3)[None])
        bad = dict(ds, lattice=np.zeros((3, 3)))
        with pytest.raises(ValueError, match='invalid lattice'):
            tr.create_task(bad, 5, ds, 5, sig=10, use_sym=False)
    finally:
        tr.__del__()
    tr2 = GDMLTrain()  # the slot is free again after __del__ (train.py:363-368)
    tr2.__del__()


GOLD = os.path.join(ROOT, 'tests', 'golden')


def test_find_perms_matches_reference():
    """Own symmetry discovery (sgdml_amd/utils/perm.py) vs the reference's find_perms output on three seeded
    molecules (tests/golden/make_golden_r2.py::case_perm_c3): same group, same order."""
    from sgdml_amd.utils import perm

    g = np.load(os.path.join(GOLD, 'perm_c3.npz'))
    assert np.array_equal(perm.find_perms(g['R'], g['z']), g['perms'])
    assert np.array_equal(perm.find_perms(g['R2'], g['z2']), g['perms2'])
    lat = g['lat']
    assert np.array_equal(perm.find_perms(g['R2'], g['z2'], lat_and_inv=(lat, np.linalg.inv(lat))), g['perms3'])
    assert g['perms'].shape[0] == 6 and g['perms2'].shape[0] == 3


def test_perm_group_helpers():
    from sgdml_amd.utils import perm

    gens = np.array([[0, 1, 2, 3, 4], [1, 2, 0, 3, 4], [0, 1, 2, 4, 3]])
    grp = perm.complete_sym_group(gens)
    assert grp.shape == (6, 5) and len({tuple(p) for p in grp}) == 6
    assert np.array_equal(grp[:3], gens)  # candidates first, discoveries appended
    assert perm.complete_sym_group(gens, n_perms_max=5) is None  # closure abandoned at the cap
    assert sorted(map(sorted, perm.to_cycles([1, 2, 0, 4, 3]))) == [[0, 1, 2], [3, 4]]
    # a 2-cycle that overlaps a 3-cycle of another candidate is dropped, the 3-cycle survives
    kept = perm.salvage_subgroup(np.array([[0, 1, 2, 3], [1, 2, 0, 3], [1, 0, 2, 3]]))
    assert [tuple(p) for p in kept] == [(0, 1, 2, 3), (1, 2, 0, 3)]


def test_draw_strat_sample_reproduces_reference_indices():
    """draw_strat_sample restates train.py:1537-1646 because the RNG call sequence must be identical for a
    drop-in: under the same seed it returns the reference's indices (fixtures from make_golden_r2.py)."""
    from sgdml_amd.train import GDMLTrain

    g = np.load(os.path.join(GOLD, 'strat_sample.npz'))
    t = GDMLTrain()
    try:
        for k in range(int(g['n_cases'])):
            excl = g['excl%d' % k]
            np.random.seed(int(g['seed%d' % k]))
            idx = t.draw_strat_sample(g['T'], int(g['n%d' % k]), excl_idxs=excl if excl.size else None)
            assert np.array_equal(np.asarray(idx, dtype=np.int64), g['idx%d' % k]), k
    finally:
        t.__del__()


def test_documented_options_match_the_library():
    """Every option key the library accepts (kKnownOptions, csrc/ctx.hip) is documented in include/gdml_hip.h and every
    documented key is accepted; every key read with ctx_opt / ctx_opt_i anywhere in csrc/ is a known one."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ctx_src = open(os.path.join(root, 'sgdml_amd', 'csrc', 'ctx.hip')).read()
    blk = ctx_src[ctx_src.index('kKnownOptions[] = {'):]
    blk = blk[:blk.index('};')]
    known = set(re.findall(r'"([a-z_]+\.[a-z0-9_]+)"', blk))
    hdr = open(os.path.join(root, 'include', 'gdml_hip.h')).read()
    doc = hdr[hdr.index('Tuning / ablation options of a context'):hdr.index('Unknown keys return GDML_ERR_INVALID')]
    documented = set(re.findall(r'\b((?:asm|gemm|chol|trsm|trsv|predict|lu|comm|dist|nys|pcg|mem)\.[a-z0-9_]+)', doc))
    assert known == documented, (sorted(known - documented), sorted(documented - known))
    used = set()
    csrc = os.path.join(root, 'sgdml_amd', 'csrc')
    for fn in os.listdir(csrc):
        if fn.endswith(('.hip', '.h')):
            used |= set(re.findall(r'ctx_opt(?:_i)?\(\s*ctx\s*,\s*"([^"]+)"', open(os.path.join(csrc, fn)).read()))
    assert used <= known, sorted(used - known)


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 8` without a launcher around it becomes the launcher (one rank per GPU); on a box with
    fewer GPUs it must refuse, not print a 1-GPU line labelled as the 8-GPU run."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8'], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ''
    assert '--gpus 8' in p.stderr
    # a launcher that started a different number of ranks than --gpus says is an error too
    env.update(WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4'], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0 and p.stdout.strip() == '' and 'launcher started 2 ranks' in p.stderr


def test_find_perms_matches_reference_on_the_cli_sweep_sample():
    """The training sample the reference's `sgdml all` drew in the cli_sweep fixture: our host-side symmetry search
    returns the reference's group, in its order (perm.py:395-404)."""
    from sgdml_amd.utils import perm

    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cli_sweep.npz'))
    got = perm.find_perms(fx['R'][fx['idxs_train']], fx['z'])
    assert np.array_equal(got, fx['perms'])


def test_inducing_point_memory_model_is_per_shard():
    """Iterative.max_n_inducing_pts_device: the largest k whose footprint (this rank's rows of K_nm + the m x m block + its
    backup) fits the budget; the row term shrinks with the number of ranks, the replicated m x m terms do not.  From 1 GiB
    of factor per rank the library builds the fp32 form (rounded in place over the fp64 factor) and needs T0 + four m x m work
    matrices for it (csrc/cg.hip::build_f32_form): the model counts them (round 6)."""
    from sgdml_amd.solvers.iterative import Iterative

    M, N = 5000, 21
    budget = 0.8 * 32 * 1024**3
    ks = [Iterative.max_n_inducing_pts_device(M, N, budget, w) for w in (1, 2, 4, 8)]
    assert ks == sorted(ks) and ks[-1] > ks[0]
    for w, k in zip((1, 2, 4, 8), ks):
        n_loc = -(-M // w) * 3 * N
        foot64 = lambda kk: (n_loc + 2 * 3 * N * kk) * (3 * N * kk) * 8  # (n_loc + m) m + m^2 doubles
        foot32 = lambda kk: (n_loc + 7 * 3 * N * kk) * (3 * N * kk) * 8
        assert n_loc * 3 * N * k * 8 >= 2**30  # every case here is in the fp32-form regime
        assert foot32(k) <= budget < foot32(k + 1) and foot64(k) < foot32(k)
    # below 1 GiB of factor: the fp64 form's footprint
    k_small = Iterative.max_n_inducing_pts_device(200, 9, 0.02 * 1024**3)
    foot = lambda kk: (200 * 27 + 2 * 27 * kk) * (27 * kk) * 8
    assert foot(k_small) <= 0.02 * 1024**3 < foot(k_small + 1)
    assert Iterative.max_n_inducing_pts_device(M, N, budget) == ks[0]  # default: one GPU
    assert Iterative.max_n_inducing_pts_device(10, 3, 1e15, 4) == 10  # never more than the training points
