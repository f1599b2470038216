"""Static checks of compiled code (no GPU): the instruction schedule of the dominant kernel's inner loop.

The trailing-update GEMM of the blocked Cholesky (csrc/chol.hip, gemm_nt_sub_diag_kernel<true, true, 5>: SURVEY.md 8(d), the
kernel `roofline` is quoted on) depends on the ORDER of its phases inside one k-tile: operand pairs of the second half are read
from LDS behind the first 16 MFMAs, the prefetched tile is committed and the barrier passed behind 48, the first half of the
next tile is read behind the barrier and covered by the last 16.  The order is pinned with sched_barriers; in round 4 one was
missing and an unrelated edit of the file let the machine scheduler sink the read-ahead behind two MFMA groups -- 3 % of the
benchmark step, with no change to the loop's source and every numerical test green (profiles/r04_gemm_peel_stagger_ab.txt).
This test compiles the file to gfx950 assembly and checks the order."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = '/opt/rocm/bin/hipcc'


def _phases(body):
    """Collapse an instruction list to [('mfma', n) | ('lds_read', n) | ('lds_write', n) | ('barrier', 1)] runs."""
    out = []
    for ins in body:
        if ins.startswith('v_mfma'):
            kind = 'mfma'
        elif ins.startswith('ds_read_b128'):
            kind = 'lds_read'
        elif ins.startswith('ds_write_b128'):
            kind = 'lds_write'
        elif ins.startswith('s_barrier'):
            kind = 'barrier'
        else:
            continue
        if out and out[-1][0] == kind:
            out[-1][1] += 1
        else:
            out.append([kind, 1])
    return [tuple(p) for p in out]


_ASM = {}


def _assembly(name):
    """gfx950 assembly of sgdml_amd/csrc/<name>.hip (compiled once per test session)."""
    if name not in _ASM:
        tmp = tempfile.mkdtemp()
        try:
            asm = os.path.join(tmp, name + '.s')
            subprocess.check_call([HIPCC, '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only', '-o', asm,
                                   os.path.join(ROOT, 'sgdml_amd', 'csrc', name + '.hip')], stderr=subprocess.DEVNULL)
            _ASM[name] = open(asm).read()
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return _ASM[name]


def _kernel(text, sym):
    start = text.index('\n' + sym + ':')
    return [l.strip() for l in text[start:text.index('s_endpgm', start)].split('\n') if l.strip() and not l.strip().startswith(';')]


def _loads_before_first_full_wait(ins):
    """Global loads issued before the first `s_waitcnt vmcnt(...)` of a kernel."""
    n = 0
    for l in ins:
        if l.startswith('global_load'):
            n += 1
        elif l.startswith('s_waitcnt') and 'vmcnt' in l:
            break
    return n


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_step_chain_prologues_keep_their_loads_in_flight():
    """The 64-column step kernels of the panel chain and the prediction epilogue are latency bound: their global loads must
    be issued in batches, not one per loop iteration with a wait behind each (a loop around a conditional load compiles to
    that; profiles/r04_step_chain.txt, r04_latency_path.txt: 30 % of the small-n factorisation, 18-26 % of the small-batch
    prediction latency)."""
    chol = _assembly('chol')
    # potrf_trsm64_kernel: 16 rows of the diagonal block per wavefront (+ this thread's strip row: 32 16-byte loads)
    assert _loads_before_first_full_wait(_kernel(chol, '_Z19potrf_trsm64_kernelPdS_lillPiS_PKdS_i')) >= 16
    # stand-alone potrf64_kernel: all 64 rows of the block
    assert _loads_before_first_full_wait(_kernel(chol, '_Z14potrf64_kernelPdlilPi')) >= 64
    # panel_trsm_prep_kernel: 16 entries per thread
    assert _loads_before_first_full_wait(_kernel(chol, '_Z22panel_trsm_prep_kernelPKdlPd')) >= 16
    # prediction epilogue: the row-split partials in batches of 8
    pred = _assembly('predict')
    ins = _kernel(pred, '_Z23predict_epilogue_kernelILb1EEvPKdS1_S1_liiiPdS2_')
    runs, cur = [], 0
    for l in ins:
        if l.startswith('global_load'):
            cur += 1
        elif l.startswith('s_waitcnt') and 'vmcnt' in l:
            if cur:
                runs.append(cur)
            cur = 0
    assert runs and max(runs) >= 8 and sum(1 for r in runs if r == 1) <= 2, runs


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_trailing_update_loop_keeps_its_schedule():
    text = _assembly('chol')
    sym = '_Z23gemm_nt_sub_diag_kernelILb1ELb1ELi5EEv8GemmArgs'
    start = text.index('\n' + sym + ':')
    kern = text[start:text.index('.Lfunc_end', start)]  # (the kernel has several exits since it carries the panel-solve strips)
    # inner loops = from an "Inner Loop Header" label to the backward branch to that label
    loops = []
    for m in re.finditer(r'^(\.LBB\d+_\d+):[^\n]*Inner Loop Header[^\n]*\n', kern, flags=re.M):
        label = m.group(1)
        tail = kern[m.end():]
        br = re.search(r'^\s*s_cbranch_\w+\s+' + re.escape(label) + r'\s*$', tail, flags=re.M)
        if not br:
            continue
        body = [l.strip() for l in tail[:br.start()].split('\n') if l.strip() and not l.strip().startswith(';')]
        if sum(1 for l in body if l.startswith('v_mfma_f64_16x16x4')) == 64:
            loops.append(_phases(body))
    # the interior-tile loop of the trailing update and of the second problem carried by a fused launch
    assert len(loops) >= 2, 'k-tile loops with 64 MFMAs: %d' % len(loops)
    for ph in loops:
        seq = [p for p in ph if p[0] != 'lds_write']
        # MFMA runs may be split by s_waitcnt (not listed): merge neighbours
        merged = []
        for kind, n in seq:
            if merged and merged[-1][0] == kind:
                merged[-1] = (kind, merged[-1][1] + n)
            else:
                merged.append((kind, n))
        assert merged == [('mfma', 16), ('lds_read', 8), ('mfma', 32), ('barrier', 1), ('lds_read', 8), ('mfma', 16)], merged
        # the commit of the prefetched tile sits between the 48th MFMA and the barrier
        kinds = [k for k, _ in ph]
        assert kinds.index('lds_write') > kinds.index('lds_read') and kinds.index('lds_write') < kinds.index('barrier'), ph


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_predict_kernel_requests_table_rows_ahead():
    """predict_kernel (batches below the MFMA kernel; the single-geometry latency path): the loads of the next table rows are
    issued at the head of the row loop, before the reductions / sqrt / exp of the current row -- not one row per memory round
    trip (profiles/r04_latency_path.txt)."""
    pred = _assembly('predict')
    sym = '_Z14predict_kernelILi4ELi1EEv8PredArgs'
    start = pred.index('\n' + sym + ':')
    kern = pred[start:pred.index('s_endpgm', start)]
    m = re.search(r'^(\.LBB\d+_\d+):[^\n]*Inner Loop Header[^\n]*\n', kern, flags=re.M)
    assert m, 'row loop not found'
    body = [l.strip() for l in kern[m.end():].split('\n') if l.strip() and not l.strip().startswith(';')]
    first_math = next(i for i, l in enumerate(body) if l.startswith(('v_rsq_f64', 'v_sqrt_f64', 'v_exp_f32', 'v_ldexp_f64')))
    loads_ahead = sum(1 for l in body[:first_math] if l.startswith('global_load'))
    assert loads_ahead >= 16, loads_ahead  # X and J alpha of two rows (KPL = 4: 8 loads per row)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_perm2_kernel_keeps_scratch_out_of_its_store_passes_and_inner_loops():
    """assemble_perm2_kernel (csrc/assemble_perm2.hip) holds 72 of its 168 VGPRs as accumulators; what spills decides its speed:
    a scratch reload between two global stores waits for the stores (vmcnt counts both), a reload in a per-group phase costs a memory
    round trip, and LDS reads the compiler serialises ('read one value, wait, use it') cost 400-700 cycles each under load
    (profiles/r05_assemble_perm2.txt: 125 -> 17 scratch instructions was 18.9 -> 17.5 ms; three-entry trips for every row type spill and
    were 16.4 -> 19.2 ms).  Checked on the compiled production kernel: few scratch instructions overall, at most three behind the first
    row store (today: two accumulator tuples reloaded for a later pass, the table pointer of the image request), LDS-only barriers between the row passes, and no chain of three or more single-read LDS round trips."""
    ins = _kernel(_assembly('assemble_perm2'), '_Z21assemble_perm2_kernelILb0ELb0ELb0EEv9Perm2Args')  # the staged-rows variant (asm.perm2_direct = 0)
    scratch = [k for k, l in enumerate(ins) if l.startswith('scratch_')]
    assert len(scratch) <= 45, len(scratch)
    stores = [k for k, l in enumerate(ins) if l.startswith('global_store')]
    assert stores, 'no row stores found'
    assert len([k for k in scratch if k > stores[0]]) <= 3, 'scratch traffic between / behind the row stores'
    # the barriers between the row passes must not drain the stores: the instruction in front of an s_barrier behind the first
    # store is `s_waitcnt lgkmcnt(0)` (lds_barrier()), never a vmcnt wait
    for k, l in enumerate(ins):
        if l.startswith('s_barrier') and k > stores[0]:
            assert ins[k - 1].startswith('s_waitcnt') and 'vmcnt' not in ins[k - 1], (k, ins[k - 1])
    # serialised LDS round trips: runs of `one LDS read -> s_waitcnt lgkmcnt(0)`
    run, n_reads, worst = 0, 0, 0
    for l in ins:
        if l.startswith('ds_read') or l.startswith('ds_bpermute'):
            n_reads += 1
        elif l.startswith('s_waitcnt') and 'lgkmcnt(0)' in l:
            run = run + 1 if n_reads == 1 else 0
            worst = max(worst, run)
            n_reads = 0
    assert worst < 3, worst
