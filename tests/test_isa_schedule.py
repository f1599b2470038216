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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_trailing_update_loop_keeps_its_schedule():
    tmp = tempfile.mkdtemp()
    try:
        asm = os.path.join(tmp, 'chol.s')
        subprocess.check_call([HIPCC, '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only', '-o', asm,
                               os.path.join(ROOT, 'sgdml_amd', 'csrc', 'chol.hip')], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    sym = '_Z23gemm_nt_sub_diag_kernelILb1ELb1ELi5EEv8GemmArgs'
    start = text.index('\n' + sym + ':')
    kern = text[start:text.index('s_endpgm', start)]
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
