"""Scan the compiled kernels for memory round trips that the compiler serialised (round 4; no GPU needed).

  python tools/isa_scan.py [file.hip ...]        (default: every sgdml_amd/csrc/*.hip)

Compiles each source to gfx950 assembly (hipcc -S --cuda-device-only) and lists the basic blocks that belong to a loop, contain
at most four global loads and wait for ALL of them (s_waitcnt vmcnt(0)) inside the block: one memory round trip per loop
iteration.  Typical sources: a loop around a conditional load (`if (ok) v = p[i];`), a run-time reduction loop over partial
results, row-by-row prefetch-less table walks.  In a throughput kernel with several workgroups per CU this is harmless (the
strip-assembly prologue: measured neutral); in a latency-bound kernel -- one workgroup, one wavefront, a dependent chain of small
launches -- every hit is ~1 us per iteration: the panel step chain (profiles/r04_step_chain.txt) and the prediction latency path
(profiles/r04_latency_path.txt) were found this way."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def scan(asm_text):
    kern, cur, blocks = None, None, []
    for line in asm_text.split('\n'):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            kern = m.group(1)
        m = re.match(r'^(\.LBB\d+_\d+):(.*)', line)
        if m:
            cur = {'kernel': kern, 'label': m.group(1), 'loop': 'Loop' in m.group(2), 'loads': 0, 'wait0': 0, 'n': 0}
            blocks.append(cur)
            continue
        s = line.strip()
        if cur is None or not s or s.startswith(';'):
            continue
        cur['n'] += 1
        if s.startswith(('global_load', 'flat_load', 'buffer_load', 'scratch_load')):
            cur['loads'] += 1
        if s.startswith('s_waitcnt') and 'vmcnt(0)' in s:
            cur['wait0'] += 1
    return [b for b in blocks if b['loop'] and 0 < b['loads'] <= 4 and b['wait0'] > 0 and b['n'] < 80]


def main(argv):
    files = argv or sorted(glob.glob(os.path.join(ROOT, 'sgdml_amd', 'csrc', '*.hip')))
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            asm = os.path.join(tmp, os.path.basename(f) + '.s')
            subprocess.check_call([HIPCC, '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only', '-o', asm, f],
                                  stderr=subprocess.DEVNULL)
            hits = scan(open(asm).read())
            per_kernel = {}
            for b in hits:
                per_kernel.setdefault(b['kernel'], []).append(b)
            for k, bs in sorted(per_kernel.items()):
                print('%-22s %-72s %3d block(s), e.g. %s: %d load(s), %d instruction(s)' % (
                    os.path.basename(f), (k or '?')[:72], len(bs), bs[0]['label'], bs[0]['loads'], bs[0]['n']))


if __name__ == '__main__':
    main(sys.argv[1:])
