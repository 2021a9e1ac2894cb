"""Which kernels hold loads that the compiler serialised behind per-element predicates?

    python scratch/r6/serial_loads.py [file.hip ...]      (default: every dynmm_amd/csrc/*.hip; no GPU needed)

`v = ok ? p[i] : 0.f` inside an unrolled loop compiles to a divergent branch per element whose load is followed by its own
`s_waitcnt vmcnt(0)`: N dependent round trips instead of N loads in flight (round 6: LayerNorm, attention and the loss head's
prologue).  The scan saves the gfx950 ISA of each file and counts, per kernel, the loads whose NEXT memory event is a full
vmcnt(0) wait with no other load in between.  It reads static code: a count says where to look (a rarely taken generic path
scores like a hot loop), the fix is an unconditional load on a clamped address with the value selected afterwards."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'dynmm_amd', 'csrc')


def scan(asm, floor=4):
    res, name, ev = {}, None, None
    for line in open(asm):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            name, ev = m.group(1), []
            res[name] = ev
            continue
        if name is None:
            continue
        if 'global_load' in line or 'buffer_load' in line:
            ev.append('L')
        elif re.search(r's_waitcnt.*vmcnt\(0\)', line):
            ev.append('W')
        elif 's_endpgm' in line:
            name = None
    out = []
    for k, ev in res.items():
        s = ''.join(ev)
        n = len(re.findall(r'(?<!L)LW', s)) + (1 if s.startswith('LW') else 0)
        if n >= floor:
            out.append((n, s.count('L'), k))
    return sorted(out, reverse=True)


def compile_and_scan(hip_file, floor=0):
    """[(lone-wait loads, loads, mangled kernel name)] of one kernel file (compiles it for gfx950; no GPU needed)"""
    stem = os.path.splitext(os.path.basename(hip_file))[0]
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', f'-I{ROOT}/include', f'-I{CSRC}',
                        '-Wno-inline-asm', '--save-temps=obj', '-c', os.path.abspath(hip_file), '-o', os.path.join(tmp, stem + '.o')],
                       cwd=tmp, check=True, stderr=subprocess.DEVNULL)
        return scan(os.path.join(tmp, f'{stem}-hip-amdgcn-amd-amdhsa-gfx950.s'), floor)


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    for f in files:
        stem = os.path.splitext(os.path.basename(f))[0]
        for n, nl, k in compile_and_scan(f, 4):
            print(f'{n:3d} of {nl:3d} loads wait alone   {stem:20s} {k[:120]}')


if __name__ == '__main__':
    main()
