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


def scan(asm):
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
        if n >= 4:
            out.append((n, s.count('L'), k))
    return sorted(out, reverse=True)


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            stem = os.path.splitext(os.path.basename(f))[0]
            subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', f'-I{ROOT}/include', f'-I{CSRC}',
                            '-Wno-inline-asm', '--save-temps=obj', '-c', os.path.abspath(f), '-o', os.path.join(tmp, stem + '.o')],
                           cwd=tmp, check=True, stderr=subprocess.DEVNULL)
            for n, nl, k in scan(os.path.join(tmp, f'{stem}-hip-amdgcn-amd-amdhsa-gfx950.s')):
                print(f'{n:3d} of {nl:3d} loads wait alone   {stem:20s} {k[:120]}')


if __name__ == '__main__':
    main()
