"""A static gate on the built kernels' resources (no GPU): occupancy the design counts on, and no register spills.

Reads the gfx950 code objects out of dynmm_amd/csrc/build/*.o (llvm-objcopy -> clang-offload-bundler -> llvm-readelf --notes) and
checks, per kernel, the VGPR count, the static LDS and the scratch bytes the compiler settled on:
  * the kernels whose speed DESIGN.md ties to a workgroup count per CU still fit that count (512 VGPRs per SIMD lane, 160 KB of
    LDS per CU): the F(4,3) input-gradient kernel at three workgroups per CU (round 6: 126 -> 92 us per launch came from exactly
    that), the vertical F(2,3) forward at four, the four-wave attention kernels and the LayerNorm kernels at four waves per SIMD;
  * nothing spills to scratch except the kernels on the allow-list, and those not more than they do today."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, 'dynmm_amd', 'csrc', 'build')
LLVM = '/opt/rocm/lib/llvm/bin/'
TOOLS = [LLVM + t for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-readelf')]

# kernel-name fragment -> (max VGPRs, max static LDS bytes): workgroups of 256 threads, W workgroups per CU <=> VGPRs <= 512 / W
OCCUPANCY = {
    'conv_wino43_kernelE': (170, 160 * 1024 // 3),                                   # three workgroups per CU
    'conv_wino_kernelILi64ELi1ELb0ELb0ELb0ELb0ELb1ELi0E': (128, 160 * 1024 // 4),    # vertical F(2,3) forward, 2-slot ring: four
    'mha_fwd_kernelILi24ELi4E': (128, 0), 'mha_bwd_kernelILi24ELi4E': (128, 0),      # (their LDS is dynamic: 31 / 41 KB at T = 50)
    'ln_fwd_kernelILi1E': (128, 2048), 'ln_bwd_dx_kernelILi1E': (128, 2048),
    'up2ce_bwd_kernel': (256, 160 * 1024 // 3), 'up2ce_fwd_kernel': (256, 160 * 1024 // 3),
}
# kernel-name fragment -> scratch bytes it may use (everything else: none)
SCRATCH_OK = {'conv_wgrad_kernelILi64ELi192ELi32ELi96ELb1ELb0E': 8, 'conv_stem_fwd_kernelILi1ELb1E': 32, 'ffn_kernelILi': 256}


def _resources(obj):
    with tempfile.TemporaryDirectory() as t:
        fat, dev = os.path.join(t, 'fat.bin'), os.path.join(t, 'dev.co')
        subprocess.run([TOOLS[0], '--dump-section', f'.hip_fatbin={fat}', obj], check=True)
        subprocess.run([TOOLS[1], '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--input={fat}', f'--output={dev}',
                        '--unbundle'], check=True)
        txt = subprocess.run([TOOLS[2], '--notes', dev], check=True, capture_output=True, text=True).stdout
    out, cur = {}, {}
    for line in txt.splitlines():
        m = re.match(r'\s+(?:- )?\.(\w+):\s+(\S+)', line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == 'agpr_count':                       # first key of a kernel's record (keys are sorted)
            if cur.get('name'):
                out[cur['name']] = cur
            cur = {}
        if k in ('group_segment_fixed_size', 'private_segment_fixed_size', 'vgpr_count'):
            cur[k] = int(v)
        elif k == 'name':
            cur['name'] = v
    if cur.get('name'):
        out[cur['name']] = cur
    return out


@pytest.mark.skipif(not all(os.path.exists(t) for t in TOOLS) or shutil.which('make') is None, reason='needs the ROCm LLVM tools')
def test_kernel_occupancy_and_spills():
    if not glob.glob(os.path.join(BUILD, '*.o')):
        subprocess.run(['make', '-C', os.path.dirname(BUILD), '-j8'], check=True, stdout=subprocess.DEVNULL)
    kernels = {}
    for obj in sorted(glob.glob(os.path.join(BUILD, '*.o'))):
        if b'.hip_fatbin' in open(obj, 'rb').read():
            kernels.update(_resources(obj))
    assert len(kernels) > 150, len(kernels)
    for frag, (max_vgpr, max_lds) in OCCUPANCY.items():
        hits = {n: r for n, r in kernels.items() if frag in n}
        assert hits, f'{frag}: kernel not found (renamed? update the table)'
        for n, r in hits.items():
            assert r['vgpr_count'] <= max_vgpr, f'{n}: {r["vgpr_count"]} VGPRs (budget {max_vgpr})'
            assert r['group_segment_fixed_size'] <= max_lds, f'{n}: {r["group_segment_fixed_size"]} B of static LDS (budget {max_lds})'
    for n, r in kernels.items():
        allowed = max([b for frag, b in SCRATCH_OK.items() if frag in n] or [0])
        assert r['private_segment_fixed_size'] <= allowed, f'{n}: {r["private_segment_fixed_size"]} B of scratch (allowed {allowed})'
