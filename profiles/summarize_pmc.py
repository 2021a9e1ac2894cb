"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (mean per launch) and emit the
HBM-traffic record bench.py reads (profiles/pmc_dominant_kernel.json).

FETCH_SIZE / WRITE_SIZE are reported in KiB.  gfx950 correction (MI355X_MICROARCH.md, section HBM):
FETCH_SIZE under-reports a wide (16 B/lane) coalesced read stream by exactly 2x; the implicit-GEMM
gathers are 4 B/lane (uncalibrated width), so both the raw and the x2 figure are recorded and the
JSON carries the raw one plus the bound.  WRITE_SIZE matched the algorithmic output bytes exactly in
a calibration launch (76,800 KiB reported vs 78.6 MB written)."""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} | {'launches': len(next(iter(d.values())))} for k, d in agg.items()}


def main(fetch_csv, write_csv, sq_csv, out_md, out_json, dominant_substr, steps_profiled=4, sq2_csv=None):
    steps_profiled = int(steps_profiled)
    f, w, q = per_kernel(fetch_csv), per_kernel(write_csv), per_kernel(sq_csv)
    q2 = per_kernel(sq2_csv) if sq2_csv else {}
    names = sorted(f, key=lambda k: -f[k].get('FETCH_SIZE', 0) * f[k]['launches'])
    lines = ['| kernel | launches | FETCH_SIZE KiB/launch (raw) | WRITE_SIZE KiB/launch | MFMA busy / (SIMDs x GUI cycles) | WAIT_INST_ANY / WAVE_CYCLES | waves per SIMD (mean) | VALU inst per MFMA inst | LDS inst per MFMA inst | LDS bank-conflict / LDS active cycles | WAIT_INST_LDS / WAVE_CYCLES |',
             '|---|---|---|---|---|---|---|---|---|---|---|']
    for k in names[:25]:
        sq = q.get(k, {})
        util = ''
        if sq.get('GRBM_GUI_ACTIVE'):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
            util = f"{sq.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * sq['GRBM_GUI_ACTIVE'] / 8):.2f}"
        wi = f"{sq.get('SQ_WAIT_INST_ANY', 0) / sq['SQ_WAVE_CYCLES']:.2f}" if sq.get('SQ_WAVE_CYCLES') else ''
        short = k if len(k) < 120 else k[:117] + '...'
        s2 = q2.get(k, {})
        occ = f"{4 * sq['SQ_WAVE_CYCLES'] / (1024 * sq['GRBM_GUI_ACTIVE'] / 8):.2f}" if sq.get('GRBM_GUI_ACTIVE') and sq.get('SQ_WAVE_CYCLES') else ''
        nm = sq.get('SQ_INSTS_MFMA', 0)
        valu = f"{(s2.get('SQ_INSTS_VALU', 0) - nm) / nm:.1f}" if nm and s2 else ''
        ldsi = f"{s2.get('SQ_INSTS_LDS', 0) / nm:.2f}" if nm and s2 else ''
        conf = f"{s2.get('SQ_LDS_BANK_CONFLICT', 0) / s2['SQ_LDS_IDX_ACTIVE']:.2f}" if s2.get('SQ_LDS_IDX_ACTIVE') else ''
        wl = f"{s2.get('SQ_WAIT_INST_LDS', 0) / sq['SQ_WAVE_CYCLES']:.3f}" if s2 and sq.get('SQ_WAVE_CYCLES') else ''
        lines.append(f"| `{short}` | {f[k]['launches']} | {f[k].get('FETCH_SIZE', 0):.0f} | {w.get(k, {}).get('WRITE_SIZE', 0):.0f} | {util} | {wi} | {occ} | {valu} | {ldsi} | {conf} | {wl} |")
    open(out_md, 'w').write('\n'.join(lines) + '\n')
    # bench.py kernel-variant name -> demangled template instance
    # (substring of the demangled instance, FETCH_SIZE correction): the x2 correction of the guide holds for 16 B/lane
    # streaming reads — the vectorised weight-gradient kernel (conv_wgrad_v4) loads dwordx4; the implicit-GEMM
    # kernels gather 4 B/lane (uncalibrated): raw value, x2 recorded as the upper bound
    variants = {'conv_igemm_fwd<128x64>': ('conv_igemm_kernel<128, 64, 64, 32, false, 0>', 1),
                'conv_igemm_dgrad<128x64>': ('conv_igemm_kernel<128, 64, 64, 32, true, 0>', 1),
                'conv_igemm_fwd<64x128>': ('conv_igemm_kernel<64, 128, 32, 64, false, 0>', 1),
                'conv_igemm_dgrad<64x128>': ('conv_igemm_kernel<64, 128, 32, 64, true, 0>', 1),
                'conv_igemm_fwd<128x32>': ('conv_igemm_kernel<128, 32, 32, 32, false, 0>', 1),
                'conv_igemm_dgrad<128x32>': ('conv_igemm_kernel<128, 32, 32, 32, true, 0>', 1),
                'conv_wgrad_v4<co128>': ('conv_wgrad_v4_kernel<128, 128, 1>', 2),
                'conv_wgrad_v6<co128,1x3>': ('conv_wgrad_v6_kernel<2, 3, 2, false>', 2),
                'conv_wgrad_v6<co128,3x3>': ('conv_wgrad_v6_kernel<2, 3, 2, true>', 2),
                'conv_wgrad_v6<co128,3x1>': ('conv_wgrad_wino_vt_kernel<2>', 2),
                'conv_wgrad_v6<co64,1x3>': ('conv_wgrad_v6_kernel<1, 3, 3, false>', 2),
                'conv_wgrad_v6<co64,3x3>': ('conv_wgrad_v6_kernel<1, 3, 3, true>', 2),
                'conv_wgrad_v6<co64,3x1>': ('conv_wgrad_wino_vt_kernel<1>', 2),
                'conv_wgrad_s2<co128,3x1>': ('conv_wgrad_s2_kernel<2, true, 2>', 2),
                'conv_wgrad_s2<co128,1x3>': ('conv_wgrad_s2_kernel<2, false, 2>', 2),
                # Winograd kernels (conv_wino.hip / conv_wino43.hip / conv_wino2d.hip; 16 B/lane direct-to-LDS streams).  keys =
                # bench.py's kernel_instance(); conv_wino_kernel's template arguments are <TCO, MCO, VERT, DGRAD, S2, TAIL, STATS,
                # BNRED>, conv_wino2d_kernel's <DGRAD, TAIL, STATS>: a label that several compiled instances serve (the training
                # forward before a BatchNorm runs the STATS instance, the input gradient behind relu(BN(.)) the BNRED one, the
                # 40-class conv_out the TAIL one) lists them all — the record is their launch-weighted mean
                'conv_wino_fwd<horizontal>': (['conv_wino_kernel<64, 1, false, false, false, false, false, 0>',
                                               'conv_wino_kernel<64, 1, false, false, false, false, true, 0>'], 2),
                'conv_wino_fwd<vertical>': (['conv_wino_kernel<64, 1, true, false, false, false, false, 0>'], 2),
                'conv_wino_dgrad<horizontal>': (['conv_wino_kernel<64, 1, false, true, false, false, false, 0>'], 2),
                'conv_wino_dgrad<vertical>': (['conv_wino_kernel<64, 1, true, true, false, false, false, 0>',
                                               'conv_wino_kernel<64, 1, true, true, false, false, false, 1>',
                                               'conv_wino_kernel<64, 1, true, true, false, false, false, 2>'], 2),
                'conv_wino_dgrad<horizontal,s2>': (['conv_wino_kernel<64, 1, false, true, true, false, false, 0>'], 2),
                'conv_wino_dgrad<vertical,s2>': (['conv_wino_kernel<64, 1, true, true, true, false, false, 0>'], 2),
                'conv_wino43_dgrad<horizontal>': (['conv_wino43_kernel('], 2),
                'conv_wino2d_fwd<3x3>': (['conv_wino2d_kernel<false, false, false>', 'conv_wino2d_kernel<false, true, false>',
                                          'conv_wino2d_kernel<false, false, true>'], 2),
                'conv_wino2d_dgrad<3x3>': (['conv_wino2d_kernel<true, false, false>'], 2),
                # the operand-ring kernels stream 16 B/lane (global_load_lds_dwordx4): the guide's x2 correction applies
                'conv_igemm_v5_fwd<128x64,kw3>': ('conv_igemm_v5_kernel<128, 64, 64, 32, 3, false', 2),
                'conv_igemm_v5_fwd<128x64,kw1>': ('conv_igemm_v5_kernel<128, 64, 64, 32, 1, false', 2),
                'conv_igemm_v5_dgrad<128x64,kw3>': ('conv_igemm_v5_kernel<128, 64, 64, 32, 3, true', 2),
                'conv_igemm_v5_dgrad<128x64,kw1>': ('conv_igemm_v5_kernel<128, 64, 64, 32, 1, true', 2),
                'conv_igemm_v5_fwd<64x128,kw3>': ('conv_igemm_v5_kernel<64, 128, 32, 64, 3, false', 2),
                'conv_igemm_v5_fwd<64x128,kw1>': ('conv_igemm_v5_kernel<64, 128, 32, 64, 1, false', 2),
                'conv_igemm_v5_dgrad<64x128,kw3>': ('conv_igemm_v5_kernel<64, 128, 32, 64, 3, true', 2),
                'conv_igemm_v5_dgrad<64x128,kw1>': ('conv_igemm_v5_kernel<64, 128, 32, 64, 1, true', 2),
                'conv_wgrad<co64>': ('conv_wgrad_kernel<64, 192, 32, 96, false, true>', 1)}
    out = {'note': 'mean per launch over one bench step (batch 32), KiB->bytes: hbm_bytes_per_launch = fetch_factor * '
                   'FETCH_SIZE + WRITE_SIZE.  fetch_factor 2 = the guide\'s gfx950 correction for 16 B/lane streaming '
                   'reads (conv_wgrad_v4 loads dwordx4); 1 = raw (4 B/lane gathers: uncalibrated width, '
                   '*_if_fetch_x2 is the upper bound).  The slab reduction kernel that follows each split wgrad '
                   'launch is listed separately (reduce_slabs*).'}
    for bench_name, (sub, factor) in variants.items():
        subs = sub if isinstance(sub, list) else [sub]
        ks = [k for k in f if any(x in k for x in subs)]
        if not ks:
            continue
        nl = sum(f[k]['launches'] for k in ks)
        fe = sum(f[k].get('FETCH_SIZE', 0) * f[k]['launches'] for k in ks) / nl
        wr = sum(w.get(k, {}).get('WRITE_SIZE', 0) * f[k]['launches'] for k in ks) / nl
        k = ' + '.join(ks)
        out[bench_name] = {'kernel': k, 'launches_profiled': nl,
                           'launches_per_step': nl / steps_profiled, 'fetch_kib_per_launch_raw': fe,
                           'write_kib_per_launch': wr, 'fetch_factor': factor,
                           'hbm_bytes_per_launch': (factor * fe + wr) * 1024,
                           'hbm_bytes_per_launch_if_fetch_x2': (2 * fe + wr) * 1024}
    for sub in ('reduce_slabs_perm_kernel', 'reduce_slabs_kernel<4>'):
        ks = [k for k in f if sub in k]
        if ks:
            k = ks[0]
            out[sub] = {'kernel': k, 'launches_profiled': f[k]['launches'], 'fetch_kib_per_launch_raw': f[k].get('FETCH_SIZE', 0),
                        'write_kib_per_launch': w.get(k, {}).get('WRITE_SIZE', 0)}
    json.dump(out, open(out_json, 'w'), indent=1)
    print(json.dumps(out, indent=1)[:1500])
    print('\n'.join(lines[:12]))


if __name__ == '__main__':
    main(*sys.argv[1:9])
