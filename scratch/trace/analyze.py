"""TF/s over time + workgroup lifetime statistics from gpurun_out/trace.npz (scratch/trace/run_trace.py)."""
import sys
import numpy as np
d = np.load(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/trace.npz')
tile = {64: 64 * 128, 128: 128 * 64, 256: 128 * 64, 512: 128 * 64}
for k in d.files:
    t = d[k]
    C, hw, kk = k.split('_')
    C = int(C); KH, KW = map(int, kk.split('x'))
    t0 = t[:, 0].min()
    rel = (t[:, :4] - t0) / 100.0
    T = rel[:, 3].max()
    bins = np.arange(0, T + 5, 5.0)
    prog = np.zeros(len(bins) - 1)
    for a, b in zip(rel[:, 1], rel[:, 2]):
        lo = np.clip(bins[:-1], a, b); hi = np.clip(bins[1:], a, b)
        prog += (hi - lo) / max(b - a, 1e-9)
    flop = tile[C] * C * KH * KW * 2
    clk = ((t[:, 5] - t[:, 4]) / ((t[:, 3] - t[:, 0]) / 100.0)).mean()
    print(f'{k}: {len(t)} blocks, span {T:.1f} us, mean clock {clk:.0f} MHz, avg {len(t) * flop / T / 1e6:.1f} TF/s')
    for nm, a, b in (('prologue', 0, 1), ('loop', 1, 2), ('epilogue', 2, 3)):
        x = rel[:, b] - rel[:, a]
        print(f'   {nm:9s} mean {x.mean():6.2f} p10 {np.percentile(x, 10):6.2f} p50 {np.median(x):6.2f} p90 {np.percentile(x, 90):6.2f} max {x.max():6.2f}')
    print('   TF/s per 5 us:', ' '.join(f'{v:.0f}' for v in prog * flop / 5e-6 / 1e12))
