"""Step-level Gantt from a rocpd kernel trace: per stream busy time, union busy, idle gaps, and who runs alone."""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.stream_id, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
print(len(rows), 'dispatches')
# steps: find the optimizer kernel (sgd) as step delimiter
sgd = [i for i, r in enumerate(rows) if 'sgd' in r[0].lower()]
print('sgd launches', len(sgd))
ends = [i for j, i in enumerate(sgd) if j + 1 == len(sgd) or sgd[j + 1] - i > 50]      # last optimizer launch of each step
print('steps', len(ends))
lo, hi = ends[-2] + 1, ends[-1] + 1           # the last full step
step = rows[lo:hi]
t0, t1 = step[0][1], max(r[2] for r in step)
print(f'step wall {(t1 - t0) / 1e6:.2f} ms, {len(step)} dispatches')
def cls(n):
    n = n.lower()
    if 'wgrad' in n or 'reduce_slabs' in n: return 'wgrad'
    if 'igemm' in n or 'wino' in n or 'conv_stem' in n or 'conv_co8' in n: return 'conv'
    if 'bn_' in n: return 'bn'
    return 'other'
by_q = collections.defaultdict(float)
for n, s, e, st, q in step: by_q[(q, st)] += (e - s) / 1e6
print('busy ms per (queue, stream):', {k: round(v, 2) for k, v in by_q.items()})
by_c = collections.defaultdict(float)
for n, s, e, st, q in step: by_c[cls(n)] += (e - s) / 1e6
print('kernel ms by class:', {k: round(v, 2) for k, v in by_c.items()}, 'sum', round(sum(by_c.values()), 2))
# sweep: time with k kernels running, and time where only non-MFMA kernels run
ev = []
for n, s, e, st, q in step:
    c = cls(n); ev.append((s, 1, c)); ev.append((e, -1, c))
ev.sort()
act = collections.Counter(); last = t0; hist = collections.defaultdict(float)
for t, d, c in ev:
    key = ('idle' if sum(act.values()) == 0 else '+'.join(sorted(k for k, v in act.items() if v > 0)))
    hist[key] += (t - last) / 1e6; last = t
    act[c] += d
for k, v in sorted(hist.items(), key=lambda kv: -kv[1]): print(f'  {k:28s} {v:7.2f} ms')
# biggest idle gaps
gaps = []; cur_end = t0
for n, s, e, st, q in sorted(step, key=lambda r: r[1]):
    if s > cur_end: gaps.append((s - cur_end, cur_end - t0, n))
    cur_end = max(cur_end, e)
gaps.sort(reverse=True)
print('idle total %.2f ms in %d gaps; largest:' % (sum(g[0] for g in gaps) / 1e6, len(gaps)))
for g in gaps[:8]: print(f'   {g[0] / 1e3:8.1f} us at +{g[1] / 1e6:6.2f} ms before {g[2][:70]}')
