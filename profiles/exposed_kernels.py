"""Which kernels are EXPOSED in the multi-stream step: time during which no matrix-core kernel (conv / wgrad class) runs,
attributed to the kernels that run then; plus the timeline of exposed intervals.  Input: rocpd db of the 3-stream trace."""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.stream_id, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
sgd = [i for i, r in enumerate(rows) if 'sgd' in r[0].lower()]
ends = [i for j, i in enumerate(sgd) if j + 1 == len(sgd) or sgd[j + 1] - i > 50]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t0, t1 = step[0][1], max(r[2] for r in step)


def short(n):
    m = re.match(r'_ZN5dynmm\d+([a-z0-9_]+)_kernel(?:I((?:L[ib]\d+E)+)E)?', n)
    s = m.group(1) if m else n[:40]
    if m and m.group(2):
        s += '<' + re.sub(r'L[ib](\d+)E', r'\1,', m.group(2)).rstrip(',') + '>'
    return s


def mfma(n):
    n = n.lower()
    return ('wgrad' in n and 'co8' not in n) or 'igemm' in n or 'wino' in n or 'conv_stem' in n


ev = []
for i, (n, s, e, st, q) in enumerate(step):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set(); last = t0
exposed = collections.defaultdict(float); total = collections.defaultdict(float); cnt = collections.Counter()
segs = []
for n, s, e, st, q in step:
    total[short(n)] += (e - s) / 1e3; cnt[short(n)] += 1
for t, d, i in ev:
    dt = (t - last) / 1e3
    if dt > 0:
        names = [step[j][0] for j in active]
        if not any(mfma(x) for x in names):
            if names:
                for x in names: exposed[short(x)] += dt / len(names)
            else:
                exposed['(idle)'] += dt
            segs.append(((last - t0) / 1e6, dt, '+'.join(sorted(short(x) for x in names)) or '(idle)'))
    last = t
    if d > 0: active.add(i)
    else: active.discard(i)
print(f'step wall {(t1 - t0) / 1e6:.2f} ms; exposed (no matrix-core kernel running) {sum(exposed.values()) / 1e3:.2f} ms')
print('kernel | launches | total us | exposed us')
for k, v in sorted(exposed.items(), key=lambda kv: -kv[1])[:45]:
    print(f'{k:60s} {cnt.get(k, 0):4d} {total.get(k, 0):9.1f} {v:9.1f}')
# merge adjacent segments into windows: exposed time per ms of the step
per_ms = collections.defaultdict(float)
for at, dt, _ in segs: per_ms[int(at)] += dt
print('exposed us per ms of the step:')
print(' '.join(f'{int(per_ms.get(i, 0)):d}' for i in range(int((t1 - t0) / 1e6) + 1)))
print('exposed segments >= 15 us, in time order:')
for at, dt, names in segs:
    if dt >= 15: print(f'  +{at:6.2f} ms {dt:7.1f} us  {names[:150]}')
