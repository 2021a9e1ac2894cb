# HBM traffic of ONE ungrouped weight gradient: FETCH_SIZE / WRITE_SIZE per launch for a single C=128 3x1
# convolution at batch 32 (algorithmic: x 78.6 MB + dy 78.6 MB read, 33 MB of split-K slabs written), and for the forward /
# input-gradient kernels on the same shape (78.6 MB read + 78.6 MB written, + 78.6 MB mask for the dgrad).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wgrad_pmc
rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o p -- python $R/scratch/conv_micro.py 128 60 80 all 3 > $O/$c.log 2>&1
done
cd $R
python - <<'PY'
import csv, collections, glob, os
O = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'wgrad_pmc')
res = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(os.path.join(O, c, '**', '*counter_collection.csv'), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == c:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        res[k][c] = (sum(v) / len(v), len(v))
lines = ['| kernel | launches | FETCH_SIZE MiB/launch raw | x2 (16 B/lane correction) | WRITE_SIZE MiB/launch |', '|---|---|---|---|---|']
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get('FETCH_SIZE', (0, 0))[0]):
    if 'dynmm' not in k:
        continue
    fe, n = d.get('FETCH_SIZE', (0, 0)); wr, _ = d.get('WRITE_SIZE', (0, 0))
    lines.append(f'| `{k[:90]}` | {n} | {fe / 1024:.1f} | {2 * fe / 1024:.1f} | {wr / 1024:.1f} |')
open(os.path.join(O, 'summary.md'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
