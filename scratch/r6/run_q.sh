#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6q; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_blocks.py -x -q -m gpu -k "wgrad or grouped or block or encoder or decoder" > $O/pytest.log 2>&1
tail -n 3 $O/pytest.log
DYNMM_BENCH_SHAPES=$O/shapes.txt timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/bench.log 2>&1
grep "3, 1, 1, 1" $O/shapes.txt | grep wgrad | head -8
bash scratch/r6/ab_multi.sh r6q - scratch/r6/libdynmm_strips3b.so
bash scratch/r6/ab_multi.sh r6q - scratch/r6/libdynmm_strips3b.so
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing --single-stream > $O/pmc_$c.log 2>&1
  python - <<PY
import csv, collections
rows = list(csv.DictReader(open('$O/pmc_$c/p_counter_collection.csv')))
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if 'wgrad_wino_vt' in r['Kernel_Name'] and r['Counter_Name'] == '$c':
        k = r['Kernel_Name'][:60]
        acc[k][0] += 1; acc[k][1] += float(r['Counter_Value'])
for k, (n, v) in acc.items():
    print('$c', k, n, 'launches', round(v / n), 'KiB per launch')
PY
  rm -rf $O/pmc_$c
done
