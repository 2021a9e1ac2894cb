cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tail_pmc
rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
python $R/scratch/tail_micro.py 5
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -o p -- python $R/scratch/tail_micro.py 1 > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES --kernel-trace --output-format csv -d $O/b -o p -- python $R/scratch/tail_micro.py 1 > $O/b.log 2>&1
python - <<'PY'
import csv, glob, collections, os
for d in ('a', 'b'):
    for f in glob.glob(os.environ.get('GRAFT_REPO_ROOT', '.') + f'/gpurun_out/tail_pmc/{d}/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:40]
            if 'up2ce' not in k: continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, v in agg.items():
            print(k, {a: f'{b:.3g}' for a, b in v.items()})
PY
