cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/stem_pmc
rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -o p -- python $R/scratch/gate_micro.py > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/b -o p -- python $R/scratch/gate_micro.py > $O/b.log 2>&1
python - <<'PY'
import csv, glob, collections, os
for d in ('a', 'b'):
    for f in glob.glob(os.environ.get('GRAFT_REPO_ROOT', '.') + f'/gpurun_out/stem_pmc/{d}/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:44]
            if 'stem' not in k and 'co8' not in k: continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, v in agg.items():
            print(k, {a: f'{b:.3g}' for a, b in v.items()})
PY
