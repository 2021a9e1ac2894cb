# round-5 first diagnosis: bound-splitting experiments + SQ counters of the isolated Winograd kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python scratch/r5/wino_exp.py > $O/wino_exp.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES" \
            "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- python $R/scratch/r5/wino_iso.py > $O/pmc_$tag.log 2>&1
done
cd $R
python - <<'PY' > $O/pmc_iso.md 2>&1
import csv, glob, collections, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r5'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'wino' not in r['Kernel_Name'] or 'pack' in r['Kernel_Name']: continue
        agg[(r['Kernel_Name'][:70], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k)
    for c, vs in sorted(d.items()):
        print(f'   {c:28s} {sum(vs) / len(vs):16.0f}  (n={len(vs)})')
PY
rm -rf $O/pmc_*/*/*.db
cat $O/wino_exp.log
