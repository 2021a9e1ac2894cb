cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc1
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq1 -o p -- python $R/scratch/conv_micro.py 128 60 80 all 3 > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/sq2 -o p -- python $R/scratch/conv_micro.py 128 60 80 all 3 > $O/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/f -o p -- python $R/scratch/conv_micro.py 128 60 80 all 3 > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/w -o p -- python $R/scratch/conv_micro.py 128 60 80 all 3 > $O/w.log 2>&1
find $O -name "*.csv" | head; tail -2 $O/sq1.log
