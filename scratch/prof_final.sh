cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_final
rm -rf $O; mkdir -p $O
# (1) kernel trace of the roofline-leg configuration (single stream: isolated per-kernel durations)
rocprofv3 --kernel-trace --stats -d $O/single -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --single-stream > $O/single_stdout.log 2>&1
# (2) kernel trace of the default 3-stream run
rocprofv3 --kernel-trace --stats -d $O/multi -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/multi_stdout.log 2>&1
# (3) PMC: HBM traffic per launch, separate passes (FETCH_SIZE 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --single-stream > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --single-stream > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --single-stream > $O/pmc_sq.log 2>&1
ls -la $O $O/*/ | head -40
tail -1 $O/single_stdout.log | cut -c1-200
cd $R
python profiles/summarize_rocpd.py $O/single/bench_results.db $O/single.md > /dev/null
python profiles/summarize_rocpd.py $O/multi/bench_results.db $O/multi.md > /dev/null
python profiles/summarize_pmc.py $O/pmc_fetch/p_counter_collection.csv $O/pmc_write/p_counter_collection.csv $O/pmc_sq/p_counter_collection.csv $O/pmc.md $O/pmc.json conv_igemm > $O/pmc_summary.log 2>&1
