# Round-6 profiling recipe (run on the GPU box through gpurun): rocprofv3 kernel traces of the bench command in the
# roofline-leg configuration (single stream) and the default 3-stream schedule (+ the step-level gantt of the latter),
# then three separate --pmc passes.  `sh profiles/r06_recipe.sh trace` stops after the traces.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r06
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d $O/single -o bench -- python $R/bench.py --steps 3 --warmup 1 $B --single-stream > $O/single_stdout.log 2>&1
# (--no-kernel-timing: the LAST step of the trace must be a step of the timed loop, not the single-stream instrumented one)
rocprofv3 --kernel-trace --stats -d $O/multi -o bench -- python $R/bench.py --steps 3 --warmup 1 $B --no-kernel-timing > $O/multi_stdout.log 2>&1
cd $R
python profiles/summarize_rocpd.py $O/single/bench_results.db $O/single.md > /dev/null
python profiles/summarize_rocpd.py $O/multi/bench_results.db $O/multi.md > /dev/null
python scratch/gantt.py $O/multi/bench_results.db > $O/gantt_multi.txt 2>&1
python scratch/gantt.py $O/single/bench_results.db > $O/gantt_single.txt 2>&1
# which kernels run while NO matrix-core kernel does (the exposed HBM / latency-bound time of the 3-stream step), in time order
python profiles/exposed_kernels.py $O/multi/bench_results.db > $O/exposed.txt 2>&1
# the whole step as ONE hipGraph replay on today's schedule (4 streams, least-priority weight-gradient streams,
# grouped launches) — its trace and gantt beside the eager one's
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/graph -o bench -- python $R/bench.py --steps 3 --warmup 1 $B --no-kernel-timing --graph > $O/graph_stdout.log 2>&1
cd $R
python profiles/summarize_rocpd.py $O/graph/bench_results.db $O/graph.md > /dev/null
python scratch/gantt.py $O/graph/bench_results.db > $O/gantt_graph.txt 2>&1
python profiles/exposed_kernels.py $O/graph/bench_results.db > $O/exposed_graph.txt 2>&1
if [ "$1" != "trace" ]; then
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py --steps 1 --warmup 1 $B --no-kernel-timing --single-stream > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python $R/bench.py --steps 1 --warmup 1 $B --no-kernel-timing --single-stream > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --steps 1 --warmup 1 $B --no-kernel-timing --single-stream > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- python $R/bench.py --steps 1 --warmup 1 $B --no-kernel-timing --single-stream > $O/pmc_sq2.log 2>&1
cd $R
python profiles/summarize_pmc.py $O/pmc_fetch/p_counter_collection.csv $O/pmc_write/p_counter_collection.csv $O/pmc_sq/p_counter_collection.csv $O/pmc.md $O/pmc.json conv_ 4 $O/pmc_sq2/p_counter_collection.csv > $O/pmc_summary.log 2>&1
fi
rm -rf $O/*/*.db $O/pmc_*/p_*.csv    # keep the merge-back small: summaries only
tail -1 $O/single_stdout.log | cut -c1-300
head -40 $O/single.md | cut -c1-170
cat $O/gantt_multi.txt
