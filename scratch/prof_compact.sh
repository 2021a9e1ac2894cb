cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_compact
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extra --no-kernel-timing --mode fwd --batch 32 --hard --branches uniform"
rocprofv3 --kernel-trace --stats -d $O/c -o bench -- python $R/bench.py --steps 5 --warmup 2 $B --compact > $O/c.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/d -o bench -- python $R/bench.py --steps 5 --warmup 2 $B --no-compact > $O/d.log 2>&1
cd $R
python profiles/summarize_rocpd.py $O/c/bench_results.db $O/c.md > /dev/null
python profiles/summarize_rocpd.py $O/d/bench_results.db $O/d.md > /dev/null
rm -rf $O/*/*.db
tail -1 $O/c.log | cut -c1-200; tail -1 $O/d.log | cut -c1-200
head -24 $O/c.md | cut -c1-150
echo ---- dense
head -16 $O/d.md | cut -c1-150
