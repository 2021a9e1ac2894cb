# full validation of the tree: GPU tests, the default bench line, the profiling recipe + exposed-kernel analysis
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1
tail -3 $O/gputests.log
timeout 900 python bench.py > $O/bench_full.log 2> $O/bench_full.err
tail -1 $O/bench_full.log | cut -c1-300
sh profiles/r05_recipe.sh > $O/recipe.log 2>&1
tail -30 $O/recipe.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
