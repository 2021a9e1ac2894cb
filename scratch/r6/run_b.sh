#!/bin/bash
mkdir -p gpurun_out/r6b
cd $GRAFT_REPO_ROOT
timeout 600 python scratch/r6/w43_time.py v0 v2 > gpurun_out/r6b/w43_time.log 2>&1
cat gpurun_out/r6b/w43_time.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6b/pytest.log 2>&1
tail -n 8 gpurun_out/r6b/pytest.log
timeout 900 python bench.py > gpurun_out/r6b/bench_full.log 2>&1
tail -c 6000 gpurun_out/r6b/bench_full.log
