#!/bin/bash
mkdir -p gpurun_out/r6c
cd $GRAFT_REPO_ROOT
timeout 600 python scratch/r6/w43_time.py v0 v2 v3 > gpurun_out/r6c/w43_time.log 2>&1
cat gpurun_out/r6c/w43_time.log
timeout 600 python scratch/r6/wino_time.py v0 v2 > gpurun_out/r6c/wino_time.log 2>&1
cat gpurun_out/r6c/wino_time.log
