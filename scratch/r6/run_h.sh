#!/bin/bash
mkdir -p gpurun_out/r6h
cd $GRAFT_REPO_ROOT
TAPS=v timeout 600 python scratch/r6/wino_time.py cur o5 > gpurun_out/r6h/wino_vert_o5.log 2>&1
cat gpurun_out/r6h/wino_vert_o5.log
bash scratch/r6/ab_multi.sh r6h - scratch/r6/libdynmm_vtnst2.so scratch/r6/libdynmm_rounds2.so scratch/r6/libdynmm_rounds4.so
