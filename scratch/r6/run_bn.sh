#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6bn
timeout 600 python scratch/r6/bn_time.py scratch/r6/libdynmm_bn1.so scratch/r6/libdynmm_bn2.so scratch/r6/libdynmm_bn4.so scratch/r6/libdynmm_bn8.so > gpurun_out/r6bn/bn_time.log 2>&1
cat gpurun_out/r6bn/bn_time.log
