#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6o; mkdir -p $O
for lib in strips strips2; do
timeout 900 python scratch/r6/pytest_lib.py scratch/r6/libdynmm_$lib.so tests/test_hip_ops.py tests/test_hip_blocks.py -x -q -m gpu -k "wgrad or grouped or block or encoder or decoder" > $O/pytest_$lib.log 2>&1
tail -n 4 $O/pytest_$lib.log
done
bash scratch/r6/ab_multi.sh r6o - scratch/r6/libdynmm_strips.so scratch/r6/libdynmm_strips2.so
