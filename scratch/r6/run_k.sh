#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6k
timeout 600 python -m pytest tests/test_hip_blocks.py -x -q -m gpu > gpurun_out/r6k/pytest.log 2>&1
tail -n 4 gpurun_out/r6k/pytest.log
bash scratch/r6/ab_multi.sh r6k - scratch/r6/libdynmm_vtnst2.so
bash scratch/r6/ab_multi.sh r6k - scratch/r6/libdynmm_vtnst2.so
