#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6m
timeout 900 python -m pytest tests/test_hip_blocks.py tests/test_hip_ops.py -x -q -m gpu -k "feature_tap or grouped_weight or wgrad or conv" > gpurun_out/r6m/pytest.log 2>&1
tail -n 4 gpurun_out/r6m/pytest.log
bash scratch/r6/ab_multi.sh r6m - scratch/r6/libdynmm_winoring2.so
bash scratch/r6/ab_multi.sh r6m - scratch/r6/libdynmm_winoring2.so
