#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6l
timeout 600 python -m pytest tests/test_hip_blocks.py tests/test_hip_ops.py -x -q -m gpu -k "feature_tap or grouped_weight or wgrad" > gpurun_out/r6l/pytest.log 2>&1
tail -n 4 gpurun_out/r6l/pytest.log
bash scratch/r6/ab_multi.sh r6l - scratch/r6/libdynmm_v6nst2.so scratch/r6/libdynmm_v6s2nst2.so
bash scratch/r6/ab_multi.sh r6l - scratch/r6/libdynmm_v6nst2.so scratch/r6/libdynmm_v6s2nst2.so
