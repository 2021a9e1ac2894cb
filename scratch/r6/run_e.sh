#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scratch/r6/ab_lib.sh r6e scratch/r6/libdynmm_w43wg2.so
timeout 900 python -m pytest tests/test_engine.py tests/test_hip_ops.py -x -q -m gpu -k "infer_step or wino or conv2d" > gpurun_out/r6e/pytest.log 2>&1
tail -n 5 gpurun_out/r6e/pytest.log
