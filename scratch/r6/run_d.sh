#!/bin/bash
mkdir -p gpurun_out/r6d
cd $GRAFT_REPO_ROOT
timeout 600 python scratch/r6/w43_time.py v0 v3 v4 > gpurun_out/r6d/w43_time.log 2>&1
cat gpurun_out/r6d/w43_time.log
timeout 900 python -m pytest tests/test_engine.py tests/test_esanet.py tests/test_hip_model.py -x -q -m gpu -k "infer_step or esanet or compaction or evaluate or nyu8" > gpurun_out/r6d/pytest.log 2>&1
tail -n 30 gpurun_out/r6d/pytest.log
