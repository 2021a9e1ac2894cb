#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6j
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_engine.py tests/test_dp_gpu.py -x -q -m gpu -k "stream or infer or eval_run or rccl or two_train" > gpurun_out/r6j/pytest.log 2>&1
tail -n 8 gpurun_out/r6j/pytest.log
sh profiles/r06_recipe.sh > gpurun_out/r6j/recipe.log 2>&1
tail -n 60 gpurun_out/r6j/recipe.log
