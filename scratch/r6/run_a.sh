#!/bin/bash
# round 6, call A: the stream-lottery experiment + the tests the stream plan touches
mkdir -p gpurun_out/r6a
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  timeout 300 python scratch/r6/r5_models_in_sequence.py 8 > gpurun_out/r6a/r5_seq_$rep.log 2>&1
  timeout 300 python scratch/r6/stream_lottery.py plan 8 > gpurun_out/r6a/plan_$rep.log 2>&1
  timeout 300 python scratch/r6/stream_lottery.py pool 8 > gpurun_out/r6a/pool_$rep.log 2>&1
done
timeout 400 python scratch/r6/stream_lottery.py sweep 6 > gpurun_out/r6a/sweep.log 2>&1
tail -n 3 gpurun_out/r6a/*.log
timeout 900 python -m pytest tests/test_engine.py tests/test_dp_gpu.py tests/test_hip_ops.py -x -q -m gpu -k "stream or rccl or two_train_steps_match or sharded or bucket" > gpurun_out/r6a/pytest.log 2>&1
tail -n 15 gpurun_out/r6a/pytest.log
