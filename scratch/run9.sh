mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py tests/test_hip_blocks.py tests/test_engine.py -m gpu -q --timeout=900 > gpurun_out/t9.log 2>&1
grep -E "^E   |FAILED|passed|failed" gpurun_out/t9.log | cut -c1-300 | head -20
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_v9.log 2>&1; tail -1 gpurun_out/bench_v9.log | cut -c1-330
