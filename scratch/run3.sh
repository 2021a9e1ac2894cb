mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py tests/test_hip_blocks.py -m gpu -q --timeout=900 -x > gpurun_out/t3.log 2>&1
grep -E "^E   |FAILED|passed|failed" gpurun_out/t3.log | cut -c1-300 | head -20
DYNMM_BENCH_SHAPES=gpurun_out/shapes_v2.txt python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_v2.log 2>&1; tail -1 gpurun_out/bench_v2.log | cut -c1-400
grep wgrad gpurun_out/shapes_v2.txt | head -12
