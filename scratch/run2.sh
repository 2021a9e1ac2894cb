mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py tests/test_hip_blocks.py -m gpu -q --timeout=900 -x > gpurun_out/t2.log 2>&1
grep -E "^E   |FAILED|passed|failed" gpurun_out/t2.log | cut -c1-300 | head -20
DYNMM_BENCH_SHAPES=gpurun_out/shapes_v1.txt python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_v1.log 2>&1; tail -1 gpurun_out/bench_v1.log | cut -c1-1500
head -40 gpurun_out/shapes_v1.txt
