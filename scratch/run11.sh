python -m pytest tests/test_hip_ops.py tests/test_hip_blocks.py -m gpu -q --timeout=900 2>&1 | tail -3
DYNMM_BENCH_SHAPES=gpurun_out/shapes11.txt python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c60-200
grep "wgrad<co64>" gpurun_out/shapes11.txt | head -4
