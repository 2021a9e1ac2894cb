export DYNMM_PRECISION=bf16x3
python -m pytest tests/test_hip_ops.py tests/test_hip_blocks.py -m gpu -q --timeout=900 2>&1 | grep -E "FAILED|passed|failed" | head -20
python -m pytest tests/test_hip_model.py tests/test_engine.py -m gpu -q --timeout=900 2>&1 | grep -E "^E  .*assert|FAILED|passed|failed" | cut -c1-250 | head -20
DYNMM_BENCH_SHAPES=gpurun_out/shapes_bf.txt python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
