mkdir -p gpurun_out
python -m pytest tests/test_hip_model.py -m gpu -q --timeout=900 -k goldens > gpurun_out/model3.log 2>&1
grep -E "^E   |FAILED|passed|failed" gpurun_out/model3.log | cut -c1-300 | head -20
python bench.py --steps 3 --warmup 1 --batch 8 --no-graph > gpurun_out/bench_b8_eager.log 2>&1; tail -2 gpurun_out/bench_b8_eager.log
python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline > gpurun_out/bench_b8_graph.log 2>&1; tail -2 gpurun_out/bench_b8_graph.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_b32.log 2>&1; tail -2 gpurun_out/bench_b32.log
