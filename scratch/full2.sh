mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout=1200 > gpurun_out/full_gpu2.log 2>&1
grep -E "^E   |FAILED|passed|failed" gpurun_out/full_gpu2.log | cut -c1-300 | head -20
python bench.py > gpurun_out/bench_default2.log 2>&1; tail -1 gpurun_out/bench_default2.log
