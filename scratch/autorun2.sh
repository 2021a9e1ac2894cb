python -m pytest tests -m gpu -q --timeout=1200 > gpurun_out/auto2.log 2>&1; grep -E "^E  .*assert|FAILED|passed|failed" gpurun_out/auto2.log | cut -c1-250 | head -30
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c60-200
