python -m pytest tests -m gpu -q --timeout=1200 2>&1 | grep -E "^E  .*assert|FAILED|passed|failed" | cut -c1-250 | head -20
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c60-200
python bench.py --mode fwd --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | cut -c50-200
DYNMM_PRECISION=fp32 python bench.py --mode fwd --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | cut -c50-200
