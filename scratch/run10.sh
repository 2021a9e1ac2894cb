python -m pytest tests/test_engine.py tests/test_hip_blocks.py -m gpu -q --timeout=900 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
