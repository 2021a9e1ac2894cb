python scratch/trace/run_trace.py > gpurun_out/trace.txt 2>&1
grep -A4 "^C=" gpurun_out/trace.txt
python -m pytest tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -3
python scratch/ksweep.py 2>&1 | tail -12
