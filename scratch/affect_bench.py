"""ModalityDynMM affect path stand-alone (for rocprofv3): python scratch/affect_bench.py [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
print(json.dumps(bench.measure_affect(torch.device("cuda", 0), steps)))
