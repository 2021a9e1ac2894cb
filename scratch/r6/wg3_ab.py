"""A/B wrapper: python scratch/r6/wg3_ab.py <n weight-gradient streams> [bench args] — with the stream plan's pair probe choosing
non-colliding streams, is a fifth busy stream still a 10 ms cliff (profiles/r05_ab_runs.md)?"""
import os, sys, runpy, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dynmm_amd import ops
n = int(sys.argv[1])
ops.WGRAD_STREAMS = n
ops.MAX_BUSY_STREAMS = 2 + n
import torch
plan = ops.stream_plan()
print('plan report', json.dumps(plan.report), file=sys.stderr)
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
