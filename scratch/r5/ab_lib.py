"""A/B wrapper (scratch): python scratch/r5/ab_lib.py <path to an alternative libdynmm_hip.so | -> [bench args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dynmm_amd import lib as L
if sys.argv[1] != '-':
    L.LIB_PATH = os.path.join(ROOT, sys.argv[1])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
