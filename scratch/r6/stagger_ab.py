"""A/B wrapper: python scratch/r6/stagger_ab.py <stages, e.g. 12 | 1234 | -> [bench args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dynmm_amd.nn import net
if sys.argv[1] != '-':
    net.STAGGER_STAGES = tuple(int(c) for c in sys.argv[1])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
