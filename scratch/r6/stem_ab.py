import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dynmm_amd.nn import net
net.STEM_DUAL = sys.argv[1] == '1'
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
