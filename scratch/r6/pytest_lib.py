"""python scratch/r6/pytest_lib.py <alternative libdynmm_hip.so> [pytest args]: the GPU tests against another build of the library"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from dynmm_amd import lib as L
L.LIB_PATH = os.path.join(ROOT, sys.argv[1])
import pytest
sys.exit(pytest.main(sys.argv[2:]))
