#!/bin/bash
# same-box comparison of the three-tap weight-gradient kernel against the v4 kernel (grouped launches of 4)
export GROUP=4
run() { echo "=== $*"; env "$@" timeout 300 python scratch/wgrad_v6_check.py 2>&1 | grep -v amdgpu.ids | grep -A1 "^(32" | grep "group"; }
run DYNMM_WGRAD_NO_V6=1
run DYNMM_WGRAD_V6_OCC=0
run DYNMM_WGRAD_V6_BLOCKS=1024
