cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h
mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "stem" 2>&1 | tail -2
python scratch/r5/host_profile.py > $O/host_profile.log 2>&1
head -60 $O/host_profile.log
