cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O
for rep in 1 2 3; do for m in plain forced; do timeout 300 python scratch/r5/dp_streams.py $m 2>$O/dp_err.log | tail -1 | tee -a $O/dp_streams.log; done; done
tail -3 $O/dp_err.log
