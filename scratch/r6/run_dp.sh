cd $GRAFT_REPO_ROOT
O=gpurun_out/r6dp; mkdir -p $O; rm -f $O/dp_exchange.log
for rep in 1 2 3; do for m in plain wgrad depth comm; do timeout 300 python scratch/r6/dp_exchange_ab.py $m 2>$O/err_$m.log | grep "ms per step" | tee -a $O/dp_exchange.log; done; done
tail -3 $O/err_comm.log
