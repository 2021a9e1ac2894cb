#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in v6 v4; do
  if [ $mode = v4 ]; then export DYNMM_WGRAD_NO_V6=1; else unset DYNMM_WGRAD_NO_V6; fi
  GROUP=4 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/wg_$mode -o out -- python $R/scratch/wgrad_v6_check.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/wg_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode"; grep -i "wgrad\|reduce_slabs" $f | cut -c1-200
done
