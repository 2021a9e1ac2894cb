#!/bin/bash
# 3-stream step under rocprofv3 --kernel-trace: the rocpd database comes back for scratch/gantt.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/gantt -o g -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra --no-kernel-timing > $R/gpurun_out/gantt/stdout.log 2>&1
ls -la $R/gpurun_out/gantt/
tail -1 $R/gpurun_out/gantt/stdout.log | cut -c1-200
