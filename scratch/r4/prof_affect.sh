#!/bin/sh
# rocprofv3 kernel stats of the ModalityDynMM step (graph replay, 5 branch streams): sh scratch/r4/prof_affect.sh [tag]
tag=${1:-affect}
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r04_affect
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/$tag -o aff -- python $GRAFT_REPO_ROOT/scratch/r4/affect_graph_probe.py 128 train 20 > $O/$tag.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py $(find $O/$tag -name '*.db' | head -1) > $O/$tag.md 2>>$O/$tag.log
tail -2 $O/$tag.log
head -45 $O/$tag.md | cut -c1-190
