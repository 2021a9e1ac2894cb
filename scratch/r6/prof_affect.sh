#!/bin/sh
# rocprofv3 kernel stats of the ModalityDynMM step (graph replay): sh scratch/r6/prof_affect.sh TAG [STREAMS=1|0]
tag=${1:-affect}
export STREAMS=${2:-1}
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r06_affect
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/$tag -o aff -- python $GRAFT_REPO_ROOT/scratch/r4/affect_graph_probe.py 128 train 20 > $O/$tag.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py $(find $O/$tag -name '*.db' | head -1) > $O/$tag.md 2>>$O/$tag.log
rm -rf $O/$tag
tail -1 $O/$tag.log
head -40 $O/$tag.md | cut -c1-170
