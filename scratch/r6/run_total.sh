#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scratch/r6/ab_multi.sh r6total scratch/r6/libdynmm_r5.so -
bash scratch/r6/ab_multi.sh r6total scratch/r6/libdynmm_r5.so -
