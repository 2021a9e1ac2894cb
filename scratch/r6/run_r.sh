#!/bin/bash
cd $GRAFT_REPO_ROOT
sh profiles/r06_recipe.sh > gpurun_out/r6r_recipe.log 2>&1
tail -n 12 gpurun_out/r6r_recipe.log
