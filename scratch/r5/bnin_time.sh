cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h
mkdir -p $O
git init -q . 2>/dev/null
patch -p1 < scratch/r5/bnin_async_apply.patch > $O/patch.log 2>&1 || (tail -5 $O/patch.log; exit 1)
make -C dynmm_amd/csrc -j8 > $O/make.log 2>&1 || (tail -5 $O/make.log; exit 1)
python scratch/r5/bnin_time.py 2>&1 | tee $O/bnin_time.log
