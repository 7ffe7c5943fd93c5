cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 300 python tools/probe/r06_fold_diff.py > gpurun_out/r06/fold_diff.log 2>&1
