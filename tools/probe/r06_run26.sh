set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for d in 0 1 2 3 4 8 11 15 16 31; do
  LPM_FA_FOLD_DBG=$d timeout 200 python tools/time_factored_fold.py dx 2>&1 | grep "mode="
done > gpurun_out/r06/time_fold_dbg.log 2>&1
