set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for d in 0 15 47 63 0; do
  LPM_FA_FOLD_DBG=$d timeout 200 python tools/time_factored_fold.py dx 2>&1 | grep "mode="
done > gpurun_out/r06/time_fold_dbg2.log 2>&1
LPM_FA_FOLD=2 timeout 200 python tools/time_factored_fold.py copy 2>&1 | grep "mode=" >> gpurun_out/r06/time_fold_dbg2.log
