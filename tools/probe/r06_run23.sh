set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export LPM_HIP_LIBRARY=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib/liblpm_hip_k1exp.so
for d in 0 1 2 4 6 8 16 22 64 0; do
  LPM_K1_WIDE_DBG=$d timeout 120 python tools/k1_bf16_loop.py 300 2>&1 | grep "K1 bf16"
done > gpurun_out/r06/k1_wide_ablations.log 2>&1
