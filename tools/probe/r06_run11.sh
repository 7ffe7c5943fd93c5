set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -s -k "k2_bf16_clip or netvlad_bf16_storage" 2>&1 | grep -E "passed|failed|Error|assert" | cut -c1-300 > gpurun_out/r06/clip16_tests6.log
for d in 0 1 2 4 8 16 32 12 33; do
  LPM_VB_DBG=$d timeout 300 python tools/time_k2_bf16.py 9 2>&1 | grep "clip" | sed "s/^/dbg=$d /" >> gpurun_out/r06/time_k2_bf16_ablate.log
done
