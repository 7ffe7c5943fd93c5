set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
rm -f gpurun_out/r06/time_k2_bf16_delay.log
for d in 0 4 8 12 16; do
  if [ $d = 0 ]; then D=0; else D=64; fi
  LPM_VB_DELAY=$d LPM_VB_DBG=$D timeout 300 python tools/time_k2_bf16.py 9 2>&1 | grep "clip" | sed "s/^/delay=$d dbg=$D /" >> gpurun_out/r06/time_k2_bf16_delay.log
done
