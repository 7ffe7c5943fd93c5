set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
rm -f gpurun_out/r06/time_k2_bf16_nt.log
for nt in 1 0; do for d in 0 32; do
  LPM_VB_NT=$nt LPM_VB_DBG=$d timeout 300 python tools/time_k2_bf16.py 9 2>&1 | grep "clip" | sed "s/^/nt=$nt dbg=$d /" >> gpurun_out/r06/time_k2_bf16_nt.log
done; done
LPM_VB_NT=0 timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench12_cfg5_nt0.json 2> gpurun_out/r06/bench12_cfg5_nt0.err
LPM_VB_NT=1 timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench12_cfg5_nt1.json 2> gpurun_out/r06/bench12_cfg5_nt1.err
