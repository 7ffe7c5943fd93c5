# cfg-2, same box, interleaved: old projection kernels | new forward + library dx | new forward + own dx  (whole steps, 100 timed steps each)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  for v in old newfwd new; do
    unset LPM_HIP_LIBRARY LPM_PROJ_DX_STREAM_MIN_N
    if [ $v = old ]; then export LPM_HIP_LIBRARY=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib/liblpm_hip_oldproj.so LPM_PROJ_DX_STREAM_MIN_N=1024; fi
    if [ $v = newfwd ]; then export LPM_PROJ_DX_STREAM_MIN_N=1024; fi
    echo "cfg2 $v $(python bench.py --config cfg2 --steps 100 --warmup 10 --no-cpu-baseline --no-dispatch-count 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")"
  done
done
