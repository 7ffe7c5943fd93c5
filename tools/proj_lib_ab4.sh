# cfg-2, same box, interleaved, one stream and two: old projection kernels | new (nt weight stream) | new with the default cache policy
cd $GRAFT_REPO_ROOT
for ss in 1 0; do
 if [ $ss = 1 ]; then export LPM_SINGLE_STREAM=1; else unset LPM_SINGLE_STREAM; fi
 for rep in 1 2 3; do
  for v in old new plain; do
    unset LPM_HIP_LIBRARY LPM_PROJ_DX_STREAM_MIN_N
    if [ $v = old ]; then export LPM_HIP_LIBRARY=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib/liblpm_hip_oldproj.so LPM_PROJ_DX_STREAM_MIN_N=1024; fi
    if [ $v = plain ]; then export LPM_HIP_LIBRARY=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib/liblpm_hip_projplain.so; fi
    echo "cfg2 single_stream=$ss $v $(python bench.py --config cfg2 --steps 100 --warmup 10 --no-cpu-baseline --no-dispatch-count 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")"
  done
 done
done
