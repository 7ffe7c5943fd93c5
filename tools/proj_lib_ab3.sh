# cfg-2, same box, interleaved, ONE stream (LPM_SINGLE_STREAM=1): old vs new projection kernels
cd $GRAFT_REPO_ROOT
export LPM_SINGLE_STREAM=1
for rep in 1 2 3; do
  for v in old new; do
    unset LPM_HIP_LIBRARY LPM_PROJ_DX_STREAM_MIN_N
    if [ $v = old ]; then export LPM_HIP_LIBRARY=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib/liblpm_hip_oldproj.so LPM_PROJ_DX_STREAM_MIN_N=1024; fi
    echo "cfg2 single-stream $v $(python bench.py --config cfg2 --steps 100 --warmup 10 --no-cpu-baseline --no-dispatch-count 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")"
  done
done
