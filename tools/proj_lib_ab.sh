# same-box A/B of the projection kernels before / after round 4's counted-wait rings: whole steps of cfg-5 and cfg-2, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for c in cfg5 cfg2 cfg3; do
    for v in new old; do
      if [ $v = new ]; then unset LPM_HIP_LIBRARY LPM_PROJ_DX_STREAM_MIN_N; else export LPM_HIP_LIBRARY=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib/liblpm_hip_oldproj.so LPM_PROJ_DX_STREAM_MIN_N=1024; fi
      echo "$c $v $(python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline --no-dispatch-count 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")"
    done
  done
done
