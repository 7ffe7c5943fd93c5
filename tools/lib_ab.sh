# A/B of alternative builds of the library inside one box: bash tools/lib_ab.sh <cfg> <lib-suffix> [<lib-suffix> ...]   ("" = the default build)
cd $GRAFT_REPO_ROOT
C=$1; shift
for rep in 1 2; do
  for v in default "$@"; do
    if [ $v = default ]; then unset LPM_HIP_LIBRARY; else export LPM_HIP_LIBRARY=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib/liblpm_hip_$v.so; fi
    echo "$C $v $(python bench.py --config $C --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")"
  done
done
