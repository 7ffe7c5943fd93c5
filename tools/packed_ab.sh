# A/B of two builds of liblpm_hip.so: default flags (no packed fp32 ops) vs _lib/liblpm_hip_packed.so (compiler free to use v_pk_*_f32)
cd $GRAFT_REPO_ROOT
TAILN=400 bash tools/determinism.sh "blocks 120 0 tap" > gpurun_out/det_nopacked.log 2>&1
echo "in-situ mismatches (no packed ops): $(grep -c 'first call' gpurun_out/det_nopacked.log)"; grep "runs," gpurun_out/det_nopacked.log
P=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib/liblpm_hip_packed.so
for c in cfg2 cfg3 cfg5; do
  for v in nopacked packed nopacked packed; do
    if [ $v = packed ]; then export LPM_HIP_LIBRARY=$P; else unset LPM_HIP_LIBRARY; fi
    echo "$c $v $(python bench.py --config $c --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")"
  done
done
