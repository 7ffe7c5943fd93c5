# cfg-3, same box, interleaved: the V2 encoder's round-4 fusions (batch norm + dense as one node; the layer norm's operand image for the
# feed-forward network) on / off
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in "on" "off"; do
    if [ $v = on ]; then unset LPM_BN_DENSE_FUSED LPM_LN_IMAGE; else export LPM_BN_DENSE_FUSED=0 LPM_LN_IMAGE=0; fi
    echo "cfg3 fusions=$v $(python bench.py --config cfg3 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('dispatches_per_step',{}).get('value'))")"
  done
done
