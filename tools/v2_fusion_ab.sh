# cfg-3, same box, interleaved: the V2 encoder's round-4 fusions on / off -- (a) batch norm + dense as one node and the layer norm's operand
# image for the feed-forward network, (b) the dropout + bias add between output_transform and the layer norm inside the layer norm's passes
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in "all" "no-dropout-fusion" "none"; do
    unset LPM_BN_DENSE_FUSED LPM_LN_IMAGE LPM_LN_DROPOUT_FUSED
    if [ $v = none ]; then export LPM_BN_DENSE_FUSED=0 LPM_LN_IMAGE=0 LPM_LN_DROPOUT_FUSED=0; fi
    if [ $v = no-dropout-fusion ]; then export LPM_LN_DROPOUT_FUSED=0; fi
    echo "cfg3 fusions=$v $(python bench.py --config cfg3 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('dispatches_per_step',{}).get('value'))")"
  done
done
