# bench.py with two ranks sharing the one GPU over gloo on each route of hidden1_weights (LPM_HIDDEN1_ROUTE), cfg-2 and cfg-5, then eight ranks on
# the default route: prints the replica consistency report of every run
cd $GRAFT_REPO_ROOT
for r in sharded factored allreduce; do
  for c in cfg2 cfg5; do
    LPM_HIDDEN1_ROUTE=$r LPM_SHARE_GPU=1 timeout 600 python bench.py --config $c --gpus 2 --steps 3 --warmup 1 --spinup-seconds 0 --no-cpu-baseline 2>/tmp/e.log | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('$r $c', d['n_gpus'], d['ms_per_step'], d.get('replicas'))" || tail -5 /tmp/e.log
  done
done
LPM_SHARE_GPU=1 timeout 900 python bench.py --config cfg2 --gpus 8 --steps 2 --warmup 1 --spinup-seconds 0 --no-cpu-baseline 2>/tmp/e.log | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('default cfg2 x8', d['n_gpus'], d['ms_per_step'], d.get('replicas'))" || tail -5 /tmp/e.log
