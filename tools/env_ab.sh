# A/B of a library environment switch inside one box: bash tools/env_ab.sh <cfg> <VAR> <value A> <value B>
cd $GRAFT_REPO_ROOT
C=$1; V=$2; shift 2
for rep in 1 2; do
  for val in "$@"; do
    echo "$C $V=$val $(env $V=$val python bench.py --config $C --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")"
  done
done
