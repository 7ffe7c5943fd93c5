# GPU clock / power samples (rocm-smi, every 0.5 s) while bench.py runs a long timed region
cd $GRAFT_REPO_ROOT
CFG=${1:-cfg2}; STEPS=${2:-1500}
rocm-smi --showmaxpower 2>&1 | grep -i "max" | head -2
( for i in $(seq 1 36); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*(\([0-9]*\)Mhz).*/sclk \1/; s/.*(W): \([0-9.]*\)/W \1/' | tr '\n' ' '; echo; sleep 0.5; done ) > /tmp/smi.log &
SP=$!
python bench.py --config $CFG --steps $STEPS --warmup 20 --no-cpu-baseline --no-dispatch-count 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('bench', d['ms_per_step'], d['value'])"
wait $SP
python - <<'PY'
import re
rows=[l.split() for l in open("/tmp/smi.log") if "sclk" in l and "W" in l]
v=[(int(r[1]), float(r[3])) for r in rows if len(r)>=4]
busy=[x for x in v if x[0]>1000]
print("samples under load:", len(busy), " sclk MHz min/mean/max", min(b[0] for b in busy), round(sum(b[0] for b in busy)/len(busy)), max(b[0] for b in busy),
      " package W min/mean/max", min(b[1] for b in busy), round(sum(b[1] for b in busy)/len(busy)), max(b[1] for b in busy))
print("idle:", [x for x in v if x[0]<=1000][:3])
PY
