# does torch's TunableOp (per-shape search over the hipBLASLt / rocBLAS solutions) cover the split-bf16 image GEMMs (bf16 in, fp32 out), and what
# does it buy the cfg-2 step?  Writes gpurun_out/tunable/{results.csv, base.json, tuned.json, tuned2.json}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tunable
python bench.py --no-cpu-baseline > gpurun_out/tunable/base.json 2> gpurun_out/tunable/base.err
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$GRAFT_REPO_ROOT/gpurun_out/tunable/results.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=60 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 PYTORCH_TUNABLEOP_VERBOSE=1
t0=$(date +%s)
python bench.py --no-cpu-baseline > gpurun_out/tunable/tuned.json 2> gpurun_out/tunable/tuned.err
echo "tuning run $(( $(date +%s) - t0 )) s"
export PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_VERBOSE=0
python bench.py --no-cpu-baseline > gpurun_out/tunable/tuned2.json 2> gpurun_out/tunable/tuned2.err
unset PYTORCH_TUNABLEOP_ENABLED
python bench.py --no-cpu-baseline > gpurun_out/tunable/base2.json 2> gpurun_out/tunable/base2.err
ls -la gpurun_out/tunable/; wc -l gpurun_out/tunable/results*.csv; head -5 gpurun_out/tunable/results*.csv | cut -c1-200
python - <<'PY'
import json
for n in ("base", "tuned", "tuned2", "base2"):
    try:
        d = json.loads(open(f"gpurun_out/tunable/{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -5 gpurun_out/tunable/tuned.err | cut -c1-300
