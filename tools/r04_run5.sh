python -m pytest tests/test_gpu_kernels.py -q -x -k "weight_pack" 2>&1 | tail -5
python -m pytest tests/test_gpu_models.py -q -x 2>&1 | tail -4
for v in "LPM_WEIGHT_PACK=1" "LPM_WEIGHT_PACK=0" "LPM_WEIGHT_PACK=1" "LPM_WEIGHT_PACK=0"; do
  env $v python bench.py --no-cpu-baseline > gpurun_out/r04_ab.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_ab.json").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["value"], d.get("dispatches_per_step",{}).get("value"))
PY
done
