python -m pytest tests/test_gpu_kernels.py -q -x -k "sum_splits or weight_pack or layer_norm_image" 2>&1 | tail -3
python -m pytest tests/test_gpu_models.py -q -x -k "cfg2 or train_steps or reproducible" 2>&1 | tail -3
for v in "LPM_SUM_SPLITS=1" "LPM_SUM_SPLITS=0" "LPM_SUM_SPLITS=1" "LPM_SUM_SPLITS=0"; do
  env $v python bench.py --no-cpu-baseline > gpurun_out/r04_ab.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_ab.json").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["value"], d.get("dispatches_per_step",{}).get("value"), d["assign_gemm"]["avg_kernel_ms"])
PY
done
