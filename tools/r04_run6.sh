for v in "LPM_TG_WIDE_NS=4" "LPM_TG_WIDE_NS=5" "LPM_TG_WIDE_NS=4" "LPM_TG_WIDE_NS=5"; do
  env $v python bench.py --no-cpu-baseline --no-dispatch-count > gpurun_out/r04_ab.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_ab.json").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["assign_gemm"]["avg_kernel_ms"], d["assign_gemm"]["mfma_util_vs_bf16_peak"], d["roofline"]["avg_kernel_ms"])
PY
done
