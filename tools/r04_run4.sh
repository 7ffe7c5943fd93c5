for v in "LPM_VC_NT=1 LPM_VC_NS=4" "LPM_VC_NT=0 LPM_VC_NS=4" "LPM_VC_NT=1 LPM_VC_NS=3" "LPM_VLAD_CLIP=0"; do
  env $v python bench.py > gpurun_out/r04_ab.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_ab.json").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["roofline"].get("a5_function",{}).get("kernel_ms"))
PY
done
