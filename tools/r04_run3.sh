python -m pytest tests/test_gpu_models.py -q -x -k "lazily" 2>&1 | tail -2
for cfg in cfg3 cfg5; do for nt in 0 3; do
  LPM_T3_NT=$nt python bench.py --config $cfg > gpurun_out/r04_nt_${cfg}_$nt.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_nt_${cfg}_$nt.json").read().strip().splitlines()[-1])
print("$cfg LPM_T3_NT=$nt", d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"].get("a5_function",{}).get("kernel_ms"))
PY
done; done
bash tools/pmc_a5.sh 35bec0b pmc_r04_cfg2 cfg2 all > gpurun_out/r04_pmc_cfg2.log 2>&1; tail -30 gpurun_out/r04_pmc_cfg2.log
