# round-4 evidence run on the GPU box: kernel-trace tables (every kernel) for cfg-2 / cfg-3 / cfg-5, PMC passes of the cfg-2 a5 chain,
# the three bench lines, the projection kernels' A/B (isolated + in the step against the old proj_gemm.hip, tools/build_old_proj_variant.sh
# first) and the power samples.  usage: bash tools/r04_evidence.sh <commit> <tag>      -> gpurun_out/<tag>/
C=${1:-unknown}
T=${2:-r04e}
mkdir -p gpurun_out/$T
bash tools/pmc_a5.sh $C $T/pmc_r04_cfg2 cfg2 all > gpurun_out/$T/pmc_cfg2.log 2>&1
bash tools/profile_bench.sh $T/r04_bench_kernel_stats_v2 > gpurun_out/$T/prof_cfg2.log 2>&1; tail -4 gpurun_out/$T/prof_cfg2.log
bash tools/profile_bench.sh $T/r04_cfg3_kernel_stats_v2 --config cfg3 > gpurun_out/$T/prof_cfg3.log 2>&1; tail -4 gpurun_out/$T/prof_cfg3.log
bash tools/profile_bench.sh $T/r04_cfg5_kernel_stats_v2 --config cfg5 > gpurun_out/$T/prof_cfg5.log 2>&1; tail -4 gpurun_out/$T/prof_cfg5.log
python bench.py --config all > gpurun_out/$T/bench_all.jsonl 2> gpurun_out/$T/bench_all.err; echo "bench all rc=$?"
bash tools/proj_dx_ab.sh > gpurun_out/$T/r04_proj_kernels.txt 2>&1
if [ -f learnablepoolingmethods_amd/_lib/liblpm_hip_oldproj.so ]; then
  bash tools/proj_lib_ab.sh > gpurun_out/$T/r04_proj_step_ab.txt 2>&1
  bash tools/proj_prof_ab.sh > gpurun_out/$T/r04_proj_step_kernel_diff.txt 2>&1
fi
for c in cfg2 cfg5 cfg3; do bash tools/power_sample.sh $c 1200; done > gpurun_out/$T/r04_power_samples.txt 2>&1
bash tools/step_anatomy.sh $T/r04 cfg2 > /dev/null 2>&1; python tools/step_gaps.py gpurun_out/$T/r04_step_cfg2.md 8 | tail -2
