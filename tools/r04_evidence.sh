# round-4 evidence run on the GPU box: kernel-trace tables (every kernel) for cfg-2 / cfg-3 / cfg-5, PMC passes of the cfg-2 a5 chain,
# the three bench lines.  usage: bash tools/r04_evidence.sh <commit> <tag>      -> gpurun_out/<tag>/
C=${1:-unknown}
T=${2:-r04e}
mkdir -p gpurun_out/$T
bash tools/pmc_a5.sh $C $T/pmc_r04_cfg2 cfg2 all > gpurun_out/$T/pmc_cfg2.log 2>&1
bash tools/profile_bench.sh $T/r04_bench_kernel_stats_v1 > gpurun_out/$T/prof_cfg2.log 2>&1; tail -4 gpurun_out/$T/prof_cfg2.log
bash tools/profile_bench.sh $T/r04_cfg3_kernel_stats_v1 --config cfg3 > gpurun_out/$T/prof_cfg3.log 2>&1; tail -4 gpurun_out/$T/prof_cfg3.log
bash tools/profile_bench.sh $T/r04_cfg5_kernel_stats_v1 --config cfg5 > gpurun_out/$T/prof_cfg5.log 2>&1; tail -4 gpurun_out/$T/prof_cfg5.log
python bench.py --config all > gpurun_out/$T/bench_all.jsonl 2> gpurun_out/$T/bench_all.err; echo "bench all rc=$?"
