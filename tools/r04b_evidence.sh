# round-4 (second session) evidence run on the GPU box: the GPU test tier, kernel-trace tables (every kernel) for cfg-2 / cfg-3 / cfg-5,
# HBM-traffic PMC passes of the a5 chain of each, the PMC pass of K1, the three bench lines.  usage: bash tools/r04b_evidence.sh <commit> <tag>
C=${1:-unknown}
T=${2:-r04f}
mkdir -p gpurun_out/$T
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/$T/gpu_tests_tail.log; tail -2 gpurun_out/$T/gpu_tests_tail.log
for c in cfg2 cfg3 cfg5; do timeout 300 bash tools/pmc_a5.sh $C $T/pmc_$c $c hbm > gpurun_out/$T/pmc_$c.log 2>&1; grep chain_vs gpurun_out/$T/pmc_$c.log; done
timeout 300 bash tools/k1_pmc.sh > gpurun_out/$T/k1_pmc.log 2>&1; cp gpurun_out/pmc_k1w_summary.txt gpurun_out/$T/
timeout 400 bash tools/profile_bench.sh $T/cfg2_kernel_stats > gpurun_out/$T/prof_cfg2.log 2>&1; tail -4 gpurun_out/$T/prof_cfg2.log | cut -c1-160
timeout 400 bash tools/profile_bench.sh $T/cfg3_kernel_stats --config cfg3 > gpurun_out/$T/prof_cfg3.log 2>&1; tail -4 gpurun_out/$T/prof_cfg3.log | cut -c1-160
timeout 400 bash tools/profile_bench.sh $T/cfg5_kernel_stats --config cfg5 > gpurun_out/$T/prof_cfg5.log 2>&1; tail -4 gpurun_out/$T/prof_cfg5.log | cut -c1-160
timeout 400 python bench.py --config all > gpurun_out/$T/bench_all.jsonl 2> gpurun_out/$T/bench_all.err; echo "bench all rc=$?"
timeout 200 bash tools/step_anatomy.sh $T/r04b cfg2 > /dev/null 2>&1
