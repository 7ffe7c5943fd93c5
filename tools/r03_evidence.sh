# round-3 evidence run on the GPU box: full GPU test tier, kernel-trace tables (every kernel) for cfg-2 / cfg-3 / cfg-5, PMC passes of the
# a5 chains, bench lines.  usage: bash tools/r03_evidence.sh <commit> <tag>      -> gpurun_out/<tag>/
C=${1:-unknown}
T=${2:-r03e}
mkdir -p gpurun_out/$T
python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/$T/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/gputests.log; tail -4 gpurun_out/$T/gputests.log
bash tools/pmc_a5.sh $C $T/pmc_r03_cfg2 cfg2 all > gpurun_out/$T/pmc_cfg2.log 2>&1
bash tools/pmc_a5.sh $C $T/pmc_r03_cfg3 cfg3 hbm > gpurun_out/$T/pmc_cfg3.log 2>&1
bash tools/pmc_a5.sh $C $T/pmc_r03_cfg5 cfg5 hbm > gpurun_out/$T/pmc_cfg5.log 2>&1
for c in cfg2 cfg3 cfg5; do cp gpurun_out/$T/pmc_r03_$c/a5_hbm_traffic_$c.json profiles/; done
bash tools/profile_bench.sh $T/r03_bench_kernel_stats_v9 > gpurun_out/$T/prof_cfg2.log 2>&1; tail -4 gpurun_out/$T/prof_cfg2.log
bash tools/profile_bench.sh $T/r03_cfg3_kernel_stats_v9 --config cfg3 > gpurun_out/$T/prof_cfg3.log 2>&1; tail -4 gpurun_out/$T/prof_cfg3.log
bash tools/profile_bench.sh $T/r03_cfg5_kernel_stats_v9 --config cfg5 > gpurun_out/$T/prof_cfg5.log 2>&1; tail -4 gpurun_out/$T/prof_cfg5.log
for c in cfg2 cfg3 cfg5; do python bench.py --config $c > gpurun_out/$T/bench_$c.json 2> gpurun_out/$T/bench_$c.err; echo "bench $c rc=$?"; done
