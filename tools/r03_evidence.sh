mkdir -p gpurun_out/r03d
python -m pytest tests -m gpu -x -q > gpurun_out/r03d/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03d/gputests.log; tail -4 gpurun_out/r03d/gputests.log
bash tools/profile_bench.sh r03d/r03_bench_kernel_stats_v1 > gpurun_out/r03d/prof_cfg2.log 2>&1; tail -4 gpurun_out/r03d/prof_cfg2.log
bash tools/profile_bench.sh r03d/r03_cfg3_kernel_stats_v1 --config cfg3 > gpurun_out/r03d/prof_cfg3.log 2>&1; tail -4 gpurun_out/r03d/prof_cfg3.log
bash tools/profile_bench.sh r03d/r03_cfg5_kernel_stats_v1 --config cfg5 > gpurun_out/r03d/prof_cfg5.log 2>&1; tail -4 gpurun_out/r03d/prof_cfg5.log
bash tools/pmc_a5.sh c811cad r03d/pmc_r03_cfg2 cfg2 all > gpurun_out/r03d/pmc_cfg2.log 2>&1; tail -3 gpurun_out/r03d/pmc_cfg2.log
bash tools/pmc_a5.sh c811cad r03d/pmc_r03_cfg3 cfg3 hbm > gpurun_out/r03d/pmc_cfg3.log 2>&1; tail -3 gpurun_out/r03d/pmc_cfg3.log
bash tools/pmc_a5.sh c811cad r03d/pmc_r03_cfg5 cfg5 hbm > gpurun_out/r03d/pmc_cfg5.log 2>&1; tail -3 gpurun_out/r03d/pmc_cfg5.log
for c in cfg2 cfg3 cfg5; do python bench.py --config $c > gpurun_out/r03d/bench_$c.json 2> gpurun_out/r03d/bench_$c.err; echo "bench $c rc=$?"; done
