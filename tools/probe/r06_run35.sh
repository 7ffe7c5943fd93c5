cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench35_cfg5.json 2> gpurun_out/r06/bench35_cfg5.err
LPM_FA_FOLD=2 timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench35_cfg5_rowblock.json 2> gpurun_out/r06/bench35_cfg5_rowblock.err
timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench35_cfg5_b.json 2> gpurun_out/r06/bench35_cfg5_b.err
LPM_FA_FOLD=2 timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench35_cfg5_rowblock_b.json 2> gpurun_out/r06/bench35_cfg5_rowblock_b.err
bash tools/profile_bench.sh r06/r06c_cfg2 > gpurun_out/r06/profile_cfg2_c.log 2>&1
bash tools/profile_bench.sh r06/r06c_cfg5 --config cfg5 > gpurun_out/r06/profile_cfg5_c.log 2>&1
