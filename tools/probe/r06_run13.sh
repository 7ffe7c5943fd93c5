set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r06/gpu_tests13.log
bash tools/profile_bench.sh r06/r06_cfg2 > gpurun_out/r06/profile_cfg2.log 2>&1
bash tools/profile_bench.sh r06/r06_cfg3 --config cfg3 > gpurun_out/r06/profile_cfg3.log 2>&1
bash tools/profile_bench.sh r06/r06_cfg5 --config cfg5 > gpurun_out/r06/profile_cfg5.log 2>&1
bash tools/pmc_a5.sh 9f0d4dc r06/pmc_cfg2 cfg2 hbm > gpurun_out/r06/pmc_cfg2.log 2>&1
bash tools/pmc_a5.sh 9f0d4dc r06/pmc_cfg3 cfg3 hbm > gpurun_out/r06/pmc_cfg3.log 2>&1
bash tools/pmc_a5.sh 9f0d4dc r06/pmc_cfg5 cfg5 hbm > gpurun_out/r06/pmc_cfg5.log 2>&1
cd $GRAFT_REPO_ROOT
LPM_SINGLE_STREAM=1 timeout 600 python tools/graph_ab.py cfg2 40 > gpurun_out/r06/graph_ab_cfg2_single.log 2>&1
