cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "adam" 2>&1 | tail -5 > gpurun_out/r06/l2_kernel_tests.log
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_trajectory.py tests/test_gpu_dp_trainer.py -q -x 2>&1 | tail -5 > gpurun_out/r06/l2_model_tests.log
timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench43_cfg5.json 2> gpurun_out/r06/bench43_cfg5.err
LPM_L2_FOLD=0 timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench43_cfg5_off.json 2> gpurun_out/r06/bench43_cfg5_off.err
timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench43_cfg5_b.json 2> gpurun_out/r06/bench43_cfg5_b.err
LPM_L2_FOLD=0 timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench43_cfg5_off_b.json 2> gpurun_out/r06/bench43_cfg5_off_b.err
