cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "factored or adam" 2>&1 | tail -5 > gpurun_out/r06/adam_kernel_tests.log
timeout 300 python tools/probe/r06_fold_diff.py > gpurun_out/r06/fold_diff2.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_models.py -q -x 2>&1 | tail -8 > gpurun_out/r06/adam_model_tests.log
