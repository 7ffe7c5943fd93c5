cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "factored_update_returns" 2>&1 | tail -40 > gpurun_out/r06/fold_kernel_tests3.log
