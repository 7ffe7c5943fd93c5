set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "factored" 2>&1 | tail -8 > gpurun_out/r06/fold_kernel_tests.log
{
timeout 200 python tools/time_factored_fold.py copy
LPM_FA_FOLD=2 timeout 200 python tools/time_factored_fold.py copy
timeout 200 python tools/time_factored_fold.py dx
timeout 200 python tools/time_factored_fold.py copy
LPM_FA_FOLD=2 timeout 200 python tools/time_factored_fold.py copy
timeout 200 python tools/time_factored_fold.py dx
} > gpurun_out/r06/time_fold.log 2>&1
