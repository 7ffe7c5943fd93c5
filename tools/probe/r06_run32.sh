cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "factored or adam" 2>&1 | tail -3 > gpurun_out/r06/adam_kernel_tests2.log
{
timeout 200 python tools/time_factored_fold.py dx
LPM_FA_FOLD_DBG=4 timeout 200 python tools/time_factored_fold.py dx
timeout 200 python tools/time_factored_fold.py copy
timeout 200 python tools/time_factored_fold.py dx
} 2>&1 | grep -A1 "mode=" > gpurun_out/r06/time_fold4.log
timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k "cfg5" 2>&1 | tail -3 > gpurun_out/r06/fold_model_tests2.log
timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench32_cfg5.json 2> gpurun_out/r06/bench32_cfg5.err
LPM_FOLD_DX=0 timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench32_cfg5_off.json 2> gpurun_out/r06/bench32_cfg5_off.err
timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench32_cfg5_b.json 2> gpurun_out/r06/bench32_cfg5_b.err
