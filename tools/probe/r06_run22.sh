set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "mha" 2>&1 | tail -8 > gpurun_out/r06/v2img_kernel_tests.log
timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k "gradient_image or v2 or cfg3" 2>&1 | tail -12 > gpurun_out/r06/v2img_model_tests.log
timeout 600 python bench.py --config cfg3 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench22_cfg3.json 2> gpurun_out/r06/bench22_cfg3.err
LPM_MHA_BN_GRAD_IMAGE=0 timeout 600 python bench.py --config cfg3 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench22_cfg3_off.json 2> gpurun_out/r06/bench22_cfg3_off.err
