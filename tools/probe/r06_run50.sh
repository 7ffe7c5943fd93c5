cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "frame_sample" 2>&1 | tail -5 > gpurun_out/r06/fsplit_kernel_tests.log
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_trajectory.py -q -x -k "v2 or V2 or cfg3 or gradient_image or attention_half" 2>&1 | tail -5 > gpurun_out/r06/fsplit_model_tests.log
timeout 600 python bench.py --config cfg3 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench50_cfg3.json 2> gpurun_out/r06/bench50_cfg3.err
LPM_FRAME_SPLIT=0 timeout 600 python bench.py --config cfg3 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench50_cfg3_off.json 2> gpurun_out/r06/bench50_cfg3_off.err
timeout 600 python bench.py --config cfg3 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench50_cfg3_b.json 2> gpurun_out/r06/bench50_cfg3_b.err
LPM_FRAME_SPLIT=0 timeout 600 python bench.py --config cfg3 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench50_cfg3_off_b.json 2> gpurun_out/r06/bench50_cfg3_off_b.err
