set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k "gradient_image or v2 or cfg3" 2>&1 | tail -12 > gpurun_out/r06/v2img_model_tests.log
timeout 900 python -m pytest tests/test_gpu_trajectory.py -q -x -k "v2 or V2" 2>&1 | tail -5 > gpurun_out/r06/v2img_traj_tests.log
bash tools/profile_bench.sh r06/r06b_cfg3 --config cfg3 > gpurun_out/r06/profile_cfg3_b.log 2>&1
