set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r06/gpu_tests5.log
timeout 900 python -m pytest "tests/test_gpu_trajectory.py::test_the_eighty_clip_single_batch_run_freezes_for_the_oracle_as_for_the_product" -x -q -s 2>&1 | grep -E "^\[traj|passed|failed|Error|assert" > gpurun_out/r06/traj_b80.log
