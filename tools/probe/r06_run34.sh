cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "bn or batch_norm or ffn_mod" 2>&1 | tail -3 > gpurun_out/r06/bn_tests.log
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_trajectory.py -q -x -k "v2 or V2 or cfg3" 2>&1 | tail -3 > gpurun_out/r06/bn_model_tests.log
timeout 600 python bench.py --config cfg3 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench34_cfg3.json 2> gpurun_out/r06/bench34_cfg3.err
LPM_BN_COL_CHUNKS=1 timeout 600 python bench.py --config cfg3 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench34_cfg3_c1.json 2> gpurun_out/r06/bench34_cfg3_c1.err
timeout 600 python bench.py --config cfg3 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench34_cfg3_b.json 2> gpurun_out/r06/bench34_cfg3_b.err
bash tools/profile_bench.sh r06/r06c_cfg3 --config cfg3 > gpurun_out/r06/profile_cfg3_c.log 2>&1
