cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 300 python tools/probe/r06_find_adds.py cfg5 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r06/find_adds_cfg5.log
timeout 300 python tools/probe/r06_find_adds.py cfg3 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r06/find_adds_cfg3.log
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -q -x -k "bf16 or cfg5 or netvlad" 2>&1 | tail -3 > gpurun_out/r06/bn16_tests.log
timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench42_cfg5.json 2> gpurun_out/r06/bench42_cfg5.err
