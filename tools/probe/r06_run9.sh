set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python tools/time_k2_bf16.py 15 > gpurun_out/r06/time_k2_bf16_b.log 2>&1
LPM_VB_NS=4 timeout 600 python tools/time_k2_bf16.py 15 > gpurun_out/r06/time_k2_bf16_b_ns4.log 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -s -k "k2_bf16_clip or netvlad_bf16_storage" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | cut -c1-300 > gpurun_out/r06/clip16_tests4.log
timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench9_cfg5.json 2> gpurun_out/r06/bench9_cfg5.err
