set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -s -k "k2_bf16_clip or netvlad_bf16_storage" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | cut -c1-400 > gpurun_out/r06/clip16_tests.log
timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench6_cfg5.json 2> gpurun_out/r06/bench6_cfg5.err
LPM_VLAD_CLIP16=0 timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench6_cfg5_old.json 2> gpurun_out/r06/bench6_cfg5_old.err
LPM_VB_NS=4 timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench6_cfg5_ns4.json 2> gpurun_out/r06/bench6_cfg5_ns4.err
timeout 600 python tools/graph_ab.py cfg2 40 > gpurun_out/r06/graph_ab_cfg2.log 2>&1
timeout 600 python tools/graph_ab.py cfg5 40 > gpurun_out/r06/graph_ab_cfg5.log 2>&1
timeout 600 python tools/graph_ab.py cfg3 40 > gpurun_out/r06/graph_ab_cfg3.log 2>&1
