set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "k1 or assign or bf16" 2>&1 | tail -4 > gpurun_out/r06/k1_tests.log
timeout 300 python tools/k1_bf16_loop.py 300 --clock > gpurun_out/r06/k1_wide_lds_epilogue.log 2>&1
timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench21_cfg5.json 2> gpurun_out/r06/bench21_cfg5.err
