set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "k2_bf16_clip or netvlad_bf16_storage or assign_tiles or bf16" 2>&1 | tail -3 > gpurun_out/r06/at_tests.log
timeout 600 python tools/time_k2_bf16.py 15 2>&1 | grep -E "assign_tiles|clip" > gpurun_out/r06/time_at_nstep2.log
LPM_ASSIGN_TILES_NSTEP=1 timeout 600 python tools/time_k2_bf16.py 15 2>&1 | grep -E "assign_tiles" > gpurun_out/r06/time_at_nstep1.log
timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench19_cfg5.json 2> gpurun_out/r06/bench19_cfg5.err
LPM_ASSIGN_TILES_NSTEP=1 timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench19_cfg5_n1.json 2> gpurun_out/r06/bench19_cfg5_n1.err
