set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "factored" 2>&1 | tail -3 > gpurun_out/r06/fold_kernel_tests2.log
for d in 0 4 16 0; do
  LPM_FA_FOLD_DBG=$d timeout 200 python tools/time_factored_fold.py dx 2>&1 | grep "mode="
done > gpurun_out/r06/time_fold_dbg3.log 2>&1
LPM_FA_FOLD=2 timeout 200 python tools/time_factored_fold.py copy 2>&1 | grep "mode=" >> gpurun_out/r06/time_fold_dbg3.log
timeout 200 python tools/time_factored_fold.py copy 2>&1 | grep -A1 "mode=" >> gpurun_out/r06/time_fold_dbg3.log
timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k "cfg5" 2>&1 | tail -8 > gpurun_out/r06/fold_model_tests.log
timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench28_cfg5.json 2> gpurun_out/r06/bench28_cfg5.err
LPM_FOLD_DX=0 timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 40 > gpurun_out/r06/bench28_cfg5_off.json 2> gpurun_out/r06/bench28_cfg5_off.err
