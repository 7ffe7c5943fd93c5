cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r06/full_gpu_final2.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r06/smoke_final2.log 2>&1
