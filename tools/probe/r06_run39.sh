cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r06/full_gpu_final.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r06/smoke_final.log 2>&1
timeout 600 python bench.py > gpurun_out/r06/bench_final_default.json 2> gpurun_out/r06/bench_final_default.err
timeout 900 python bench.py --config all > gpurun_out/r06/bench_final_all.jsonl 2> gpurun_out/r06/bench_final_all.err
