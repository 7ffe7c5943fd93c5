cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python bench.py > gpurun_out/r06/bench_final2_default.json 2> gpurun_out/r06/bench_final2_default.err
timeout 900 python bench.py --config all > gpurun_out/r06/bench_final2_all.jsonl 2> gpurun_out/r06/bench_final2_all.err
