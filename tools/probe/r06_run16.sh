set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python bench.py --config all > gpurun_out/r06/bench_all.jsonl 2> gpurun_out/r06/bench_all.err
timeout 600 python bench.py > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err
timeout 300 python __graft_entry__.py smoke > gpurun_out/r06/smoke.log 2>&1
