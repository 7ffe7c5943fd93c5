cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r06/full_gpu_final3.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r06/smoke_final3.log 2>&1
timeout 600 python bench.py > gpurun_out/r06/bench_final3_default.json 2> gpurun_out/r06/bench_final3_default.err
timeout 900 python bench.py --config all > gpurun_out/r06/bench_final3_all.jsonl 2> gpurun_out/r06/bench_final3_all.err
bash tools/profile_bench.sh r06/r06d_cfg3 --config cfg3 > gpurun_out/r06/profile_cfg3_d.log 2>&1
bash tools/profile_bench.sh r06/r06d_cfg5 --config cfg5 > gpurun_out/r06/profile_cfg5_d.log 2>&1
