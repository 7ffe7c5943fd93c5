cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e3
python bench.py --config all > gpurun_out/r04e3/bench_all.jsonl 2> gpurun_out/r04e3/bench_all.err; echo "bench all rc=$?"
# TunableOp on the fp32 library GEMMs of cfg-5 (MoE head, gating): does the per-shape search buy anything?
python bench.py --config cfg5 --no-cpu-baseline --no-dispatch-count 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('cfg5 base', d['ms_per_step'])"
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$GRAFT_REPO_ROOT/gpurun_out/r04e3/tunable_cfg5.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=60 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
python bench.py --config cfg5 --no-cpu-baseline --no-dispatch-count 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('cfg5 tuning run', d['ms_per_step'])"
export PYTORCH_TUNABLEOP_TUNING=0
python bench.py --config cfg5 --no-cpu-baseline --no-dispatch-count 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('cfg5 tuned', d['ms_per_step'])"
unset PYTORCH_TUNABLEOP_ENABLED
python bench.py --config cfg5 --no-cpu-baseline --no-dispatch-count 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('cfg5 base', d['ms_per_step'])"
cat gpurun_out/r04e3/tunable_cfg5*.csv | cut -c1-150
python -m pytest tests -q -x -m gpu 2>&1 | tail -5
