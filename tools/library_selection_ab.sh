# same-box A/B of FLAGS.library_gemm_selection (recorded hipBLASLt / rocBLAS solutions for PyTorch's fp32 GEMMs): whole steps, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for c in cfg5 cfg2 cfg3; do
    for v in 1 0; do
      echo "$c selection=$v $(LPM_LIBRARY_GEMM_SELECTION=$v python bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline --no-dispatch-count 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")"
    done
  done
done
