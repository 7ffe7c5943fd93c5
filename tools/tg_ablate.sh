# the anti-phase form (LPM_TG_AP=1) of the 128- / 256-row tile GEMM against the pipelined form: K1 and the dense shapes
for ap in 0 1; do for ns in 4 5; do echo -n "K1 AP=$ap NS=$ns: "; LPM_TG_AP=$ap LPM_TG_WIDE_NS=$ns python tools/run_k1_only.py 50 2>&1 | grep -E "assign_gemm_tiles_fwd"; done; done
for ap in 0 1; do for ns in 4 5; do echo "dense form 4 AP=$ap NS=$ns: "; LPM_TG_AP=$ap LPM_TG_WIDE4_NS=$ns python tools/bench_dense_tiles.py 4 2>&1 | grep -E "fwd|dx" | cut -c1-175; done; done
python -m pytest tests/test_gpu_kernels.py -q -x -k "assign_gemm or dense or ffn" 2>&1 | tail -3
LPM_TG_AP=1 python -m pytest tests/test_gpu_kernels.py -q -x -k "assign_gemm or dense or ffn" 2>&1 | tail -3
