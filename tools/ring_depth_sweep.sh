#!/bin/bash
# Ring depth of the pipelined tile GEMM forms (LDS-DMA steps in flight): K1 at cfg-2 (128-row form) and the 256-row dense form.
for ns in 4 5 6; do
  echo "== LPM_TG_WIDE_NS=$ns (K1, 128-row form)"
  LPM_TG_WIDE_NS=$ns python tools/run_k1_only.py 50 2>&1 | grep -E "assign_gemm_tiles_fwd"
done
for ns in 4 5; do
  echo "== LPM_TG_WIDE4_NS=$ns (dense 256-row form)"
  LPM_TG_WIDE4_NS=$ns python tools/bench_dense_tiles.py 4 2>&1 | grep -E "fwd|dx"
done
