# PMC passes over the a5 chain + K1 at cfg-2's video shape (tools/run_k2_only.py): one counter group per rocprofv3 run, nothing
# but --pmc beside it.  Writes gpurun_out/pmc_r02/<group>.csv (kernel, grid, counter, value: no truncation) and summary.txt.
# usage: bash tools/pmc_a5.sh <commit> [<output directory under gpurun_out, default pmc_r02>]     (LPM_VLAD_SOFTMAX_FUSED=1 in the environment:
# the chain with the softmax inside the aggregation kernel)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${2:-pmc_r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "commit ${1:-unknown}; target: python3 tools/run_k2_only.py 6 (B=80 T=300 D=1024 K=256, training-mode forward of the production chain: K1, assign_tiles2, K2 raw k-major, row scales)" > $OUT/summary.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rm -rf /tmp/pa_$i
  rocprofv3 --pmc $grp -d /tmp/pa_$i -o out --output-format csv -- python3 $R/tools/run_k2_only.py 6 > /tmp/pa.log 2>&1
  F=$(find /tmp/pa_$i -name '*counter_collection.csv' | head -1)
  python3 - "$F" "$OUT/pass$i.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    f.write("kernel,grid,counter,value\n")
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(",", ";")[-70:]
        if any(p in k for p in ("assign_tiles", "softmax_stats", "vlad_aggregate", "vlad_finalize", "vlad_row_scales", "tile_gemm", "split_")):
            f.write(f"{k},{r['Grid_Size']},{r['Counter_Name']},{r['Counter_Value']}\n")
PY
  python3 $R/tools/pmc_summary.py $F assign_tiles softmax_stats vlad_aggregate_tiles3 vlad_finalize2 vlad_row_scales tile_gemm_kernel >> $OUT/summary.txt
done
cat $OUT/summary.txt
