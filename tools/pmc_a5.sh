# PMC passes over the a5 chain (+ K1) of one BASELINE configuration (tools/run_k2_only.py): one counter group per rocprofv3 run, nothing
# but --pmc beside it.  Writes gpurun_out/<dir>/pass<i>.csv (kernel, grid, counter, value: no truncation), summary.txt and
# a5_hbm_traffic_<cfg>.json (tools/pmc_to_json.py: FETCH_SIZE x 2 + WRITE_SIZE per launch, the guide's gfx950 corrections).
# usage: bash tools/pmc_a5.sh <commit> [<output directory under gpurun_out, default pmc_r04_cfg2>] [cfg2|cfg3|cfg5] [hbm|all]
R=$GRAFT_REPO_ROOT
CFG=${3:-cfg2}
OUT=$R/gpurun_out/${2:-pmc_r04_$CFG}
WHAT=${4:-all}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "commit ${1:-unknown}; target: python3 tools/run_k2_only.py 6 $CFG (see its docstring for the shape and the chain)" > $OUT/summary.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  if [ $WHAT = hbm ] && [ $i -gt 2 ]; then break; fi
  rm -rf /tmp/pa_$i
  rocprofv3 --pmc $grp -d /tmp/pa_$i -o out --output-format csv -- python3 $R/tools/run_k2_only.py 6 $CFG > /tmp/pa.log 2>&1
  F=$(find /tmp/pa_$i -name '*counter_collection.csv' | head -1)
  python3 - "$F" "$OUT/pass$i.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    f.write("kernel,grid,counter,value\n")
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(",", ";")[-70:]
        if any(p in k for p in ("assign_tiles", "softmax_stats", "vlad_aggregate", "vlad_finalize", "vlad_row_scales", "vlad_kmajor", "vlad_clip", "tile_gemm", "split_", "frame_apply")):
            f.write(f"{k},{r['Grid_Size']},{r['Counter_Name']},{r['Counter_Value']}\n")
PY
  python3 $R/tools/pmc_summary.py $F assign_tiles softmax_stats vlad_aggregate_tiles3 vlad_kmajor vlad_clip vlad_finalize2 vlad_row_scales tile_gemm_kernel >> $OUT/summary.txt
done
python3 $R/tools/pmc_to_json.py $OUT $CFG ${1:-unknown} > $OUT/a5_hbm_traffic_$CFG.json
cat $OUT/summary.txt
cat $OUT/a5_hbm_traffic_$CFG.json
