# PMC passes for K1's forward on plain bf16 tiles at cfg-5's video shape (assign_wide_kernel, csrc/assign_flat.hip); one counter group per
# pass, no trace domains beside --pmc.  usage: bash tools/k1_bf16_pmc.sh   -> gpurun_out/r05_pmc_k1_bf16.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/r05_pmc_k1_bf16.txt
: > $OUT
i=0
for grp in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
i=$((i+1))
rm -rf /tmp/pkb_$i
rocprofv3 --pmc $grp -d /tmp/pkb_$i -o out --output-format csv -- python3 $R/tools/k1_bf16_loop.py 6 > /tmp/pkb.log 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/pkb_$i -name "*counter_collection.csv") assign_wide_kernel >> $OUT
done
cat $OUT
