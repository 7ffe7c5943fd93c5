# PMC passes for K1's forward (flat 96-row form, assign_flat.hip) at cfg-2's shape; one counter group per pass (no trace domains beside --pmc)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
: > $R/gpurun_out/pmc_k1w_summary.txt
i=0
for grp in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
i=$((i+1))
rm -rf /tmp/pk_$i
rocprofv3 --pmc $grp -d /tmp/pk_$i -o out --output-format csv -- python3 $R/tools/k1_fwd_loop.py 6 > /tmp/pk.log 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/pk_$i -name "*counter_collection.csv") assign_flat_kernel >> $R/gpurun_out/pmc_k1w_summary.txt
done
cat $R/gpurun_out/pmc_k1w_summary.txt
