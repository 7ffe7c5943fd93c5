# PMC passes over the logits_bn attention kernels at cfg-3's video shape (tools/run_mha_bn_only.py): one counter group per rocprofv3 run.
# -> gpurun_out/<dir>/summary.txt (per kernel: the counters summed over its launches / launches)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_r03_mha_bn}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/summary.txt
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pm_$i
  rocprofv3 --pmc $grp -d /tmp/pm_$i -o out --output-format csv -- python3 $R/tools/run_mha_bn_only.py 4 > /tmp/pm.log 2>&1
  F=$(find /tmp/pm_$i -name '*counter_collection.csv' | head -1)
  python3 - "$F" >> $OUT/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "mha_" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    print(k[-60:], {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
done
cat $OUT/summary.txt
