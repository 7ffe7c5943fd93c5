# The flat K1 (assign_flat.hip) at cfg-2's shape under rocprofv3 --kernel-trace --stats: kernel durations for the A/B switches
# LPM_K1_KB (reduction steps per barrier: 1, 2, 4), LPM_K1_DB (steps of B fragments in flight: 4, or 8 with KB = 4), LPM_K1_LUMP (1: the
# fragment reads as one group behind the first MFMAs instead of one between consecutive MFMAs), and against the
# 128-row tile-GEMM form (LPM_K1_FLAT=0).  (The ablation switches the DESIGN.md numbers come from -- no main loop / no stores / no loads /
# no MFMAs / no fragment reads -- were compiled in for the measurement and removed again; every run below has its own time limit.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/k1_flat_ablate.txt
: > $out
for cfg in "1 2 4 0" "1 2 4 1" "1 1 4 0" "1 4 4 0" "1 4 8 0" "0 2 4 0"; do
  set -- $cfg
  rm -rf /tmp/kfa
  LPM_K1_FLAT=$1 LPM_K1_KB=$2 LPM_K1_DB=$3 LPM_K1_LUMP=$4 timeout 90 rocprofv3 --kernel-trace --stats -d /tmp/kfa -o out --output-format csv -- python3 $R/tools/k1_fwd_loop.py 40 > /tmp/kfa.log 2>&1
  f=$(find /tmp/kfa -name '*kernel_stats.csv' | head -1)
  echo "flat=$1 kb=$2 db=$3 lump=$4 $(grep 'assign_flat\|tile_gemm_kernel' $f | cut -d, -f1-7 | cut -c1-200) $(grep 'err vs' /tmp/kfa.log | sed 's/.*err/err/')" >> $out
done
cat $out
