# LPM_K1_DBG ablations of the flat K1 (assign_flat.hip) at cfg-2's shape: kernel durations from rocprofv3 --kernel-trace --stats
# 0 whole; 1 prologue + epilogue; 2 no stores; 4 no loads in the loop; 8 no MFMAs; 16 no fragment reads; combinations.
# usage: k1_flat_ablate.sh "<steps per barrier ...>" "<dbg values ...>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/k1_flat_ablate.txt
: > $out
for kb in ${1:-2}; do
for dbg in ${2:-0 1 3 2 4 8 16 12 20 24 28 30}; do
  rm -rf /tmp/kfa_$dbg
  LPM_K1_KB=$kb LPM_K1_DBG=$dbg rocprofv3 --kernel-trace --stats -d /tmp/kfa_$dbg -o out --output-format csv -- python3 $R/tools/k1_fwd_loop.py 40 > /tmp/kfa.log 2>&1
  f=$(find /tmp/kfa_$dbg -name '*kernel_stats.csv' | head -1)
  echo "kb=$kb dbg=$dbg $(grep assign_flat $f | cut -d, -f1-7 | cut -c1-200) $(grep 'err vs' /tmp/kfa.log | sed 's/.*err/err/')" >> $out
done
done
cat $out
