set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
rm -f gpurun_out/r06/ln_variants.log
L=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib
for v in base nb32 nb64 nt nb32nt base2; do
  case $v in base*) unset LPM_HIP_LIBRARY;; *) export LPM_HIP_LIBRARY=$L/liblpm_hip_ln_$v.so;; esac
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/ln_$v
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ln_$v -o ln -- python3 $GRAFT_REPO_ROOT/tools/time_ln.py 20 1 > /dev/null 2>&1
  DB=$(find /tmp/ln_$v -name '*.db' | head -1)
  python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 2>/dev/null | grep "ln_" | cut -c1-100 | sed "s/^/$v /" >> $GRAFT_REPO_ROOT/gpurun_out/r06/ln_variants.log
  cd $GRAFT_REPO_ROOT
  timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bench', d['value'], d['ms_per_step'])" >> gpurun_out/r06/ln_variants.log
done
