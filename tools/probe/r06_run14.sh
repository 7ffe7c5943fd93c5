set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
for v in u4 u1 u4b u1b; do
  case $v in u1*) export LPM_HIP_LIBRARY=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib/liblpm_hip_ln1.so;; *) unset LPM_HIP_LIBRARY;; esac
  rm -rf /tmp/ln_$v
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ln_$v -o ln -- python3 $GRAFT_REPO_ROOT/tools/time_ln.py 20 1 > /dev/null 2>&1
  DB=$(find /tmp/ln_$v -name '*.db' | head -1)
  python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 2>/dev/null | grep "ln_" | cut -c1-120 | sed "s/^/$v /" >> $GRAFT_REPO_ROOT/gpurun_out/r06/ln_unroll.log
done
cd $GRAFT_REPO_ROOT
unset LPM_HIP_LIBRARY
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 40 > gpurun_out/r06/bench14_u4.json 2> gpurun_out/r06/bench14_u4.err
LPM_HIP_LIBRARY=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib/liblpm_hip_ln1.so timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 40 > gpurun_out/r06/bench14_u1.json 2> gpurun_out/r06/bench14_u1.err
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 40 > gpurun_out/r06/bench14_u4b.json 2> gpurun_out/r06/bench14_u4b.err
LPM_HIP_LIBRARY=$GRAFT_REPO_ROOT/learnablepoolingmethods_amd/_lib/liblpm_hip_ln1.so timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 40 > gpurun_out/r06/bench14_u1b.json 2> gpurun_out/r06/bench14_u1b.err
