# Kernel-trace profiles of bench.py on the GPU box: default schedule and LPM_SINGLE_STREAM=1 -> gpurun_out/<tag>_{two,single}_stream.md
# usage: bash tools/profile_bench.sh <tag> [bench.py arguments, e.g. --config cfg5]
TAG=${1:-prof}
shift
R=$GRAFT_REPO_ROOT
mkdir -p $(dirname $R/gpurun_out/${TAG}_x)
cd /tmp && export TMPDIR=/tmp
for mode in single two; do
  if [ $mode = single ]; then export LPM_SINGLE_STREAM=1; else unset LPM_SINGLE_STREAM; fi
  rm -rf /tmp/pb_$mode
  rocprofv3 --kernel-trace -d /tmp/pb_$mode -o out -- python3 $R/bench.py --steps 40 --warmup 10 --spinup-seconds 1 --no-cpu-baseline --no-dispatch-count "$@" > /tmp/pb_$mode.log 2>&1
  DB=$(find /tmp/pb_$mode -name '*.db' | head -1)
  STEPS=$(grep '"metric"' /tmp/pb_$mode.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['spinup_steps'] + d['warmup'] + d['steps'])")
  python3 $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_${mode}_stream.md $STEPS > /dev/null
  sed -n 3p $R/gpurun_out/${TAG}_${mode}_stream.md
  grep '"metric"' /tmp/pb_$mode.log | tail -1 | cut -c1-200
done
