# one steady-state step of a bench.py workload, kernel by kernel in launch order -> gpurun_out/<tag>_step_<cfg>.md
TAG=${1:-anatomy}; CFG=${2:-cfg2}
R=$GRAFT_REPO_ROOT; mkdir -p $(dirname $R/gpurun_out/${TAG}_x)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pa
rocprofv3 --kernel-trace -d /tmp/pa -o out -- python3 $R/bench.py --config $CFG --steps 20 --warmup 5 --spinup-seconds 1 --no-cpu-baseline --no-dispatch-count > /tmp/pa.log 2>&1
python3 $R/tools/rocpd_step.py $(find /tmp/pa -name '*.db' | head -1) $R/gpurun_out/${TAG}_step_${CFG}.md 3 | head -8
