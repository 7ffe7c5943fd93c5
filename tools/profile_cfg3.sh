# Kernel-trace profile of the cfg-3 (NetVladV2) training step -> gpurun_out/<tag>_cfg3.md      usage: bash tools/profile_cfg3.sh <tag>
TAG=${1:-prof}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc3
rocprofv3 --kernel-trace -d /tmp/pc3 -o out -- python3 $R/tools/bench_other_configs.py cfg3 > /tmp/pc3.log 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/pc3 -name '*.db' | head -1) $R/gpurun_out/${TAG}_cfg3.md > /dev/null
tail -1 /tmp/pc3.log
