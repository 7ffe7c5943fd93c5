set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r06/gpu_tests4.log
timeout 300 python tools/time_mha_bwd.py 20 > gpurun_out/r06/time_mha_bwd4.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06/prof_mha4 -o mha -- python3 $GRAFT_REPO_ROOT/tools/time_mha_bwd.py 5 > /dev/null 2>&1
