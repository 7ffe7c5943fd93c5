set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "mha" -s 2>&1 | tail -40 > gpurun_out/r06/mha_tests.log
timeout 300 python tools/time_mha_bwd.py 20 > gpurun_out/r06/time_mha_bwd.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06/prof_mha -o mha -- python3 $GRAFT_REPO_ROOT/tools/time_mha_bwd.py 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/r06/prof_mha | head
