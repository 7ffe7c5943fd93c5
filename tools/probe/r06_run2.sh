set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "mha" -s 2>&1 | tail -40 > gpurun_out/r06/mha_tests2.log
timeout 300 python tools/time_mha_bwd.py 20 > gpurun_out/r06/time_mha_bwd2.log 2>&1
LPM_MHA_BWD_TPW=1 timeout 300 python tools/time_mha_bwd.py 20 > gpurun_out/r06/time_mha_bwd2_tpw1.log 2>&1
LPM_MHA_BWD_TPW=2 timeout 300 python tools/time_mha_bwd.py 20 > gpurun_out/r06/time_mha_bwd2_tpw2.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06/prof_mha2 -o mha -- python3 $GRAFT_REPO_ROOT/tools/time_mha_bwd.py 5 > /dev/null 2>&1
export LPM_MHA_BWD_TPW=2
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06/prof_mha2_tpw2 -o mha -- python3 $GRAFT_REPO_ROOT/tools/time_mha_bwd.py 5 > /dev/null 2>&1
