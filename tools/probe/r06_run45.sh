cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 300 python tools/probe/r06_find_adds.py cfg2 2>&1 | grep -v amdgpu.ids | tail -34 > gpurun_out/r06/find_aten_cfg2.log
timeout 300 python tools/probe/r06_find_adds.py cfg5 2>&1 | grep -v amdgpu.ids | tail -24 > gpurun_out/r06/find_aten_cfg5.log
timeout 300 python tools/probe/r06_find_adds.py cfg3 2>&1 | grep -v amdgpu.ids | tail -34 > gpurun_out/r06/find_aten_cfg3.log
