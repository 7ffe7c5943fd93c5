cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
timeout 300 python tools/soak.py cfg5 1500
LPM_FOLD_DX=1 timeout 300 python tools/soak.py cfg5 1500
timeout 300 python tools/soak.py cfg2 1500
timeout 300 python tools/soak.py cfg3 1500
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/soak.txt
