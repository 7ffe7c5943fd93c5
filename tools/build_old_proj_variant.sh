# _lib/liblpm_hip_oldproj.so: today's library with proj_gemm.hip as it was before the LDS-DMA rings were read with counted waits (commit d593125),
# for a same-box A/B (tools/proj_lib_ab.sh; LPM_PROJ_DX_STREAM_MIN_N=1024 restores the old routing of the cfg-2 input gradient too)
cd $(dirname $0)/.. && R=$PWD && T=$R/learnablepoolingmethods_amd/csrc/_old_proj_gemm.hip && git show d593125:learnablepoolingmethods_amd/csrc/proj_gemm.hip > $T
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops -c $T -o /tmp/old_proj_gemm.o; rm -f $T
OBJS=$(ls learnablepoolingmethods_amd/_lib/*.o | grep -v proj_gemm.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/old_proj_gemm.o -o learnablepoolingmethods_amd/_lib/liblpm_hip_oldproj.so && ls -la learnablepoolingmethods_amd/_lib/liblpm_hip_oldproj.so
