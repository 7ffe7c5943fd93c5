# _lib/liblpm_hip_projplain.so: today's library with the projection kernels' weight stream on the DEFAULT cache policy instead of nt (A/B)
cd $(dirname $0)/.. 
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops -DLPM_PJ_AUX=0 -c learnablepoolingmethods_amd/csrc/proj_gemm.hip -o /tmp/proj_plain.o
OBJS=$(ls learnablepoolingmethods_amd/_lib/*.o | grep -v proj_gemm.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/proj_plain.o -o learnablepoolingmethods_amd/_lib/liblpm_hip_projplain.so && ls -la learnablepoolingmethods_amd/_lib/liblpm_hip_projplain.so
