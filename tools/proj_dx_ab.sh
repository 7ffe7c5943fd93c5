# projection kernels by rocprofv3 --kernel-trace: forward + dx (both forms, LPM_PROJ_DX_FORM) at the cfg-2 / cfg-5 shapes, the library dx beside them
cd $GRAFT_REPO_ROOT && timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "projection" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
export LPM_PROJ_DX_STREAM_MIN_N=512
for form in 2 1; do
 for shp in "80 270336 512" "128 524288 1024" "33 4112 512"; do
  rm -rf /tmp/pp; export LPM_PROJ_DX_FORM=$form
  rocprofv3 --kernel-trace -d /tmp/pp -o out -- python3 $GRAFT_REPO_ROOT/tools/time_proj.py $shp > /tmp/pp.log 2>&1
  echo "DX_FORM=$form $shp: $(grep 'max rel' /tmp/pp.log)"; python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/pp -name '*.db' | head -1) | grep -E 'proj_|Cijk' | cut -c1-130
 done
done
