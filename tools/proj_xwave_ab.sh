# A/B of the projection forward's x-loader wave (LPM_PROJ_XWAVE=1|0), kernel durations from rocprofv3 --kernel-trace
cd $GRAFT_REPO_ROOT && timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "projection_skinny or factored" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for xw in 1 0; do
 for shp in "80 270336 512" "128 540672 1024" "33 4112 512"; do
  rm -rf /tmp/pp; export LPM_PROJ_XWAVE=$xw
  rocprofv3 --kernel-trace -d /tmp/pp -o out -- python3 $GRAFT_REPO_ROOT/tools/time_proj.py $shp > /tmp/pp.log 2>&1
  echo "XWAVE=$xw $shp: $(grep 'max rel' /tmp/pp.log)"; python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/pp -name '*.db' | head -1) | grep -E 'proj_fwd' | cut -c1-110
 done
done
