cd /tmp && export TMPDIR=/tmp
for d in 0 5 6; do
  rm -rf /tmp/pp; LPM_PROJ_DBG=$d rocprofv3 --kernel-trace -d /tmp/pp -o out -- python3 $GRAFT_REPO_ROOT/tools/time_proj.py 80 270336 512 $PAD > /tmp/pp.log 2>&1
  echo "dbg=$d: $(python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/pp -name '*.db' | head -1) | grep -E 'proj_' | cut -c1-90)"
done
