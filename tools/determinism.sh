# two copies of the determinism check at a time on the one GPU (the contention the two-rank tests run under)
cd $GRAFT_REPO_ROOT
for args in "${@:-blocks 25 0}"; do
  timeout 900 python tools/determinism_check.py $args > /tmp/det_b.log 2>&1 &
  pid=$!
  timeout 900 python tools/determinism_check.py $args 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-12}
  wait $pid; echo "(second copy) $(grep -v amdgpu.ids /tmp/det_b.log | tail -${TAILN:-12})"
done
