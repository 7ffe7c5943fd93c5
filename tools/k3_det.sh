cd $GRAFT_REPO_ROOT
for args in "${@:-16 32 1024 256 400}"; do
  timeout 900 python tools/k3_determinism.py $args > /tmp/k3_b.log 2>&1 &
  pid=$!
  timeout 900 python tools/k3_determinism.py $args 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-14}
  wait $pid; echo "(second copy) $(grep -v amdgpu.ids /tmp/k3_b.log | tail -${TAILN:-14})"
done
