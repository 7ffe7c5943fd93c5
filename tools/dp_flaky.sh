cd $GRAFT_REPO_ROOT
for xw in 1 0 1 0; do
  echo "== XWAVE=$xw"; LPM_PROJ_XWAVE=$xw timeout 900 python -m pytest tests/test_gpu_dp_trainer.py -q -s -k "real_trainer and blocks" 2>&1 | grep -E "^\[dp|passed|failed|Error"
done
