for d in 0 1 2 8 14; do echo "dbg=$d"; LPM_VK_DBG=$d python tools/time_k2_forms.py 2>&1 | grep -E "none|all|rounds|chain"; done
