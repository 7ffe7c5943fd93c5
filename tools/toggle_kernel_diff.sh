cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tk
rocprofv3 --kernel-trace -d /tmp/tk -o out -- python3 $GRAFT_REPO_ROOT/tools/dx_toggle_probe.py cfg2 1 > /tmp/tk.log 2>&1
tail -1 /tmp/tk.log
python3 $GRAFT_REPO_ROOT/tools/toggle_kernel_diff.py $(find /tmp/tk -name '*.db' | head -1)
