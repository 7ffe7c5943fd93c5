# every default-on switch of the round A/B'd in process on ONE box (tools/ab_flags.py: median ms per step of 4 x 40 steps, alternating)
cd $GRAFT_REPO_ROOT
run() { echo "== $2: $1"; python tools/ab_flags.py $1 4 40 $2 2>&1 | grep -E "median" ; }
run ops.FFN_TILES cfg2
run ops.DIRECT_WGRAD cfg2
run audio_side_stream cfg2
run fused_encoder_blocks cfg2
run ops.DIRECT_WGRAD cfg5
run ops.PROJ_DX_STREAM_MIN_N=1024,4096 cfg5
run ops.FFN_MOD_FUSED cfg3
run ops.MHA_BN_ONEPASS cfg3
run ops.MHA_BN_MOMENTS cfg3
run ops.V2_SPLIT_COLUMNS cfg3
