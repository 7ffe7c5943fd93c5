# clip-wide K2 (vlad_clip.hip): ring depth x cache policy, full kernel and ablations (LPM_VC_DBG: 8 no MFMA, 4 no DMA, 2 no stores, 1 no loop)
python -m pytest tests/test_gpu_kernels.py -q -x -k "clip_wide" 2>&1 | tail -3
for ns in 4 3; do for nt in 1 0; do for d in 0; do
  echo -n "NS=$ns NT=$nt DBG=$d: "; LPM_VC_NS=$ns LPM_VC_NT=$nt LPM_VC_DBG=$d K2_FORMS=chain python tools/time_k2_forms.py 2>/dev/null | grep -E "clip|chain" | tr '\n' ' '; echo
done; done; done
for d in 8 4 2 1 16 12; do
  echo -n "NS=4 NT=1 DBG=$d: "; LPM_VC_DBG=$d K2_FORMS=none python tools/time_k2_forms.py 2>/dev/null | grep clip
done
