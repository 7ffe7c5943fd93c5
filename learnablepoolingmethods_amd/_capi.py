"""ctypes binding of liblpm_hip.so (include/lpm_hip.h).

This is the only place the Python host code touches the C ABI.  There is no CPU fallback:
``load()`` raises if the library is missing, and every wrapper raises ``LpmError`` when the
library reports a non-zero status (bad argument, unsupported shape, launch failure).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "liblpm_hip.so")

LPM_VLAD_SOFTMAX = 1
LPM_VLAD_RESIDUAL = 2
LPM_VLAD_OUT_KMAJOR = 4
LPM_VLAD_NRM_RAW = 8
LPM_VLAD_OUT_BF16 = 16
LPM_VLAD_TILES_BF16 = 32
LPM_VLAD_NRM_BF16 = 64
LPM_VLAD_RAW_KMAJOR = 128
LPM_VLAD_DEBUG_FALLBACK = 256
LPM_VLAD_WIDE_ALL = 512
LPM_VLAD_WIDE_NONE = 1024

# symbol -> (restype, argtypes); kept in one table so tests can check it against the header
_f = C.c_void_p      # device pointer
_i = C.c_int
_l = C.c_int64
_s = C.c_size_t
_fl = C.c_float
SIGNATURES = {
    "lpm_version": (_i, []),
    "lpm_last_error": (C.c_char_p, []),
    "lpm_kernel_timing_enable": (None, [_i]),
    "lpm_kernel_timing_read": (_i, [_i, C.POINTER(C.c_float), _i]),
    "lpm_clock_sampler": (_i, [_f, _i, _i, _f]),
    "lpm_clock_marker": (_i, [_f, _i, _f]),
    "lpm_l2_normalize_rows": (_i, [_f, _l, _i, _f, _f]),
    "lpm_dequantize_l2_normalize": (_i, [_f, _f, _i, _i, _i, _fl, _fl, _f, _f]),
    "lpm_frame_stats_workspace_bytes": (_s, [_i, _i, _i]),
    "lpm_frame_stats": (_i, [_f, _f, _i, _i, _i, _i, _f, _f]),
    "lpm_frame_apply": (_i, [_f, _f, _i, _i, _i, _i, _f, _f, _f, _f]),
    "lpm_frame_apply_tiles": (_i, [_f, _f, _i, _i, _i, _i, _f, _f, _f, _f, _i, _f, _i, _f]),
    "lpm_frame_apply_tiles_split": (_i, [_f, _f, _i, _i, _i, _i, _f, _f, _f, _f, _f, _i, _f, _i, _f]),
    "lpm_frame_stats_nblk": (_i, [_i, _i]),
    "lpm_frame_bn_bwd": (_i, [_f, _l, _f, _f, _i, _i, _i, _i, _f, _f, _fl, _f, _f, _f, _s, _f]),
    "lpm_frame_bn_bwd_split": (_i, [_f, _l, _f, _l, _i, _f, _f, _i, _i, _i, _i, _f, _f, _fl, _f, _f, _f, _s, _f]),
    "lpm_bn_fold": (_i, [_f, _i, _i, _l, _f, _f, _fl, _fl, _f, _f, _f, _f, _f, _f, _f]),
    "lpm_assign_gemm_nblk": (_i, [_i]),
    "lpm_assign_gemm_fwd": (_i, [_f, _l, _f, _i, _i, _i, _i, _f, _f, _f]),
    "lpm_row_tiles_bytes": (_s, [_i, _i, _i]),
    "lpm_weight_tiles_bytes": (_s, [_i, _i]),
    "lpm_assign_gemm_tiles_supported": (_i, [_i, _i, _i]),
    "lpm_assign_gemm_tiles_nblk": (_i, [_i, _i]),
    "lpm_split_rows_tiles": (_i, [_f, _l, _i, _i, _i, _f, _f]),
    "lpm_split_weight_tiles": (_i, [_f, _i, _i, _i, _f, _f]),
    "lpm_assign_gemm_tiles_fwd": (_i, [_f, _f, _i, _i, _i, _i, _f, _f, _f]),
    "lpm_assign_gemm_tiles_bwd_dx": (_i, [_f, _f, _i, _i, _i, _i, _f, _l, _f]),
    "lpm_dense_tiles_fwd": (_i, [_f, _f, _i, _i, _i, _f, _l, _i, _f]),
    "lpm_dense_tiles_supported": (_i, [_i, _i, _i]),
    "lpm_dense_tiles_act_image_fwd": (_i, [_f, _f, _f, _i, _i, _i, _f, _f]),
    "lpm_dense_tiles_relu_bwd_workspace_bytes": (_s, [_i, _i]),
    "lpm_dense_tiles_relu_bwd_image": (_i, [_f, _f, _f, _i, _i, _i, _f, _f, _f, _s, _f]),
    "lpm_image_row_tiles": (_i, [_f, _i, _i, _i, _f, _f]),
    "lpm_skinny_weight_grad_tiles": (_i, [_f, _f, _i, _i, _i, _f, _f]),
    "lpm_assign_gemm_tiles_bwd_dw_workspace_bytes": (_s, [_i, _i, _i, _i]),
    "lpm_assign_gemm_tiles_bwd_dw": (_i, [_f, _f, _i, _i, _i, _i, _f, _f, _s, _f]),
    "lpm_vlad_aggregate_fwd": (_i, [_f, _f, _f, _f, _l, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f]),
    "lpm_vlad_finalize_fwd": (_i, [_f, _f, _i, _i, _i, _i, _f, _f, _f]),
    "lpm_xt_bytes": (_s, [_i, _i, _i]),
    "lpm_at_bytes": (_s, [_i, _i, _i]),
    "lpm_split_frames": (_i, [_f, _l, _i, _i, _i, _f, _f]),
    "lpm_assign_tiles": (_i, [_f, _f, _f, _i, _i, _i, _i, _f, _f]),
    "lpm_vlad_aggregate_tiles_fwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f]),
    "lpm_vlad_tiles3_supported": (_i, [_i, _i]),
    "lpm_vlad_aggregate_tiles3_fwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f]),
    "lpm_vlad_finalize2_fwd_ld": (_i, [_f, _f, _i, _i, _i, _i, _i, _f, _l, _f, _f, _f, _f]),
    "lpm_vlad_finalize2_fwd": (_i, [_f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f]),
    "lpm_proj_supported": (_i, [_i, _l, _i]),
    "lpm_proj_fwd_workspace_bytes": (_s, [_i, _l, _i]),
    "lpm_proj_fwd": (_i, [_f, _l, _f, _i, _l, _i, _f, _f, _s, _f]),
    "lpm_proj_fwd_parts": (_i, [_f, _l, _l, _i, _f, _i, _f, _l, _f, _i, _l, _i, _f, _f, _s, _f]),
    "lpm_split_weight_tiles_parts": (_i, [_f, _l, _l, _i, _f, _i, _f, _l, _i, _l, _f, _f]),
    "lpm_proj_dx": (_i, [_f, _f, _i, _l, _i, _f, _l, _f]),
    "lpm_proj_fwd_parts_w16": (_i, [_f, _l, _l, _f, _i, _f, _l, _f, _i, _l, _i, _f, _f, _s, _f]),
    "lpm_proj_dx_w16": (_i, [_f, _f, _i, _l, _i, _f, _l, _f]),
    "lpm_frame_tiles_bf16_bytes": (_s, [_i, _i, _i]),
    "lpm_frame_steps_bf16": (_i, [_i]),
    "lpm_frame_apply_tiles2": (_i, [_f, _f, _i, _i, _i, _i, _f, _f, _f, _f, _f, _i, _f, _f, _i, _f]),
    "lpm_frame_apply_tiles_bf16": (_i, [_f, _f, _i, _i, _i, _i, _f, _f, _f, _f, _f, _i, _f, _f, _i, _f]),
    "lpm_split_weight_tiles_bf16": (_i, [_f, _i, _i, _i, _f, _f]),
    "lpm_split_frames_bf16": (_i, [_f, _l, _i, _i, _i, _f, _f]),
    "lpm_assign_gemm_tiles_fwd_bf16": (_i, [_f, _f, _i, _i, _i, _i, _f, _f, _f]),
    "lpm_k1_forms_disable": (_i, [_i]),
    "lpm_mha_bwd_set_terms": (_i, [_i]),
    "lpm_assign_gemm_tiles_bwd_dw_bf16": (_i, [_f, _f, _i, _i, _i, _i, _f, _f, _s, _f]),
    "lpm_assign_tiles_bf16": (_i, [_f, _f, _f, _i, _i, _i, _i, _f, _f]),
    "lpm_vlad_aggregate_tiles3_fwd_bf16": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f]),
    "lpm_vlad_clip16_slabs": (_i, [_i, _i]),
    "lpm_vlad_aggregate_clip_fwd_bf16": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f]),
    "lpm_vlad_aggregate_raw_kmajor_fwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f]),
    "lpm_vlad_smx_stats_bytes": (_s, [_i, _i]),
    "lpm_vlad_smx_supported": (_i, [_i, _i, _i]),
    "lpm_vlad_aggregate_raw_kmajor_smx_fwd": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f]),
    "lpm_vlad_row_scales": (_i, [_f, _i, _i, _i, _f, _f, _f, _f, _f]),
    "lpm_vlad_clip_slabs": (_i, [_i, _i]),
    "lpm_vlad_aggregate_clip_kmajor_fwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f]),
    "lpm_vlad_aggregate_clip_dmajor_fwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f]),
    "lpm_vlad_kmajor_workspace_bytes": (_s, [_i, _i, _i]),
    "lpm_vlad_aggregate_kmajor_scaled_fwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _s, _f]),
    "lpm_split_rows_scaled": (_i, [_f, _l, _l, _i, _f, _f, _f]),
    "lpm_layer_norm_act_fwd_rs": (_i, [_f, _f, _i, _f, _f, _f, _f, _i, _i, _i, _fl, _f, _l, _f, _f, _f, _s, _f]),
    "lpm_layer_norm_act_image_fwd": (_i, [_f, _f, _i, _f, _f, _f, _f, _i, _i, _i, _fl, _f, _l, _f, _f, _f, _f, _s, _f]),
    "lpm_vlad_fused_supported": (_i, [_i, _i]),
    "lpm_vlad_fused_workspace_bytes": (_s, [_i, _i, _i]),
    "lpm_vlad_aggregate_fused_fwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _s, _f]),
    "lpm_vlad_bwd_workspace_bytes": (_s, [_i, _i, _i]),
    "lpm_vlad_aggregate_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _l, _f, _i, _i, _i, _i, _i, _f, _f, _l,
                                    _i, _f, _f, _s, _f]),
    "lpm_vlad_bwd_tiles_workspace_bytes": (_s, [_i, _i, _i, _i]),
    "lpm_vlad_aggregate_bwd_tiles_ld": (_i, [_f, _l, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f, _s, _f]),
    "lpm_vlad_aggregate_bwd_tiles": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f, _s, _f]),
    "lpm_input_bn_grads": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _f, _f, _f]),
    "lpm_vlad_aggregate_bwd_tiles_dx": (_i, [_f, _s, _f, _f, _i, _i, _i, _i, _f, _l, _i, _f]),
    "lpm_bn_rows_workspace_bytes": (_s, [_i, _i]),
    "lpm_bn_small_fwd": (_i, [_f, _i, _i, _f, _f, _fl, _fl, _i, _f, _f, _f, _f, _f, _f, _f]),
    "lpm_bn_small_bwd": (_i, [_f, _f, _i, _i, _f, _f, _f, _f, _i, _f, _f, _f, _f, _f, _f]),
    "lpm_mha_bn_corrections": (_i, [_f, _i, _i, _f, _f, _f, _fl, _l, _f, _f, _f, _f, _f]),
    "lpm_bn_rows_act_fwd": (_i, [_f, _f, _i, _i, _i, _f, _f, _fl, _fl, _i, _f, _f, _f, _f, _f, _f, _s, _f]),
    "lpm_bn_act_bwd_supported": (_i, [_i, _i]),
    "lpm_bn_act_bwd_workspace_bytes": (_s, [_i, _i]),
    "lpm_bn_act_bwd": (_i, [_f, _f, _f, _i, _f, _f, _f, _fl, _i, _i, _f, _f, _f, _f, _f, _s, _f]),
    "lpm_bn_rows_act_image_fwd": (_i, [_f, _f, _i, _i, _i, _f, _f, _fl, _fl, _i, _f, _f, _f, _f, _f, _f, _s, _f]),
    "lpm_bn_act_bwd_image": (_i, [_f, _f, _f, _i, _f, _f, _f, _fl, _i, _i, _f, _f, _f, _f, _f, _s, _f]),
    "lpm_bn_rows_fwd": (_i, [_f, _i, _i, _f, _f, _fl, _fl, _i, _f, _f, _f, _f, _f, _f, _s, _f]),
    "lpm_bn_bwd_workspace_bytes": (_s, [_i, _i]),
    "lpm_bn_bwd": (_i, [_f, _f, _f, _f, _f, _fl, _i, _i, _f, _f, _f, _f, _s, _f]),
    "lpm_bn_bwd_x16": (_i, [_f, _f, _f, _f, _f, _fl, _i, _i, _f, _f, _f, _f, _s, _f]),
    "lpm_split_rows": (_i, [_f, _l, _l, _i, _f, _i, _i, _f, _f]),
    "lpm_split_weight": (_i, [_f, _i, _i, _f, _f, _f]),
    "lpm_split_rows_relu_bwd_workspace_bytes": (_s, [_l, _i]),
    "lpm_bias_act_fwd": (_i, [_f, _f, _i, _l, _i, _f]),
    "lpm_bias_act_bwd_workspace_bytes": (_s, [_l, _i]),
    "lpm_bias_act_bwd": (_i, [_f, _f, _i, _l, _i, _f, _f, _f, _s, _f]),
    "lpm_split_rows_relu_bwd": (_i, [_f, _l, _i, _f, _f, _f, _f, _s, _f]),
    "lpm_layer_norm_workspace_bytes": (_s, [_i, _i]),
    "lpm_layer_norm_fwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _fl, _f, _f, _f, _f, _s, _f]),
    "lpm_layer_norm_bwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _f, _f, _f, _f, _s, _f]),
    "lpm_layer_norm_act_fwd": (_i, [_f, _f, _i, _f, _f, _f, _i, _i, _i, _fl, _f, _l, _f, _f, _f, _s, _f]),
    "lpm_layer_norm_pair_fwd": (_i, [_f, _f, _i, _f, _f, _f, _f, _f, _i, _i, _i, _fl, _f, _l, _f, _f, _f, _f, _f, _s, _f]),
    "lpm_layer_norm_act_bwd": (_i, [_f, _l, _f, _f, _f, _f, _f, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _s, _f]),
    "lpm_layer_norm_act_mask_image_fwd": (_i, [_f, _f, _i, _f, _fl, _f, _f, _f, _i, _i, _i, _fl, _f, _l, _f, _f, _f, _f, _s, _f]),
    "lpm_layer_norm_act_mask_bwd": (_i, [_f, _l, _f, _f, _f, _f, _f, _i, _f, _fl, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _s, _f]),
    "lpm_layer_norm_act_mask_bwd_fmt": (_i, [_f, _l, _f, _f, _f, _f, _f, _i, _f, _fl, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _s, _f, _f]),
    "lpm_mha_fwd": (_i, [_f, _f, _f, _l, _i, _i, _i, _i, _fl, _f, _f, _f, _l, _f, _f]),
    "lpm_mha_fwd_x3": (_i, [_f, _f, _f, _l, _i, _i, _i, _i, _fl, _f, _f, _f, _l, _f, _f]),
    "lpm_mha_bwd": (_i, [_f, _f, _f, _l, _f, _f, _l, _f, _i, _i, _i, _i, _fl, _f, _f, _f, _f, _f, _l, _f, _f, _f, _f]),
    "lpm_mha_bn_dk_correct": (_i, [_f, _f, _l, _i, _i, _i, _i, _fl, _f, _f, _f, _l, _f, _f]),
    "lpm_mha_logit_stats_moments": (_i, [_f, _f, _l, _i, _i, _i, _i, _f, _f, _f]),
    "lpm_mha_bwd_x3": (_i, [_f, _f, _f, _l, _f, _f, _l, _f, _i, _i, _i, _i, _fl, _f, _f, _f, _f, _f, _l, _f, _f, _f, _f]),
    "lpm_mha_bwd_x3_image": (_i, [_f, _f, _f, _l, _f, _i, _f, _l, _f, _i, _i, _i, _i, _fl, _f, _f]),
    "lpm_mha_fwd_x3_image": (_i, [_f, _f, _f, _l, _i, _i, _i, _i, _fl, _f, _f, _f]),
    "lpm_mha_logit_stats_workspace_bytes": (_s, [_i, _i, _i]),
    "lpm_mha_logit_stats": (_i, [_f, _f, _l, _i, _i, _i, _i, _f, _f]),
    "lpm_moe_ce_nblk": (_i, [_i, _i]),
    "lpm_moe_ce_fwd": (_i, [_f, _f, _f, _i, _i, _i, _fl, _f, _f, _f, _f]),
    "lpm_moe_ce_bwd": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _fl, _f, _f, _f]),
    "lpm_clip_adam_scratch_bytes": (_s, [_l, _i]),
    "lpm_factored_clip_adam_scratch_bytes": (_s, [_i, _i]),
    "lpm_factored_clip_adam_q": (_i, [_f, _f, _f, _l, _f, _i, _i, _i, _f, _f, _f, _fl, _fl, _fl, _fl, _fl, _l, _f, _s, _f]),
    "lpm_factored_clip_adam_copy": (_i, [_f, _f, _f, _l, _f, _i, _i, _i, _f, _f, _f, _f, _fl, _fl, _fl, _fl, _fl, _l, _f, _s, _f]),
    "lpm_factored_clip_adam_copy_dx": (_i, [_f, _f, _f, _l, _f, _i, _i, _i, _f, _f, _f, _f, _f, _f, _l, _fl, _fl, _fl, _fl, _fl, _l, _f, _s, _f]),
    "lpm_factored_fold_supported": (_i, [_i, _i, _i]),
    "lpm_factored_clip_adam": (_i, [_f, _f, _i, _i, _i, _f, _f, _f, _fl, _fl, _fl, _fl, _fl, _l, _f, _s, _f]),
    "lpm_multi_tensor_clip_adam": (_i, [_f, _f, _f, _f, _f, _i, _l, _fl, _fl, _fl, _fl, _fl, _l, _f, _f]),
    "lpm_dropout_keep_mask": (_i, [_f, _l, _fl, C.c_uint64, _f]),
    "lpm_multi_tensor_clip_adam_l2": (_i, [_f, _f, _f, _f, _f, _f, _i, _l, _fl, _fl, _fl, _fl, _fl, _l, _f, _f]),
    "lpm_weight_pack": (_i, [_f, _i, _f]),
    "lpm_sum_splits": (_i, [_f, _i, _i, _i, _f, _f, _f, _i, _f]),          # (jobs: a HOST array of WeightPackJob)
    # round 5: the entry points that take an operand format (LpmOperandFormat*: a HOST struct, NULL = split-bf16 x3)
    "lpm_split_rows_fmt": (_i, [_f, _l, _l, _i, _f, _i, _i, _f, _f, _f, _f]),
    "lpm_split_rows_relu_bwd_fmt": (_i, [_f, _l, _i, _fl, _f, _i, _f, _f, _f, _s, _f, _f]),
    "lpm_split_weight_fmt": (_i, [_f, _i, _i, _f, _f, _i, _f]),
    "lpm_split_weight_tiles_fmt": (_i, [_f, _i, _i, _i, _f, _i, _f]),
    "lpm_split_rows_tiles_fmt": (_i, [_f, _l, _i, _i, _i, _f, _f, _f]),
    "lpm_image_row_tiles_fmt": (_i, [_f, _i, _i, _i, _f, _i, _f]),
    "lpm_dense_tiles_act_image_fwd_fmt": (_i, [_f, _i, _f, _f, _i, _i, _i, _fl, _f, _f, _f]),
    "lpm_dense_tiles_relu_bwd_image_fmt": (_i, [_f, _i, _f, _f, _i, _i, _i, _i, _fl, _f, _f, _f, _s, _f, _f]),
    "lpm_layer_norm_act_image_fwd_fmt": (_i, [_f, _f, _i, _f, _f, _f, _f, _i, _i, _i, _fl, _f, _l, _f, _f, _f, _f, _s, _f, _f]),
    "lpm_layer_norm_act_mask_image_fwd_fmt": (_i, [_f, _f, _i, _f, _fl, _f, _f, _f, _i, _i, _i, _fl, _f, _l, _f, _f, _f, _f, _s, _f, _f]),
    "lpm_layer_norm_act_bwd_fmt": (_i, [_f, _l, _f, _f, _f, _f, _f, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _s, _f, _f]),
    "lpm_mha_fwd_x3_image_fmt": (_i, [_f, _f, _f, _l, _i, _i, _i, _i, _fl, _f, _f, _f, _f]),
    "lpm_mha_bwd_x3_image_fmt": (_i, [_f, _f, _f, _l, _f, _f, _f, _l, _f, _i, _i, _i, _i, _fl, _f, _f, _f]),
    "lpm_sum_splits_scaled": (_i, [_f, _i, _i, _i, _i, _fl, _f, _f, _f, _i, _f]),
    "lpm_mha_bwd_x3_bn_image_fmt": (_i, [_f, _f, _f, _l, _f, _f, _l, _f, _i, _i, _i, _i, _fl, _f, _f, _f, _f, _f, _f, _f, _f, _f]),
    "lpm_mha_bn_dk_correct_image": (_i, [_f, _f, _l, _i, _i, _i, _i, _fl, _f, _f, _f, _f, _f, _f, _f]),
    "lpm_bn_rows_act_image_fwd_fmt": (_i, [_f, _f, _i, _i, _i, _f, _f, _fl, _fl, _i, _f, _f, _f, _f, _f, _f, _s, _f, _f]),
    "lpm_bn_act_bwd_image_fmt": (_i, [_f, _f, _f, _i, _f, _f, _f, _fl, _i, _i, _f, _f, _f, _f, _f, _s, _f, _f]),
}
LPM_OPERAND_BF16X3 = 0
LPM_OPERAND_FP16X2 = 1
LPM_OPERAND_FP16X3 = 2
LPM_OPERAND_AMAX_SUB = 32        # sub-slots of a site's max |x| record, LPM_OPERAND_AMAX_STRIDE floats apart (include/lpm_hip.h)
LPM_OPERAND_AMAX_STRIDE = 16


class OperandFormat(C.Structure):
    """LpmOperandFormat of include/lpm_hip.h (a HOST struct handed over by pointer; read during the call)."""
    _fields_ = [("kind", C.c_int), ("scale", C.c_float), ("amax", C.c_void_p)]
WEIGHT_PACK_MAX_JOBS = 24


class WeightPackJob(C.Structure):
    """LpmWeightPackJob of include/lpm_hip.h."""
    _fields_ = [("w", C.c_void_p), ("K", C.c_int), ("N", C.c_int), ("ldw", C.c_int), ("Ntot", C.c_int), ("n_off", C.c_int),
                ("w3n", C.c_void_p), ("w3k", C.c_void_p), ("wt", C.c_void_p), ("wtt", C.c_void_p), ("kind", C.c_int)]


class LpmError(RuntimeError):
    pass


class _Lib:
    def __init__(self, path: str):
        if not os.path.exists(path):
            raise LpmError(
                f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (needs hipcc). There is no CPU fallback for the hot path.")
        self._dll = C.CDLL(path)
        self.path = path
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self._dll, name)  # AttributeError if the symbol is missing -> loud failure
            fn.restype = res
            fn.argtypes = args
            setattr(self, "_" + name, fn)

    def last_error(self) -> str:
        return (self._lpm_last_error() or b"").decode()

    def check(self, status: int, what: str):
        if status != 0:
            raise LpmError(f"{what} failed with status {status}: {self.last_error()}")

    def version(self) -> int:
        return self._lpm_version()


_LIB: Optional[_Lib] = None


def load() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib(os.environ.get("LPM_HIP_LIBRARY") or LIB_PATH)      # (the override: A/B of two builds of the library, tools/)
    return _LIB


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a tensor (None -> NULL).  Raises unless the tensor is fp32/int on a GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise LpmError("lpm ops need tensors on an MI355X (cuda/hip device); got a CPU tensor. "
                       "There is no CPU fallback for the hot path.")
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    if not torch.cuda.is_available():
        raise LpmError("lpm ops need an MI355X (no HIP device is visible). There is no CPU fallback for the hot path.")
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
