"""torch.autograd bindings of the HIP hot path (liblpm_hip.so, include/lpm_hip.h).

Every op here launches hand-written gfx950 kernels on the current PyTorch HIP stream through the
C ABI.  There is NO CPU fallback: CPU tensors or a missing library raise ``LpmError``.
PyTorch only supplies device memory, streams and the autograd tape.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional

import torch

from . import _capi
from ._capi import LPM_VLAD_NRM_RAW, LPM_VLAD_OUT_KMAJOR, LPM_VLAD_RESIDUAL, LPM_VLAD_SOFTMAX, LpmError, ptr, stream_ptr

BN_EPS = 1e-3     # slim.batch_norm epsilon
BN_DECAY = 0.999  # slim.batch_norm decay

# Matrix-core arithmetic of the aggregation kernel K2:
#   "bf16x3": split-bf16 (hi/lo planes, 3 bf16 MFMAs per product, ~1e-5 relative error)  [default]
#   "f32":    exact fp32 MFMA (v_mfma_f32_32x32x2_f32, ~1e-7), 5.3x more matrix-pipe time
VLAD_PRECISION = os.environ.get("LPM_VLAD_PRECISION", "bf16x3")
# bf16x3 only: use the LDS-shared 128x128 workgroup form (vlad_tiles3.hip) where the shape allows (D, K multiples of
# 128); otherwise / when off, the register-streaming form (vlad_tiles.hip).
VLAD_TILES3 = os.environ.get("LPM_VLAD_TILES3", "1") != "0"
# ... with the finalize pass fused into the aggregation kernel (lpm_vlad_aggregate_fused_fwd).  Off by default: the descriptor is
# written once and the un-normalised sums never make a round trip (chain traffic 1.15x algorithmic instead of 2.2x), but the
# per-clip wait inside the kernel costs what the second launch cost -- measured at cfg-2 (tools/time_a5.py, whole a5 chain):
# 107 us fused vs 107 us two-pass in inference, 134 vs 108 us when the backward's copy of U is stored as well; same picture at
# B = 128, K = 512 (341 / 396 vs 340 us).  All workgroups reach their epilogue together (1.7 rounds of 768 resident workgroups),
# so the wait overlaps nothing.
VLAD_FUSED = os.environ.get("LPM_VLAD_FUSED", "0") == "1"
# a9: the hidden projection's forward / input gradient as hand-written weight-stream kernels (csrc/proj_gemm.hip); "0": library GEMMs (A/B)
PROJ_STREAM = os.environ.get("LPM_PROJ_STREAM", "1") != "0"
# ... the input-gradient kernel from this hidden size on (rocprofv3 kernel durations, round 4, the rings read without the compiler's
# vmcnt(0): forward 107 + 22 us vs the library's 202 us at cfg-2, 480 vs 1060 us at cfg-5; dx 135 vs 212 us at cfg-2's N = 512, 455 vs 1000 us
# at cfg-5's N = 1024.  Rounds 1-3, every stage waiting a full memory round trip: forward 145 / 600, dx 267 -- the library stayed -- / 634)
PROJ_DX_STREAM_MIN_N = int(os.environ.get("LPM_PROJ_DX_STREAM_MIN_N", "512"))
# ... and inside the channel-last batch norm that follows it (ops.batch_norm_rows_act: FeedForwardNetworkMod); "0": ops.bias_act + plain BN
BN_ACT_FUSED = os.environ.get("LPM_BN_ACT_FUSED", "1") != "0"
# tf.layers.dense's bias add + ReLU as one in-place pass (ops.bias_act) where the output does not feed the next GEMM's split directly; "0": A/B
BIAS_ACT_FUSED = os.environ.get("LPM_BIAS_ACT_FUSED", "1") != "0"
# K5 of hidden1_weights: the gradient's norm from quadratic forms (lpm_factored_clip_adam_q) instead of a tile-GEMM pass; "0": A/B
FACTORED_NORM_QUADFORM = os.environ.get("LPM_FACTORED_NORM_QUADFORM", "1") != "0"
# a5 with the softmax inside the aggregation kernel (lpm_vlad_aggregate_raw_kmajor_smx_fwd; the lazily normalised k-major descriptor
# of the NetVladV1 video stream): no assignment tiles, chain traffic 1.30x algorithmic instead of 1.41x (PMC, profiles/pmc_r02_smx) -- and bit for bit the
# two-kernel chain's result.  Off by default, it is slower: measured at cfg-2 in the training step (kernel durations) row statistics
# 8.7 us + aggregation 76.4 us + row scales 5.4 us = 90.5 us against 11.8 + 61.5 + 5.0 = 79 us for lpm_assign_tiles +
# lpm_vlad_aggregate_raw_kmajor_fwd.  The aggregation loop is bound by LDS bandwidth next to the L2 -> LDS delivery (48 KB of
# fragment reads per 16 KB staged), and building the A fragments in the kernel adds 8 KB of DMA writes, 8 KB of reads and 8 KB of
# stores per step (+37 %); the third workgroup per CU it also loses (78 KB of LDS) is worth only 2 us (LPM_K2_EXTRA_LDS=27000 on
# the two-kernel form: 61.5 -> 63.8 us).
VLAD_SOFTMAX_FUSED = os.environ.get("LPM_VLAD_SOFTMAX_FUSED", "0") == "1"
# a2 + a3 also emit K1's split-bf16 row tiles (lpm_frame_apply_tiles2) instead of a separate lpm_split_rows_tiles pass per stream.
# Off by default: measured at cfg-2 the step is 8.65 ms with it and 8.59 ms without -- the separate pass (26 us) leaves its 50 MB of
# tiles in the 256 MB infinity cache right before K1 reads them (K1 44 us), whereas the fused pass writes 300 MB (fp32 matrix, frame
# tiles, row tiles) and K1 then streams its operand from HBM (53 us), and the fused pass itself grows by the extra stores.
FRAME_ROW_TILES = os.environ.get("LPM_FRAME_ROW_TILES", "0") == "1"
VLAD_FUSED_DEBUG_FALLBACK = False     # tests: drive every clip through the fused kernel's time-out path + follow-up finalize

# The lazily normalised k-major descriptor from ONE launch (lpm_vlad_aggregate_kmajor_scaled_fwd, vlad_kmajor.hip: K2 on wide 256 x 128
# workgroup items + the row scales by the clip's last workgroup).  Built and measured in round 3, NOT faster than the two-launch chain
# lpm_vlad_aggregate_raw_kmajor_fwd + lpm_vlad_row_scales (78 vs 74 us on one box, tools/time_k2_forms.py): off.
VLAD_KMAJOR_SCALED = os.environ.get("LPM_VLAD_KMAJOR_SCALED", "0") == "1"
# K2 on clip-wide items (lpm_vlad_aggregate_clip_kmajor_fwd, vlad_clip.hip, round 4; K = 256): all clusters x a third of a clip's
# columns per workgroup -- 168 MB through the LDS-DMA path instead of 389 MB.  "0": the 128 x 128 form (A/B).
VLAD_CLIP = os.environ.get("LPM_VLAD_CLIP", "1") != "0"
# bf16 storage: K2 on clip-wide items (csrc/vlad_clip16.hip, round 6) where the shape allows; "0": the 128 x 128 form everywhere (A/B)
VLAD_CLIP16 = os.environ.get("LPM_VLAD_CLIP16", "1") != "0"
# FeedForwardNetwork's first dense layer and its backward on the hand-written 256-row tile GEMM with operand-image epilogues
# (lpm_dense_tiles_act_image_fwd / lpm_dense_tiles_relu_bwd_image) where the shape allows; 0: library GEMM + separate split passes (A/B).
FFN_TILES = os.environ.get("LPM_FFN_TILES", "1") != "0"
# Run-time switch of the direct weight-gradient writes (ops._dw_x3 -> the trainer's arena slots, FLAGS.direct_weight_gradients): False
# returns every gradient through autograd (A/B inside one process: tools/ab_flags.py ops.DIRECT_WGRAD).
DIRECT_WGRAD = True
# FeedForwardNetworkMod (NetVladV2's frame encoders) as one node whose batch norm writes the second dense layer's operand image and
# whose backward writes the first layer's gradient image (ops.ffn_mod_x3); 0: dense / batch_norm / dense as separate nodes (A/B).
FFN_MOD_FUSED = os.environ.get("LPM_FFN_MOD_FUSED", "1") != "0"
# Matrix-core arithmetic of the soft-assignment GEMM K1: "bf16x3" (split-bf16 tiles on the bf16 pipe, default where
# D %% 16 == 0 and K <= 512) or "f32" (exact fp32 MFMA).
ASSIGN_PRECISION = os.environ.get("LPM_ASSIGN_PRECISION", "bf16x3")

# Matrix-core arithmetic of the attention core K4: "bf16x3" (split-bf16 operands, v_mfma_f32_16x16x32_bf16) or "f32".
MHA_PRECISION = os.environ.get("LPM_MHA_PRECISION", "bf16x3")
# The logits_bn variant (MultiHeadAttentionBN, NetVladV2) runs its FORWARD in exact fp32 and both backward passes on the bf16
# pipe ("mixed"): the attention output goes straight into attention_bn, a batch norm over nearly constant columns (softmax
# close to uniform, outputs close to the mean of v), which amplifies the 1e-5 element error of a split-bf16 forward ~500x --
# 1e-2 on whole-model gradients in the small NetVladV2 parity case, while the same arithmetic in the backward passes alone
# stays at the fp32 kernels' 2e-4 (tests/diagnostics/debug_v2_grad.py).  "f32" / "bf16x3": every pass in that arithmetic.
# Round 3: that amplification is a property of SHORT sequences (the small parity case has 12 keys: softmax over 12 nearly equal logits,
# attention_bn over columns whose variance is at rounding level -- still 1e-2 with a split-bf16 forward, debug_v2_grad.py).  At the
# benchmark's sizes (300 keys, cfg-3) a split-bf16 forward changes nothing measurable: worst whole-model gradient against the fp64 oracle
# 4.8e-5 (bf16x3 forward) vs 5.1e-5 (fp32 forward) at 80 clips (tests/test_gpu_benched_shapes.py), the whole GPU tier green either way,
# and the step is 0.2 ms shorter.  "auto" for the forward pass = exact fp32 below MHA_BN_X3_MIN_KEYS keys, split-bf16 from there.
MHA_BN_PRECISION = os.environ.get("LPM_MHA_BN_PRECISION", "mixed")
# "mixed": arithmetic per pass (forward, backward statistics pass, backward main pass)
MHA_BN_MIXED = os.environ.get("LPM_MHA_BN_MIXED", "auto,bf16x3,bf16x3")
MHA_BN_X3_MIN_KEYS = int(os.environ.get("LPM_MHA_BN_X3_MIN_KEYS", "128"))

# Split-bf16 tile copies of the most recent frame_sample_bn output (produced by the same kernel that writes the fp32
# frames): {"base": weakref to y, "F": F, "Dv": .., "video": tensor, "audio": tensor|None}.  ops.netvlad / vlad_aggregate
# look a column-slice view of y up here instead of re-reading it through lpm_split_frames.
_XT_CACHE = {}
MHA_BN_MOMENTS = os.environ.get("LPM_MHA_BN_MOMENTS", "1") != "0"      # ... statistics by lpm_mha_logit_stats_moments, moments kept for the backward
MHA_BN_ONEPASS = os.environ.get("LPM_MHA_BN_ONEPASS", "1") != "0"      # logits_bn backward without the separate statistics pass (A/B switch)
LN_IMAGE = os.environ.get("LPM_LN_IMAGE", "1") != "0"              # the attention block's layer norm also writes the operand image of the feed-forward block behind it (A/B switch)
SPLIT_VECTOR = True          # NetVladV1: input_bn's gamma / beta halves with ONE concatenated gradient each (A/B switch)
V2_SPLIT_COLUMNS = True      # NetVladV2: the two streams' inputs as contiguous copies with ONE concatenated gradient (A/B switch)
DEBUG_TAP = None      # tools/determinism_check.py: a dict that the video stream's pooling backward fills with copies of its intermediates

# bench.py sets this to a list to collect (name, dims, start_event, end_event) around hot-kernel launches
# on the current stream (HIP events; nothing is recorded or synchronised when it is None).
KERNEL_TIMELINE = None


class _timed:
    def __init__(self, name, dims):
        self.name, self.dims = name, dims

    def __enter__(self):
        if KERNEL_TIMELINE is not None:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t1 = torch.cuda.Event(enable_timing=True)
            self.t0.record()
        return self

    def __exit__(self, *exc):
        if KERNEL_TIMELINE is not None:
            self.t1.record()
            KERNEL_TIMELINE.append((self.name, self.dims, self.t0, self.t1))
        return False


def _f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise LpmError(f"{what}: expected float32, got {t.dtype}")
    return t


def _rows(t: torch.Tensor, what: str) -> torch.Tensor:
    """2-D tensor whose rows are contiguous (a column slice of a wider row-major matrix is fine)."""
    _f32(t, what)
    if t.dim() != 2 or t.stride(1) != 1:
        raise LpmError(f"{what}: need a 2-D tensor with unit column stride")
    return t


def _materialised(t: torch.Tensor, what: str) -> torch.Tensor:
    """Refuse to read the fp32 handle of frames that frame_sample_bn(storage='bf16', materialize=False) wrote as tiles only."""
    c = _XT_CACHE
    if c and c.get("unmaterialised") is not None and c["base"]() is not None and t.untyped_storage().data_ptr() == c["unmaterialised"]:
        raise LpmError(f"{what}: these frames exist as bf16 operand tiles only (frame_sample_bn(storage='bf16', materialize=False))")
    return t


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# ----------------------------------------------------------------------------------------------
# batch-norm statistics -> folded affine
# ----------------------------------------------------------------------------------------------
def bn_fold(partial, nblk, C, rows, gamma, beta, moving_mean=None, moving_var=None, eps=BN_EPS, decay=BN_DECAY):
    lib = _capi.load()
    mean, var, scale, shift = (_empty((C,), partial) for _ in range(4))
    lib.check(lib._lpm_bn_fold(ptr(partial), nblk, C, rows, ptr(gamma), ptr(beta), eps, decay, ptr(mean), ptr(var),
                               ptr(scale), ptr(shift), ptr(moving_mean), ptr(moving_var), stream_ptr()), "lpm_bn_fold")
    return mean, var, scale, shift


class _BNSmall(torch.autograd.Function):
    """Training-mode slim.batch_norm of a small [M, C] matrix fused with what follows it (lpm_bn_small_fwd / _bwd): act 0 nothing,
    1 relu6, 2 the context gate  mul * sigmoid(bn(x))."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, act, mul):
        lib = _capi.load()
        x = _f32(x, "x").contiguous()
        M, C = x.shape
        mul = _f32(mul, "mul").contiguous() if mul is not None else None
        y, mean, rstd = torch.empty_like(x), _empty((C,), x), _empty((C,), x)
        lib.check(lib._lpm_bn_small_fwd(ptr(x), M, C, ptr(gamma), ptr(beta), BN_EPS, BN_DECAY, act, ptr(mul), ptr(y), ptr(mean), ptr(rstd),
                                        ptr(moving_mean), ptr(moving_var), stream_ptr()), "lpm_bn_small_fwd")
        ctx.act = act
        ctx.save_for_backward(x, gamma, beta, mean, rstd, mul)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _capi.load()
        x, gamma, beta, mean, rstd, mul = ctx.saved_tensors
        M, C = x.shape
        dy = _f32(dy, "dy").contiguous()
        dx, dgamma, dbeta = torch.empty_like(x), _empty((C,), x), _empty((C,), x)
        dmul = torch.empty_like(x) if ctx.act == 2 else None
        lib.check(lib._lpm_bn_small_bwd(ptr(dy), ptr(x), M, C, ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ctx.act, ptr(mul), ptr(dx),
                                        ptr(dgamma), ptr(dbeta), ptr(dmul), stream_ptr()), "lpm_bn_small_bwd")
        return dx, dgamma, dbeta, None, None, None, dmul


BN_SMALL = True              # the clip-level batch norms (hidden1_bn + relu6, gating_bn + gate) as one launch each way (A/B switch)


def bn_small_ok(x, mul=None):
    return (BN_SMALL and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and 1 < x.shape[0] <= 256
            and (mul is None or (mul.shape == x.shape and mul.dtype == torch.float32)))


def bn_small(x, gamma, beta, moving_mean, moving_var, act=0, mul=None):
    """act(batch_norm(x)) in training mode for a small [M, C] matrix: act 0 / 1 (relu6) / 2 (mul * sigmoid(.)); moving statistics updated
    in place (decay BN_DECAY, unbiased variance)."""
    return _BNSmall.apply(x, gamma, beta, moving_mean, moving_var, int(act), mul)


def folded_eval_affine(gamma, beta, moving_mean, moving_var, eps=BN_EPS):
    """Inference-mode batch norm is a constant affine (tiny [C] tensors: plain torch)."""
    scale = gamma * torch.rsqrt(moving_var + eps)
    return scale, beta - moving_mean * scale


def l2_normalize_rows(x):
    """tf.nn.l2_normalize over the last axis of the input batch (train.py:262-264): one read, one write.  The frames are
    data -- no gradient."""
    lib = _capi.load()
    x = _f32(x, "model_input_raw").contiguous()
    F = x.shape[-1]
    y = torch.empty_like(x)
    lib.check(lib._lpm_l2_normalize_rows(ptr(x), x.numel() // F, F, ptr(y), stream_ptr()), "lpm_l2_normalize_rows")
    return y


def dequantize_l2_normalize(q, num_frames, max_quantized_value=2.0, min_quantized_value=-2.0):
    """Quantised reader output (uint8 [B, max_frames, F]) -> dequantised (utils.Dequantize), zero-padded past num_frames,
    L2-normalised fp32 frames: the reader's tail (readers.py:176-193) and train.py:262-264 as one pass."""
    lib = _capi.load()
    if q.dtype != torch.uint8 or q.dim() != 3:
        raise LpmError("dequantize_l2_normalize: expected a uint8 [batch, max_frames, feature] tensor")
    q = q.contiguous()
    B, MF, F = q.shape
    nf = num_frames.to(device=q.device, dtype=torch.int32).reshape(-1).contiguous()
    if nf.numel() != B:
        raise LpmError("dequantize_l2_normalize: num_frames must have one entry per clip")
    y = torch.empty((B, MF, F), dtype=torch.float32, device=q.device)
    lib.check(lib._lpm_dequantize_l2_normalize(ptr(q), ptr(nf), B, MF, F, float(max_quantized_value), float(min_quantized_value),
                                               ptr(y), stream_ptr()), "lpm_dequantize_l2_normalize")
    return y


# ----------------------------------------------------------------------------------------------
# a2 + a3: SampleUniformFrames + input_bn
# ----------------------------------------------------------------------------------------------
class _FrameSampleBN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, num_frames, gamma, beta, moving_mean, moving_var, S, is_training, use_bn, storage="f32", materialize=True):
        lib = _capi.load()
        raw = _f32(raw, "model_input").contiguous()
        if raw.dim() != 3:
            raise LpmError("model_input must be [batch, max_frames, feature]")
        B, MF, F = raw.shape
        nf = num_frames.to(device=raw.device, dtype=torch.int32).contiguous()
        y = _empty((B * S, F), raw)
        mean = var = None
        if use_bn:
            if is_training:
                nblk = lib._lpm_frame_stats_nblk(B, S)
                partial = _empty((nblk, 2, F), raw)
                lib.check(lib._lpm_frame_stats(ptr(raw), ptr(nf), B, MF, F, S, ptr(partial), stream_ptr()), "lpm_frame_stats")
                mean, var, scale, shift = bn_fold(partial, nblk, F, B * S, gamma, beta, moving_mean, moving_var)
            else:
                scale, shift = folded_eval_affine(gamma, beta, moving_mean, moving_var)
                scale, shift = scale.contiguous(), shift.contiguous()
        else:
            scale = shift = None
        tiles = VLAD_PRECISION == "bf16x3" and F in (1024, 1152)
        if storage == "bf16":
            # bf16 storage: the frames leave as plain bf16 operand tiles for K1 (row tiles) and K2 (frame tiles) in one pass; the
            # fp32 matrix is written only when somebody will look at it (materialize)
            if F not in (1024, 1152):
                raise LpmError("frame_sample_bn: bf16 storage needs a 1024- or 1152-wide input")
            Dv, Da = 1024, F - 1024
            nb = lambda d: torch.empty(lib._lpm_frame_tiles_bf16_bytes(B, S, d) // 4, dtype=torch.int32, device=raw.device)
            xtv, xrv = nb(Dv), nb(Dv)
            xta, xra = (nb(Da), nb(Da)) if Da else (None, None)
            lib.check(lib._lpm_frame_apply_tiles_bf16(ptr(raw), ptr(nf), B, MF, F, S, ptr(scale), ptr(shift), ptr(y) if materialize else None,
                                                      ptr(xtv), ptr(xrv), Dv, ptr(xta), ptr(xra), Da, stream_ptr()),
                      "lpm_frame_apply_tiles_bf16")
            _XT_CACHE.clear()
            _XT_CACHE.update(base=weakref.ref(y), F=F, Dv=Dv, S=S, B=B, video=xtv, audio=xta, video_rows=xrv, audio_rows=xra,
                             storage="bf16", unmaterialised=None if materialize else y.untyped_storage().data_ptr())
        elif tiles:
            Dv, Da = 1024, F - 1024
            xtv = torch.empty(lib._lpm_xt_bytes(B, S, Dv) // 4, dtype=torch.int32, device=raw.device)
            xta = torch.empty(lib._lpm_xt_bytes(B, S, Da) // 4, dtype=torch.int32, device=raw.device) if Da else None
            xrv = xra = None
            if FRAME_ROW_TILES:
                # K1's row-tile operand leaves with the same pass (no lpm_split_rows_tiles over the fp32 matrix afterwards)
                xrv = torch.empty(lib._lpm_row_tiles_bytes(B, S, Dv) // 4, dtype=torch.int32, device=raw.device)
                xra = torch.empty(lib._lpm_row_tiles_bytes(B, S, Da) // 4, dtype=torch.int32, device=raw.device) if Da else None
                lib.check(lib._lpm_frame_apply_tiles2(ptr(raw), ptr(nf), B, MF, F, S, ptr(scale), ptr(shift), ptr(y), ptr(xtv), ptr(xrv),
                                                      Dv, ptr(xta), ptr(xra), Da, stream_ptr()), "lpm_frame_apply_tiles2")
            else:
                lib.check(lib._lpm_frame_apply_tiles(ptr(raw), ptr(nf), B, MF, F, S, ptr(scale), ptr(shift), ptr(y), ptr(xtv), Dv,
                                                     ptr(xta), Da, stream_ptr()), "lpm_frame_apply_tiles")
            _XT_CACHE.clear()
            _XT_CACHE.update(base=weakref.ref(y), F=F, Dv=Dv, S=S, B=B, video=xtv, audio=xta, video_rows=xrv, audio_rows=xra)
        else:
            lib.check(lib._lpm_frame_apply(ptr(raw), ptr(nf), B, MF, F, S, ptr(scale), ptr(shift), ptr(y), stream_ptr()),
                      "lpm_frame_apply")
        ctx.use_bn, ctx.is_training, ctx.S = use_bn, is_training, S
        if use_bn:
            if is_training:
                ctx.save_for_backward(raw, nf, mean, var)
            else:
                ctx.save_for_backward(raw, nf, moving_mean.detach().clone(), moving_var.detach().clone())
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.use_bn:
            return (None,) * 11
        lib = _capi.load()
        raw, nf, mean, var = ctx.saved_tensors
        B, MF, F = raw.shape
        dy = _rows(dy.contiguous(), "dy")
        dgamma, dbeta = _empty((F,), raw), _empty((F,), raw)
        wsb = lib._lpm_frame_stats_workspace_bytes(B, ctx.S, F)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=raw.device)
        lib.check(lib._lpm_frame_bn_bwd(ptr(dy), dy.stride(0), ptr(raw), ptr(nf), B, MF, F, ctx.S, ptr(mean), ptr(var),
                                        BN_EPS, ptr(dgamma), ptr(dbeta), ptr(ws), wsb, stream_ptr()), "lpm_frame_bn_bwd")
        return None, None, dgamma, dbeta, None, None, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# second HIP stream for the audio branch
# ----------------------------------------------------------------------------------------------
_SIDE_STREAMS = {}


class side_stream:
    """``with side_stream(*inputs) as s:`` runs the enclosed launches on a secondary HIP stream that first waits for the
    work queued so far on the current stream; ``s.join(*outputs)`` makes the current stream wait for it.  Autograd replays
    every node on the stream its forward ran on, so the backward of the enclosed ops overlaps the same way.  Tensors
    crossing streams are recorded with the allocator."""

    def __init__(self, *inputs):
        self.inputs = [t for t in inputs if t is not None]
        self.forked = False

    def __enter__(self):
        if not self.forked:                                 # re-entering continues on the side stream without a new fork
            self.forked = True
            self.main = torch.cuda.current_stream()
            dev = self.main.device
            if dev not in _SIDE_STREAMS:
                _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
            self.side = _SIDE_STREAMS[dev]
            self.side.wait_stream(self.main)
            for t in self.inputs:
                t.record_stream(self.side)
            for key in ("video", "audio", "video_rows", "audio_rows"):     # tile copies written by frame_sample_bn on the main stream
                t = _XT_CACHE.get(key)
                if t is not None:
                    t.record_stream(self.side)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)

    def join(self, *outputs):
        self.main.wait_stream(self.side)
        for t in outputs:
            t.record_stream(self.main)


class _SplitColumns(torch.autograd.Function):
    """x [M, F] -> (x[:, :c], x[:, c:]) as views.  The two streams' pooling ops write their input gradients as column views
    of ONE shared [M, F] buffer (the ``_lpm_dx_slot`` they find on their input), and the backward hands that buffer on as
    the gradient of x: no zero-fill, scatter-copy and add per stream (what two plain slices cost in autograd)."""

    @staticmethod
    def forward(ctx, x, c):
        a, b = x[:, :c], x[:, c:]
        slot = {"buf": None, "shape": tuple(x.shape), "device": x.device}
        ctx.slot, ctx.c = slot, c
        a._lpm_dx_slot, b._lpm_dx_slot = (slot, 0), (slot, c)
        return a, b

    @staticmethod
    def backward(ctx, da, db):
        slot, c = ctx.slot, ctx.c
        buf, slot["buf"] = slot["buf"], None
        M, F = slot["shape"]
        if (buf is not None and da is not None and db is not None and da.stride() == (F, 1) and db.stride() == (F, 1)
                and da.data_ptr() == buf.data_ptr() and db.data_ptr() == buf.data_ptr() + c * buf.element_size()):
            return buf, None
        if da is None:
            da = torch.zeros((M, c), dtype=torch.float32, device=slot["device"])
        if db is None:
            db = torch.zeros((M, F - c), dtype=torch.float32, device=slot["device"])
        return torch.cat([da, db], dim=1), None


class _SplitVector(torch.autograd.Function):
    """v [F] -> (v[:c], v[c:]) as views whose gradients come back as ONE concatenation (two plain slices cost autograd a zero-fill, a
    scatter copy and an add each: ten 5-us launches per step for input_bn's gamma and beta)."""

    @staticmethod
    def forward(ctx, v, c):
        ctx.c, ctx.n, ctx.like = c, v.shape[0], v
        return v[:c], v[c:]

    @staticmethod
    def backward(ctx, da, db):
        c, n, like = ctx.c, ctx.n, ctx.like
        if da is None:
            da = torch.zeros(c, dtype=like.dtype, device=like.device)
        if db is None:
            db = torch.zeros(n - c, dtype=like.dtype, device=like.device)
        return torch.cat([da.reshape(-1), db.reshape(-1)]), None


def split_vector(v, c):
    """(v[:c], v[c:]) of a 1-D tensor with a concatenated gradient (see _SplitVector)."""
    return _SplitVector.apply(v, int(c))


def split_columns(x, c):
    """(x[:, :c], x[:, c:]) with a shared gradient buffer (see _SplitColumns)."""
    return _SplitColumns.apply(x, int(c))


def _dx_slot_view(slot_ref, D):
    """The [M, D] column view of the shared input-gradient buffer this op should write dx into (None: allocate its own)."""
    if slot_ref is None:
        return None
    slot, c0 = slot_ref
    if slot["buf"] is None:
        slot["buf"] = torch.empty(slot["shape"], dtype=torch.float32, device=slot["device"])
    return slot["buf"][:, c0:c0 + D]


def frame_sample_bn(raw, num_frames, S, gamma=None, beta=None, moving_mean=None, moving_var=None, is_training=True, storage="f32",
                    materialize=True):
    """[B, max_frames, F] -> [B*S, F]: uniform frame sampling (model_utils.py:101-122) fused with
    input_bn (frame_level_models.py:2265-2271).  The frames are data: no gradient flows to ``raw``.
    storage="bf16" (BASELINE cfg-5): the result is written as plain bf16 operand tiles for ops.netvlad(storage="bf16"); the fp32
    matrix that is returned is filled in only with ``materialize`` (otherwise it is a handle nobody may read)."""
    use_bn = gamma is not None
    return _FrameSampleBN.apply(raw, num_frames, gamma, beta, moving_mean, moving_var, int(S), bool(is_training), use_bn, storage,
                                bool(materialize))


class _FrameSampleBNSplit(torch.autograd.Function):
    """ops.frame_sample_bn with the rgb / audio column blocks as TWO contiguous matrices (round 6: NetVladV2 -- each stream's frame
    encoder and aggregation take whole rows, so the column slices of one [B S, F] matrix cost a contiguous copy per stream, a second
    frame-tile pass per stream because the copies were not the tensors the tiles were cached for, and a concatenation of the two gradients).
    The same values bit for bit; the frame tiles written in the same pass are found for the two outputs (ops._cached_tiles)."""

    @staticmethod
    def forward(ctx, raw, num_frames, gamma, beta, moving_mean, moving_var, S, is_training, Dv):
        lib = _capi.load()
        raw = _f32(raw, "model_input").contiguous()
        B, MF, F = raw.shape
        Da = F - Dv
        nf = num_frames.to(device=raw.device, dtype=torch.int32).contiguous()
        if is_training:
            nblk = lib._lpm_frame_stats_nblk(B, S)
            partial = _empty((nblk, 2, F), raw)
            lib.check(lib._lpm_frame_stats(ptr(raw), ptr(nf), B, MF, F, S, ptr(partial), stream_ptr()), "lpm_frame_stats")
            mean, var, scale, shift = bn_fold(partial, nblk, F, B * S, gamma, beta, moving_mean, moving_var)
        else:
            scale, shift = folded_eval_affine(gamma, beta, moving_mean, moving_var)
            scale, shift = scale.contiguous(), shift.contiguous()
            mean, var = moving_mean.detach().clone(), moving_var.detach().clone()
        yv, ya = _empty((B * S, Dv), raw), _empty((B * S, Da), raw)
        xtv = torch.empty(lib._lpm_xt_bytes(B, S, Dv) // 4, dtype=torch.int32, device=raw.device)
        xta = torch.empty(lib._lpm_xt_bytes(B, S, Da) // 4, dtype=torch.int32, device=raw.device)
        lib.check(lib._lpm_frame_apply_tiles_split(ptr(raw), ptr(nf), B, MF, F, S, ptr(scale), ptr(shift), ptr(yv), ptr(ya), ptr(xtv), Dv,
                                                   ptr(xta), Da, stream_ptr()), "lpm_frame_apply_tiles_split")
        _XT_CACHE.clear()
        _XT_CACHE.update(split=True, F=F, Dv=Dv, S=S, B=B, video=xtv, audio=xta, video_t=yv, audio_t=ya,
                         video_ver=yv._version, audio_ver=ya._version)
        ctx.S, ctx.Dv = S, Dv
        ctx.save_for_backward(raw, nf, mean, var)
        return yv, ya

    @staticmethod
    def backward(ctx, dv, da):
        lib = _capi.load()
        raw, nf, mean, var = ctx.saved_tensors
        B, MF, F = raw.shape
        Dv = ctx.Dv
        M = B * ctx.S
        dv = _rows(dv.contiguous(), "dy") if dv is not None else torch.zeros((M, Dv), dtype=torch.float32, device=raw.device)
        da = _rows(da.contiguous(), "dy") if da is not None else torch.zeros((M, F - Dv), dtype=torch.float32, device=raw.device)
        dgamma, dbeta = _empty((F,), raw), _empty((F,), raw)
        wsb = lib._lpm_frame_stats_workspace_bytes(B, ctx.S, F)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=raw.device)
        lib.check(lib._lpm_frame_bn_bwd_split(ptr(dv), dv.stride(0), ptr(da), da.stride(0), Dv, ptr(raw), ptr(nf), B, MF, F, ctx.S, ptr(mean),
                                              ptr(var), BN_EPS, ptr(dgamma), ptr(dbeta), ptr(ws), wsb, stream_ptr()), "lpm_frame_bn_bwd_split")
        return None, None, dgamma, dbeta, None, None, None, None, None


FRAME_SPLIT = os.environ.get("LPM_FRAME_SPLIT", "1") != "0"      # "0": NetVladV2 slices one [B S, F] matrix (A/B)


def frame_sample_bn_split_ok(raw, Dv):
    return bool(FRAME_SPLIT and raw.is_cuda and raw.dim() == 3 and raw.dtype == torch.float32 and VLAD_PRECISION == "bf16x3"
                and 0 < Dv < raw.shape[2] and Dv % 32 == 0 and (raw.shape[2] - Dv) % 32 == 0)


def frame_sample_bn_split(raw, num_frames, S, gamma, beta, moving_mean, moving_var, is_training, Dv):
    """-> (rgb [B*S, Dv], audio [B*S, F - Dv]): uniform frame sampling + input_bn (model_utils.py:101-122, frame_level_models.py:2265-2271)
    with the two streams' blocks as separate contiguous matrices; gamma / beta receive ONE gradient each."""
    return _FrameSampleBNSplit.apply(raw, num_frames, gamma, beta, moving_mean, moving_var, int(S), bool(is_training), int(Dv))


# ----------------------------------------------------------------------------------------------
# K1 + K2 (+K3): NetVLAD pooling
# ----------------------------------------------------------------------------------------------
_K1_CHECKED = set()
K1_SELFCHECK = os.environ.get("LPM_K1_SELFCHECK", "1") != "0"


def _k1_selfcheck(lib, planes, dev):
    """Once per process and operand form: K1's hand-scheduled forward kernels (csrc/assign_flat.hip: registers handed to asynchronous loads
    behind hand-counted waits) against the tile-GEMM form of the SAME entry point on a small problem -- logits to 2e-5 of their
    maximum (fp32; two bf16 ulps for bf16 storage), statistics to 1e-5.  A form that disagrees is switched off for the
    process (lpm_k1_forms_disable) with a warning: a toolchain change that breaks the inline-assembly register hand-over turns into a
    slower step, not into wrong logits (tests/test_build_flags.py is the build-time half of this)."""
    if not K1_SELFCHECK or planes in _K1_CHECKED:
        return
    _K1_CHECKED.add(planes)
    import warnings
    B, T, D = 3, 120, 1024                               # 360 rows: three clips straddled by the row groups, a partial last group
    g = torch.Generator(device="cpu").manual_seed(20251)
    x = torch.randn(B * T, D, generator=g).to(dev)
    st = stream_ptr()
    prev = lib._lpm_k1_forms_disable(0)                   # (an earlier check's verdict stays in force: restored below)
    try:
        for K, bit in ((256, 1), (512, 2)) if planes == 1 else ((256, 1),):
            W = (torch.randn(D, K, generator=g) / 32).to(dev)
            nblk = lib._lpm_assign_gemm_tiles_nblk(B, T)
            if planes == 2:
                xr = torch.empty(lib._lpm_row_tiles_bytes(B, T, D) // 4, dtype=torch.int32, device=dev)
                wt = torch.empty(lib._lpm_weight_tiles_bytes(D, K) // 4, dtype=torch.int32, device=dev)
                lib.check(lib._lpm_split_rows_tiles(ptr(x), D, B, T, D, ptr(xr), st), "lpm_split_rows_tiles")
                lib.check(lib._lpm_split_weight_tiles(ptr(W), D, K, 0, ptr(wt), st), "lpm_split_weight_tiles")
                run = lambda lg, pt: lib.check(lib._lpm_assign_gemm_tiles_fwd(ptr(xr), ptr(wt), B, T, D, K, ptr(lg), ptr(pt), st), "lpm_assign_gemm_tiles_fwd")
                dt = torch.float32
            else:
                # plain bf16 row tiles come from the frame pass (lpm_frame_apply_tiles_bf16: identity sampling, unit affine), into buffers of
                # this check's own -- the step's tile cache (_XT_CACHE) is not touched
                raw = torch.cat([x.view(B, T, D), torch.zeros(B, T, 128, device=dev)], 2).contiguous()
                nfr = torch.full((B,), T, dtype=torch.int32, device=dev)
                one, zero = torch.ones(D + 128, device=dev), torch.zeros(D + 128, device=dev)
                nb = lambda d: torch.empty(lib._lpm_frame_tiles_bf16_bytes(B, T, d) // 4, dtype=torch.int32, device=dev)
                xtv, xr, xta, xra = nb(D), nb(D), nb(128), nb(128)
                lib.check(lib._lpm_frame_apply_tiles_bf16(ptr(raw), ptr(nfr), B, T, D + 128, T, ptr(one), ptr(zero), None, ptr(xtv), ptr(xr), D,
                                                          ptr(xta), ptr(xra), 128, st), "lpm_frame_apply_tiles_bf16")
                wt = torch.empty(lib._lpm_weight_tiles_bytes(D, K) // 8, dtype=torch.int32, device=dev)
                lib.check(lib._lpm_split_weight_tiles_bf16(ptr(W), D, K, 0, ptr(wt), st), "lpm_split_weight_tiles_bf16")
                run = lambda lg, pt: lib.check(lib._lpm_assign_gemm_tiles_fwd_bf16(ptr(xr), ptr(wt), B, T, D, K, ptr(lg), ptr(pt), st),
                                               "lpm_assign_gemm_tiles_fwd_bf16")
                dt = torch.bfloat16
            res = []
            for mask in (prev & 3, 3):                      # the forms in force; every hand-scheduled form off
                lib._lpm_k1_forms_disable(mask)
                lg = torch.zeros((B * T, K), dtype=dt, device=dev)
                pt = torch.zeros((nblk, 2, K), dtype=torch.float32, device=dev)
                run(lg, pt)
                res.append((lg, pt.sum(0)))
            (la, sa), (lb, sb) = res
            # (the forms order a step's three split-bf16 products differently: fp32 rounding apart, 1e-6; a bf16 result may land one ulp
            # apart.  A broken register hand-over gives garbage, orders of magnitude beyond either)
            tol = 2e-5 if planes == 2 else 2.0 ** -7
            ok = (bool(((la.float() - lb.float()).abs() <= tol * lb.float().abs().max()).all())
                  and bool(((sa - sb).abs() <= 1e-5 * sb.abs().max()).all()))
            if not ok:
                prev |= bit
                warnings.warn(f"learnablepoolingmethods_amd: K1's hand-scheduled forward kernel ({'split-bf16' if planes == 2 else 'plain bf16'}, "
                              f"K={K}) disagrees with the tile-GEMM form on the self-check problem; it is switched OFF for this process "
                              "(slower, correct).  Rebuild with the pinned toolchain and run tests/test_build_flags.py.")
    finally:
        lib._lpm_k1_forms_disable(prev & 3)


def _cached_tiles(x, B, T, D, rows=False, storage="f32"):
    """The tile copy written by frame_sample_bn, if ``x`` is the rgb / audio column slice of its latest output.  rows: the row
    tiles (K1's operand; bf16 storage only) instead of the frame tiles."""
    c = _XT_CACHE
    if not c or c["B"] != B or c["S"] != T or c.get("storage", "f32") != storage:
        return None
    if c.get("split"):                      # frame_sample_bn_split: two contiguous matrices, each with its own tiles
        if rows:
            return None
        for which, Dw in (("video", c["Dv"]), ("audio", c["F"] - c["Dv"])):
            t = c[which + "_t"]             # (held until the next frame_sample_bn: its memory cannot have been handed to anyone else)
            if (D == Dw and x.data_ptr() == t.data_ptr() and tuple(x.shape) == tuple(t.shape) and x.is_contiguous()
                    and t._version == c[which + "_ver"]):
                return c[which]
        return None
    base = c["base"]()
    if base is None or x._base is not base or x.stride(0) != c["F"]:
        return None
    suffix = "_rows" if rows else ""
    if x.storage_offset() == 0 and D == c["Dv"]:
        return c.get("video" + suffix)
    if x.storage_offset() == c["Dv"] and D == c["F"] - c["Dv"] and c["audio"] is not None:
        return c.get("audio" + suffix)
    return None


def _nrm_raw_ok(lib, T, D, K):
    """The forward may leave ``nrm`` un-normalised (no in-place write in the finalize pass): the LDS-shared K2 form produces
    it and K3's tile form, which rebuilds the normalised value where it reads it, will consume it."""
    return (VLAD_PRECISION == "bf16x3" and VLAD_TILES3 and bool(lib._lpm_vlad_tiles3_supported(D, K)) and _bwd_tiles_ok(lib, T, D, K))


def _aggregate_fwd(lib, assign, scale, shift, x, centres, B, T, D, K, flags, kmajor, nrm_raw=False, save_u=True, lazy=False):
    """-> out, nrm, asum, colsq, csq, gsq, xt (the split-bf16 frame tiles of x, or None on the fp32 path).
    nrm_raw (only with _nrm_raw_ok): ``nrm`` comes back as the un-normalised sums U."""
    xt = None
    nrm = None if lazy else _empty((B, D, K), x)
    asum, colsq, csq = (_empty((B, K), x) for _ in range(3))
    if VLAD_PRECISION == "bf16x3":
        st = stream_ptr()
        xt = _cached_tiles(x, B, T, D)
        if xt is None:
            xt = torch.empty(lib._lpm_xt_bytes(B, T, D) // 4, dtype=torch.int32, device=x.device)
            with _timed("split_frames", (B, T, D)):
                lib.check(lib._lpm_split_frames(ptr(x), x.stride(0), B, T, D, ptr(xt), st), "lpm_split_frames")
        if (lazy and VLAD_SOFTMAX_FUSED and (flags & LPM_VLAD_SOFTMAX) and kmajor and VLAD_TILES3 and assign.dtype == torch.float32
                and lib._lpm_vlad_smx_supported(T, D, K)):
            # softmax -> residual aggregation in ONE kernel (frame_level_models.py:2798-2817): the assignment never exists in memory,
            # not even as tiles; a small launch before it leaves the per-frame row maximum and 1 / row sum
            raw = _empty((B, K, D), x)
            P = D // 128
            part = _empty((B, P, K), x)
            stats = torch.empty(lib._lpm_vlad_smx_stats_bytes(B, T) // 4, dtype=torch.float32, device=x.device)
            with _timed("vlad_aggregate_fwd", (B, T, D, K)):
                lib.check(lib._lpm_vlad_aggregate_raw_kmajor_smx_fwd(ptr(assign), ptr(scale), ptr(shift), ptr(xt), ptr(centres), B, T, D, K,
                                                                     flags & (LPM_VLAD_RESIDUAL | LPM_VLAD_SOFTMAX), ptr(raw), ptr(asum),
                                                                     ptr(part), ptr(stats), st), "lpm_vlad_aggregate_raw_kmajor_smx_fwd")
            rs = _empty((B, K), x)
            gsq = _empty((B,), x)
            with _timed("vlad_finalize", (B, D, K)):
                lib.check(lib._lpm_vlad_row_scales(ptr(part), P, B, K, ptr(rs), ptr(colsq), ptr(csq), ptr(gsq), st), "lpm_vlad_row_scales")
            raw._lpm_row_scale = rs
            return raw, raw, asum, colsq, csq, gsq, xt
        at = torch.empty(lib._lpm_at_bytes(B, T, K) // 4, dtype=torch.int32, device=x.device)
        with _timed("assign_tiles", (B, T, K)):
            lib.check(lib._lpm_assign_tiles(ptr(assign), ptr(scale), ptr(shift), B, T, K, flags, ptr(at), st), "lpm_assign_tiles")
        if lazy:
            # the un-normalised sums k-major, written once, + one scale per (clip, cluster): the consumers normalise (see netvlad())
            if not (kmajor and VLAD_TILES3 and lib._lpm_vlad_tiles3_supported(D, K) and K <= 1024):
                raise LpmError("internal: lazy descriptor without the LDS-shared aggregation form")
            raw = _empty((B, K, D), x)
            P = D // 128
            if VLAD_KMAJOR_SCALED:
                # K2 and the row scales in one launch (vlad_kmajor.hip): wide workgroups at K = 256, the last workgroup of a clip
                # turns the clip's partial norms into its row scales
                rs = _empty((B, K), x)
                gsq = _empty((B,), x)
                wsb = lib._lpm_vlad_kmajor_workspace_bytes(B, D, K)
                ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device)
                with _timed("vlad_aggregate_fwd", (B, T, D, K)):
                    lib.check(lib._lpm_vlad_aggregate_kmajor_scaled_fwd(ptr(at), ptr(xt), ptr(centres), B, T, D, K, flags & LPM_VLAD_RESIDUAL,
                                                                        ptr(raw), ptr(rs), ptr(asum), ptr(colsq), ptr(csq), ptr(gsq),
                                                                        ptr(ws), wsb, st), "lpm_vlad_aggregate_kmajor_scaled_fwd")
                raw._lpm_row_scale = rs
                return raw, raw, asum, colsq, csq, gsq, xt
            Pc = lib._lpm_vlad_clip_slabs(D, K) if VLAD_CLIP else 0
            if Pc:
                P = Pc
            part = _empty((B, P, K), x)
            with _timed("vlad_aggregate_fwd", (B, T, D, K)):
                if Pc:
                    lib.check(lib._lpm_vlad_aggregate_clip_kmajor_fwd(ptr(at), ptr(xt), ptr(centres), B, T, D, K, flags & LPM_VLAD_RESIDUAL,
                                                                      ptr(raw), ptr(asum), ptr(part), st), "lpm_vlad_aggregate_clip_kmajor_fwd")
                else:
                    lib.check(lib._lpm_vlad_aggregate_raw_kmajor_fwd(ptr(at), ptr(xt), ptr(centres), B, T, D, K, flags & LPM_VLAD_RESIDUAL,
                                                                     ptr(raw), ptr(asum), ptr(part), st), "lpm_vlad_aggregate_raw_kmajor_fwd")
            rs = _empty((B, K), x)
            gsq = _empty((B,), x)
            with _timed("vlad_finalize", (B, D, K)):
                lib.check(lib._lpm_vlad_row_scales(ptr(part), P, B, K, ptr(rs), ptr(colsq), ptr(csq), ptr(gsq), st), "lpm_vlad_row_scales")
            raw._lpm_row_scale = rs
            return raw, raw, asum, colsq, csq, gsq, xt
        if VLAD_TILES3 and VLAD_FUSED and nrm_raw and lib._lpm_vlad_fused_supported(D, K):
            # LDS-shared form with both normalisations fused in: the descriptor is written once, normalised; the un-normalised
            # sums go to HBM only when a backward will read them (save_u: some input of the op requires a gradient)
            out = _empty((B, K, D) if kmajor else (B, D * K), x)
            gsq = _empty((B,), x)
            wsb = lib._lpm_vlad_fused_workspace_bytes(B, D, K)
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device)
            ffl = (flags & LPM_VLAD_RESIDUAL) | (LPM_VLAD_OUT_KMAJOR if kmajor else 0) | (LPM_VLAD_NRM_RAW if save_u else 0)
            if VLAD_FUSED_DEBUG_FALLBACK:
                ffl |= _capi.LPM_VLAD_DEBUG_FALLBACK
            with _timed("vlad_aggregate_fwd", (B, T, D, K)):
                lib.check(lib._lpm_vlad_aggregate_fused_fwd(ptr(at), ptr(xt), ptr(centres), B, T, D, K, ffl, ptr(nrm), ptr(out), ptr(asum),
                                                            ptr(colsq), ptr(csq), ptr(gsq), ptr(ws), wsb, st),
                          "lpm_vlad_aggregate_fused_fwd")
            return out, nrm, asum, colsq, csq, gsq, xt
        if VLAD_TILES3 and lib._lpm_vlad_tiles3_supported(D, K):
            # LDS-shared form: un-normalised sums + partial square norms; finalize2 applies both normalisations
            P = D // 128
            part = _empty((B, P, K), x)
            with _timed("vlad_aggregate_fwd", (B, T, D, K)):
                lib.check(lib._lpm_vlad_aggregate_tiles3_fwd(ptr(at), ptr(xt), ptr(centres), B, T, D, K, flags, ptr(nrm),
                                                             ptr(asum), ptr(part), st), "lpm_vlad_aggregate_tiles3_fwd")
            out = _empty((B, K, D) if kmajor else (B, D * K), x)
            gsq = _empty((B,), x)
            ffl = (LPM_VLAD_OUT_KMAJOR if kmajor else 0) | (LPM_VLAD_NRM_RAW if nrm_raw else 0)
            with _timed("vlad_finalize", (B, D, K)):
                lib.check(lib._lpm_vlad_finalize2_fwd(ptr(nrm), ptr(part), P, B, D, K, ffl, ptr(out), ptr(colsq), ptr(csq), ptr(gsq), st),
                          "lpm_vlad_finalize2_fwd")
            return out, nrm, asum, colsq, csq, gsq, xt
        if nrm_raw:
            raise LpmError("internal: nrm_raw without the LDS-shared aggregation form")
        with _timed("vlad_aggregate_fwd", (B, T, D, K)):
            lib.check(lib._lpm_vlad_aggregate_tiles_fwd(ptr(at), ptr(xt), ptr(centres), B, T, D, K, flags, ptr(nrm), ptr(asum),
                                                        ptr(colsq), ptr(csq), st), "lpm_vlad_aggregate_tiles_fwd")
    elif VLAD_PRECISION == "f32":
        with _timed("vlad_aggregate_fwd", (B, T, D, K)):
            lib.check(lib._lpm_vlad_aggregate_fwd(ptr(assign), ptr(scale), ptr(shift), ptr(x), x.stride(0), ptr(centres), B, T,
                                                  D, K, flags, ptr(nrm), ptr(asum), ptr(colsq), ptr(csq), stream_ptr()),
                      "lpm_vlad_aggregate_fwd")
    else:
        raise LpmError(f"unknown LPM_VLAD_PRECISION {VLAD_PRECISION!r} (bf16x3 | f32)")
    out = _empty((B, K, D) if kmajor else (B, D * K), x)
    gsq = _empty((B,), x)
    lib.check(lib._lpm_vlad_finalize_fwd(ptr(nrm), ptr(csq), B, D, K, LPM_VLAD_OUT_KMAJOR if kmajor else 0, ptr(out),
                                         ptr(gsq), stream_ptr()), "lpm_vlad_finalize_fwd")
    return out, nrm, asum, colsq, csq, gsq, xt


def _aggregate_bwd(lib, dout, nrm, asum, colsq, csq, gsq, assign, scale, shift, x, centres, B, T, D, K, flags, kmajor):
    dassign = _empty((B * T, K), x)
    dx = _empty((B * T, D), x)
    dcentres = _empty((D, K), x) if centres is not None else None
    wsb = lib._lpm_vlad_bwd_workspace_bytes(B, D, K)
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device)
    fl = flags | (LPM_VLAD_OUT_KMAJOR if kmajor else 0)
    lib.check(lib._lpm_vlad_aggregate_bwd(ptr(dout), ptr(nrm), ptr(asum), ptr(colsq), ptr(csq), ptr(gsq), ptr(assign),
                                          ptr(scale), ptr(shift), ptr(x), x.stride(0), ptr(centres), B, T, D, K, fl,
                                          ptr(dassign), ptr(dx), D, 0, ptr(dcentres), ptr(ws), wsb, stream_ptr()),
              "lpm_vlad_aggregate_bwd")
    return dassign, dx, dcentres


def _tile_buffer(nbytes, like):
    return torch.empty(nbytes // 4, dtype=torch.int32, device=like.device)


def _bwd_tiles_ok(lib, T, D, K):
    """K3 on the bf16 pipe (tile form) is available for this shape and arithmetic."""
    return VLAD_PRECISION == "bf16x3" and bool(lib._lpm_assign_gemm_tiles_supported(T, D, K))


def _aggregate_bwd_tiles(lib, dout, nrm, asum, colsq, csq, gsq, assign, scale, shift, x, xr, centres, B, T, D, K, flags, kmajor,
                         no_dx=False, nrm_raw=False, raw_kmajor=False):
    """First half of K3's tile form: -> dassign, dcentres, (workspace, bytes) for _aggregate_bwd_tiles_dx, g0.
    no_dx: the frames need no gradient -- the dx operands are not produced; g0 [B, D] = sum_k dU U and dcentres = -sum_b asum dU
    (also without a residual term) come back instead (see _NetVLAD.backward)."""
    st = stream_ptr()
    if xr is None:
        xr = _tile_buffer(lib._lpm_row_tiles_bytes(B, T, D), x)
        lib.check(lib._lpm_split_rows_tiles(ptr(x), x.stride(0), B, T, D, ptr(xr), st), "lpm_split_rows_tiles")
    dassign = _empty((B * T, K), x)
    dcentres = _empty((D, K), x) if (centres is not None or no_dx) else None
    g0 = _empty((B, D), x) if no_dx else None
    wsb = lib._lpm_vlad_bwd_tiles_workspace_bytes(B, T, D, K)
    ws = _tile_buffer(wsb, x)
    fl = flags | (LPM_VLAD_OUT_KMAJOR if kmajor else 0) | (LPM_VLAD_NRM_RAW if nrm_raw else 0)
    if raw_kmajor:          # dout and the saved sums are both k-major: no transposes (needs the no-input-gradient form)
        fl |= _capi.LPM_VLAD_RAW_KMAJOR
    # a d-major gradient that is a column slice of the concatenated descriptors' gradient (torch.cat's backward hands out views) is read
    # in place, clips dout.stride(0) elements apart; anything else contiguous
    dob = D * K
    if (dout.dim() == 2 and not kmajor and not raw_kmajor and dout.dtype == torch.float32 and dout.stride(1) == 1
            and dout.stride(0) >= D * K and dout.stride(0) % 4 == 0 and dout.data_ptr() % 16 == 0):
        dob = dout.stride(0)
    else:
        dout = dout.contiguous()
    with _timed("vlad_aggregate_bwd", (B, T, D, K)):
        lib.check(lib._lpm_vlad_aggregate_bwd_tiles_ld(ptr(dout), dob, ptr(nrm), ptr(asum), ptr(colsq), ptr(csq), ptr(gsq), ptr(assign),
                                                       ptr(scale), ptr(shift), ptr(xr), ptr(centres), B, T, D, K, fl, ptr(dassign),
                                                       ptr(dcentres), ptr(g0), ptr(ws), wsb, st), "lpm_vlad_aggregate_bwd_tiles")
    return dassign, dcentres, (ws, wsb), g0


def _aggregate_bwd_tiles_dx(lib, wspace, dlr, wtt, like, B, T, D, K, out=None):
    """Second half: dx = sum_k a dU (+ dl . W^T when the assignment GEMM's tiles are given), into ``out`` (a [B*T, D] view
    with unit column stride, e.g. a column slice of a wider gradient buffer) when given."""
    ws, wsb = wspace
    dx = out if out is not None else _empty((B * T, D), like)
    with _timed("vlad_aggregate_bwd_dx", (B, T, D, K)):
        lib.check(lib._lpm_vlad_aggregate_bwd_tiles_dx(ptr(ws), wsb, ptr(dlr), ptr(wtt), B, T, D, K, ptr(dx), dx.stride(0), 0,
                                                       stream_ptr()), "lpm_vlad_aggregate_bwd_tiles_dx")
    return dx


def _assign_gemm_dw_tiles(lib, x, xt, dl, B, T, D, K):
    """dW = x^T . dl on the bf16 pipe (frame tiles of both operands, reduction over every frame of every clip)."""
    st = stream_ptr()
    if xt is None:
        xt = _tile_buffer(lib._lpm_xt_bytes(B, T, D), x)
        lib.check(lib._lpm_split_frames(ptr(x), x.stride(0), B, T, D, ptr(xt), st), "lpm_split_frames")
    dlt = _tile_buffer(lib._lpm_xt_bytes(B, T, K), x)
    lib.check(lib._lpm_split_frames(ptr(dl), dl.stride(0), B, T, K, ptr(dlt), st), "lpm_split_frames")
    dW = _empty((D, K), x)
    wsb = lib._lpm_assign_gemm_tiles_bwd_dw_workspace_bytes(B, T, D, K)
    ws = _tile_buffer(max(wsb, 16), x)
    with _timed("assign_gemm_bwd_dw", (B * T, D, K)):
        lib.check(lib._lpm_assign_gemm_tiles_bwd_dw(ptr(xt), ptr(dlt), B, T, D, K, ptr(dW), ptr(ws), wsb, st),
                  "lpm_assign_gemm_tiles_bwd_dw")
    return dW


def _assign_gemm_dx_operands(lib, W, dl, B, T, D, K):
    """Row tiles of dl and weight tiles of W^T: the operands of dx += dl . W^T."""
    st = stream_ptr()
    dlr = _tile_buffer(lib._lpm_row_tiles_bytes(B, T, K), dl)
    wtt = _tile_buffer(lib._lpm_weight_tiles_bytes(K, D), dl)
    lib.check(lib._lpm_split_rows_tiles(ptr(dl), dl.stride(0), B, T, K, ptr(dlr), st), "lpm_split_rows_tiles")
    lib.check(lib._lpm_split_weight_tiles(ptr(W), K, D, 1, ptr(wtt), st), "lpm_split_weight_tiles")
    return dlr, wtt


class _NetVLAD(torch.autograd.Function):
    """x [B*T, D] -> pooled descriptor.  cluster_weights [D,K]; cluster_bn (gamma, beta, moving) or
    cluster_biases; cluster_weights2 [1,D,K] (None = LightVLAD).  frame_level_models.py:2773-2824."""

    @staticmethod
    def forward(ctx, x, W, gamma, beta, moving_mean, moving_var, bias, W2, T, is_training, kmajor, in_gamma=None, in_beta=None,
                storage="f32", lazy=False, out_slot=None):
        """in_gamma / in_beta ([D] slices of input_bn's gamma / beta): x is input_bn's output for these columns and needs no
        gradient of its own -- the backward then returns input_bn's gamma / beta gradients in closed form instead of dx."""
        lib = _capi.load()
        x = _rows(x, "reshaped_input")
        W = _f32(W, "cluster_weights").contiguous()
        M, D = x.shape
        K = W.shape[1]
        if M % T:
            raise LpmError(f"rows {M} not divisible by max_frames {T}")
        B = M // T
        ctx.storage = storage
        ctx.lazy = bool(lazy)
        if out_slot is not None and storage != "bf16":
            raise LpmError("netvlad: an output slot is taken by the bf16-storage form only")
        if storage == "bf16":
            return _NetVLAD._forward_bf16(ctx, lib, x, W, gamma, beta, moving_mean, moving_var, bias, W2, B, T, D, K, is_training, kmajor,
                                          in_gamma, in_beta, out_slot, lazy)
        if storage != "f32":
            raise LpmError(f"unknown storage {storage!r} (f32 | bf16)")
        _materialised(x, "reshaped_input")
        logits = _empty((M, K), x)
        if ASSIGN_PRECISION not in ("bf16x3", "f32"):
            raise LpmError(f"unknown LPM_ASSIGN_PRECISION {ASSIGN_PRECISION!r} (bf16x3 | f32)")
        tiles = ASSIGN_PRECISION == "bf16x3" and bool(lib._lpm_assign_gemm_tiles_supported(T, D, K))
        xr = None
        if tiles:
            st = stream_ptr()
            nblk = lib._lpm_assign_gemm_tiles_nblk(B, T)
            partial = _empty((nblk, 2, K), x)
            xr = _cached_tiles(x, B, T, D, rows=True)
            if xr is None:
                xr = _tile_buffer(lib._lpm_row_tiles_bytes(B, T, D), x)
                with _timed("split_rows_tiles", (M, D)):
                    lib.check(lib._lpm_split_rows_tiles(ptr(x), x.stride(0), B, T, D, ptr(xr), st), "lpm_split_rows_tiles")
            # NOT from the step's weight pack: K1 streams its 1 MB of weight tiles once per workgroup, a fresh 16 KB per step that all
            # workgroups of an XCD want at the same moment -- its loop is bound by the latency of that first touch.  Written right here
            # the tiles are still in the memory-side cache when K1 starts (42.9 us); written at the top of the step, 700 MB of frame
            # preparation earlier, they come from HBM (54.8 us; measured in one process, LPM_WEIGHT_PACK A/B, round 4).  (Reading them into every XCD's L2 with
            # a small launch in front of K1 on top of that: no effect, 43.5-45.2 vs 43.9-44.6 us.)
            wt = _weight_tiles(W, D, K, False, x, pack=False)
            _k1_selfcheck(lib, 2, x.device)
            with _timed("assign_gemm_fwd", (M, D, K)):
                lib.check(lib._lpm_assign_gemm_tiles_fwd(ptr(xr), ptr(wt), B, T, D, K, ptr(logits), ptr(partial), st),
                          "lpm_assign_gemm_tiles_fwd")
        else:
            nblk = lib._lpm_assign_gemm_nblk(M)
            partial = _empty((nblk, 2, K), x)
            with _timed("assign_gemm_fwd", (M, D, K)):
                lib.check(lib._lpm_assign_gemm_fwd(ptr(x), x.stride(0), ptr(W), M, D, K, 0, ptr(logits), ptr(partial),
                                                   stream_ptr()), "lpm_assign_gemm_fwd")
        mean = var = None
        use_bn = gamma is not None
        if use_bn:
            if is_training:
                mean, var, scale, shift = bn_fold(partial, nblk, K, M, gamma, beta, moving_mean, moving_var)
            else:
                scale, shift = folded_eval_affine(gamma, beta, moving_mean, moving_var)
                scale, shift = scale.contiguous(), shift.contiguous()
                mean, var = moving_mean.detach().clone(), moving_var.detach().clone()
        else:
            scale, shift = None, bias.contiguous()
        flags = LPM_VLAD_SOFTMAX | (LPM_VLAD_RESIDUAL if W2 is not None else 0)
        centres = W2.reshape(D, K).contiguous() if W2 is not None else None
        ctx.nrm_raw = _nrm_raw_ok(lib, T, D, K)
        ctx.no_dx = in_gamma is not None and tiles and _bwd_tiles_ok(lib, T, D, K)
        if lazy and not (kmajor and ctx.nrm_raw):
            raise LpmError("netvlad: the lazily normalised descriptor needs the k-major layout and the tile forms of K2 / K3")
        out, nrm, asum, colsq, csq, gsq, xt = _aggregate_fwd(lib, logits, scale, shift, x, centres, B, T, D, K, flags, kmajor,
                                                             nrm_raw=ctx.nrm_raw, save_u=any(ctx.needs_input_grad), lazy=lazy)
        ctx.dims = (B, T, D, K, flags, kmajor, use_bn, is_training, W2 is not None, tiles)
        ctx.dx_slot = getattr(x, "_lpm_dx_slot", None)
        if in_gamma is not None and not ctx.no_dx:
            raise LpmError("netvlad: the input_bn gradient shortcut needs the tile (bf16x3) forms of K1 and K3 for this shape")
        if lazy:          # nrm IS out (the raw sums): saving an output as such would keep the graph alive through it
            rs = out._lpm_row_scale
            nrm = out.detach()
            ctx.save_for_backward(x, W, logits, scale, shift, mean, var, gamma, centres, nrm, asum, colsq, csq, gsq, xt, xr,
                                  in_gamma, in_beta)
            out._lpm_row_scale = rs
            return out
        ctx.save_for_backward(x, W, logits, scale, shift, mean, var, gamma, centres, nrm, asum, colsq, csq, gsq, xt, xr,
                              in_gamma, in_beta)
        return out

    @staticmethod
    def _forward_bf16(ctx, lib, x, W, gamma, beta, moving_mean, moving_var, bias, W2, B, T, D, K, is_training, kmajor, in_gamma, in_beta,
                      out_slot=None, lazy=False):
        """bf16 storage (BASELINE cfg-5; include/lpm_hip.h "bf16 storage"): frames, logits / assignment and the descriptor are bf16
        in HBM, every product one bf16 MFMA with fp32 accumulation; statistics, norms and gradients fp32.  Needs the operand tiles
        ops.frame_sample_bn(storage="bf16") wrote for x, the LDS-shared K2 form (D, K multiples of 128, K <= 512), the d-major
        layout, and frames that need no gradient of their own (input_affine, or x.requires_grad False)."""
        st = stream_ptr()
        M = B * T
        if kmajor:
            raise LpmError("netvlad: bf16 storage writes the descriptor in the reference's d-major layout only")
        if not (lib._lpm_vlad_tiles3_supported(D, K) and lib._lpm_assign_gemm_tiles_supported(T, D, K) and K % 8 == 0):
            raise LpmError(f"netvlad: bf16 storage needs D, K multiples of 128 and K <= 512 (D={D} K={K})")
        if in_gamma is None and ctx.needs_input_grad[0]:
            raise LpmError("netvlad: bf16 storage has no input-gradient path (pass input_affine, or frames that need no gradient)")
        xr = _cached_tiles(x, B, T, D, rows=True, storage="bf16")
        xt = _cached_tiles(x, B, T, D, storage="bf16")
        if xr is None or xt is None:
            raise LpmError("netvlad: bf16 storage needs the operand tiles of ops.frame_sample_bn(storage='bf16') for this input")
        nblk = lib._lpm_assign_gemm_tiles_nblk(B, T)
        partial = _empty((nblk, 2, K), W)
        wt = _tile_buffer(lib._lpm_weight_tiles_bytes(D, K) // 2, W)
        lib.check(lib._lpm_split_weight_tiles_bf16(ptr(W), D, K, 0, ptr(wt), st), "lpm_split_weight_tiles_bf16")
        logits = torch.empty((M, K), dtype=torch.bfloat16, device=W.device)
        _k1_selfcheck(lib, 1, W.device)
        with _timed("assign_gemm_fwd", (M, D, K)):
            lib.check(lib._lpm_assign_gemm_tiles_fwd_bf16(ptr(xr), ptr(wt), B, T, D, K, ptr(logits), ptr(partial), st),
                      "lpm_assign_gemm_tiles_fwd_bf16")
        mean = var = None
        use_bn = gamma is not None
        if use_bn:
            if is_training:
                mean, var, scale, shift = bn_fold(partial, nblk, K, M, gamma, beta, moving_mean, moving_var)
            else:
                scale, shift = folded_eval_affine(gamma, beta, moving_mean, moving_var)
                scale, shift = scale.contiguous(), shift.contiguous()
                mean, var = moving_mean.detach().clone(), moving_var.detach().clone()
        else:
            scale, shift = None, bias.contiguous()
        flags = LPM_VLAD_SOFTMAX | (LPM_VLAD_RESIDUAL if W2 is not None else 0)
        centres = W2.reshape(D, K).contiguous() if W2 is not None else None
        steps = lib._lpm_frame_steps_bf16(T)
        at = torch.empty(B * (K // 32) * steps * 256, dtype=torch.int32, device=W.device)       # 1 KB per (cluster tile, step)
        with _timed("assign_tiles", (B, T, K)):
            lib.check(lib._lpm_assign_tiles_bf16(ptr(logits), ptr(scale), ptr(shift), B, T, K, flags, ptr(at), st), "lpm_assign_tiles_bf16")
        nrm = torch.empty((B, D, K), dtype=torch.bfloat16, device=W.device)       # the un-normalised sums, bf16 like the descriptor
        asum, colsq, csq = (_empty((B, K), W) for _ in range(3))
        # round 6: clip-wide items (256 clusters x a third of the columns per workgroup, csrc/vlad_clip16.hip) where the shape allows
        # (K a multiple of 256, D >= 384: BASELINE configs[4]'s video stream); the 128 x 128 form otherwise (its audio stream)
        P = lib._lpm_vlad_clip16_slabs(D, K) if VLAD_CLIP16 else 0
        clip16 = P > 0
        if not clip16:
            P = D // 128
        part = _empty((B, P, K), W)
        with _timed("vlad_aggregate_fwd", (B, T, D, K)):
            if clip16:
                lib.check(lib._lpm_vlad_aggregate_clip_fwd_bf16(ptr(at), ptr(xt), ptr(centres), B, T, D, K, flags, ptr(nrm), ptr(asum),
                                                                ptr(part), st), "lpm_vlad_aggregate_clip_fwd_bf16")
            else:
                lib.check(lib._lpm_vlad_aggregate_tiles3_fwd_bf16(ptr(at), ptr(xt), ptr(centres), B, T, D, K, flags, ptr(nrm), ptr(asum),
                                                                  ptr(part), st), "lpm_vlad_aggregate_tiles3_fwd_bf16")
        gsq = _empty((B,), W)
        if lazy:
            # LAZILY NORMALISED (d-major, bf16 sums): no finalize pass -- lpm_vlad_row_scales turns the partial norms into one factor per
            # (clip, cluster) and the consumer (ops.projection_parts) applies it where it reads the sums.  What is returned is an fp32
            # HANDLE of the descriptor's shape that nobody may read (never written: autograd wants a gradient of the handle's dtype, and
            # the projection's input gradient is fp32); the sums travel as its ``_lpm_raw``.
            if out_slot is not None or K % 32:
                raise LpmError("netvlad: the lazily normalised bf16 descriptor takes no output slot and needs K % 32 == 0")
            rs = _empty((B, K), W)
            with _timed("vlad_finalize", (B, D, K)):
                lib.check(lib._lpm_vlad_row_scales(ptr(part), P, B, K, ptr(rs), ptr(colsq), ptr(csq), ptr(gsq), st), "lpm_vlad_row_scales")
            out = _empty((B, D * K), W)
            out._lpm_row_scale, out._lpm_scale_ks, out._lpm_raw = rs, K, nrm.view(B, D * K)
        elif out_slot is not None:
            # the consumer (the hidden projection) computes in fp32: the normalised descriptor leaves the finalize pass as fp32 straight
            # into its column slot of the streams' joined buffer -- no bf16 copy, no concat, no casts in either direction
            out = out_slot.view()                    # [B, 1, D * K], clips out_slot.base.shape[1] elements apart
            if tuple(out.shape) != (B, 1, D * K) or out.dtype != torch.float32:
                raise LpmError("netvlad: the output slot does not match [B, 1, D * K] fp32")
            with _timed("vlad_finalize", (B, D, K)):
                lib.check(lib._lpm_vlad_finalize2_fwd_ld(ptr(nrm), ptr(part), P, B, D, K, LPM_VLAD_NRM_RAW | _capi.LPM_VLAD_NRM_BF16, ptr(out),
                                                         out.stride(0), ptr(colsq), ptr(csq), ptr(gsq), st), "lpm_vlad_finalize2_fwd_ld")
        else:
            out = torch.empty((B, D * K), dtype=torch.bfloat16, device=W.device)
            with _timed("vlad_finalize", (B, D, K)):
                lib.check(lib._lpm_vlad_finalize2_fwd(ptr(nrm), ptr(part), P, B, D, K,
                                                      LPM_VLAD_NRM_RAW | _capi.LPM_VLAD_OUT_BF16 | _capi.LPM_VLAD_NRM_BF16, ptr(out),
                                                      ptr(colsq), ptr(csq), ptr(gsq), st), "lpm_vlad_finalize2_fwd")
        ctx.nrm_raw = True
        ctx.dims = (B, T, D, K, flags, False, use_bn, is_training, W2 is not None, True)
        ctx.no_dx = True
        ctx.has_affine = in_gamma is not None
        ctx.save_for_backward(x, W, logits, scale, shift, mean, var, gamma, centres, nrm, asum, colsq, csq, gsq, xt, xr, in_gamma, in_beta)
        return out

    @staticmethod
    def _backward_bf16(ctx, dout):
        lib = _capi.load()
        st = stream_ptr()
        B, T, D, K, flags, kmajor, use_bn, is_training, has_w2, _ = ctx.dims
        x, W, logits, scale, shift, mean, var, gamma, centres, nrm, asum, colsq, csq, gsq, xt, xr, in_gamma, in_beta = ctx.saved_tensors
        M = B * T
        # the gradient of a bf16 descriptor arrives as bf16; of a slot (fp32) as a column slice of the joined buffer's gradient, which
        # K3 reads in place (clips dout.stride(0) elements apart)
        if dout.dtype == torch.float32 and ((dout.dim() == 3 and dout.shape[1] == 1) or dout.dim() == 2) and dout.stride(-1) == 1 \
                and dout.stride(0) % 4 == 0 and dout.stride(0) >= D * K and dout.data_ptr() % 16 == 0:
            dob = dout.stride(0)
        else:
            dout = dout.float().contiguous()
            dob = D * K
        dlt = _empty((M, K), W)
        dcentres = _empty((D, K), W)
        g0 = _empty((B, D), W)
        wsb = lib._lpm_vlad_bwd_tiles_workspace_bytes(B, T, D, K)
        ws = _tile_buffer(wsb, W)
        fl = flags | LPM_VLAD_NRM_RAW | _capi.LPM_VLAD_TILES_BF16
        with _timed("vlad_aggregate_bwd", (B, T, D, K)):
            lib.check(lib._lpm_vlad_aggregate_bwd_tiles_ld(ptr(dout), dob, ptr(nrm), ptr(asum), ptr(colsq), ptr(csq), ptr(gsq), ptr(logits),
                                                           ptr(scale), ptr(shift), ptr(xr), ptr(centres), B, T, D, K, fl, ptr(dlt),
                                                           ptr(dcentres), ptr(g0), ptr(ws), wsb, st), "lpm_vlad_aggregate_bwd_tiles_ld")
        dgamma = dbeta = dbias = None
        if use_bn and is_training:
            dgamma, dbeta = _empty((K,), W), _empty((K,), W)
            wsb2 = lib._lpm_bn_bwd_workspace_bytes(M, K)
            ws2 = torch.empty(wsb2 // 4, dtype=torch.float32, device=W.device)
            if logits.dtype == torch.bfloat16 and logits.is_contiguous():          # read in place (no fp32 copy of the [B T, K] logits)
                lib.check(lib._lpm_bn_bwd_x16(ptr(dlt), ptr(logits), ptr(mean), ptr(var), ptr(gamma), BN_EPS, M, K, ptr(dlt), ptr(dgamma),
                                              ptr(dbeta), ptr(ws2), wsb2, st), "lpm_bn_bwd_x16")
            else:
                lf = logits.float().contiguous()
                lib.check(lib._lpm_bn_bwd(ptr(dlt), ptr(lf), ptr(mean), ptr(var), ptr(gamma), BN_EPS, M, K, ptr(dlt), ptr(dgamma), ptr(dbeta),
                                          ptr(ws2), wsb2, st), "lpm_bn_bwd")
            dl = dlt
        elif use_bn:
            lhat = (logits.float() - mean) * torch.rsqrt(var + BN_EPS)
            dgamma, dbeta = (dlt * lhat).sum(0), dlt.sum(0)
            dl = dlt * scale
        else:
            dbias = dlt.sum(0)
            dl = dlt
        dlt16 = _tile_buffer(lib._lpm_frame_tiles_bf16_bytes(B, T, K), W)
        lib.check(lib._lpm_split_frames_bf16(ptr(dl), dl.stride(0), B, T, K, ptr(dlt16), st), "lpm_split_frames_bf16")
        dW = _empty((D, K), W)
        wsb3 = lib._lpm_assign_gemm_tiles_bwd_dw_workspace_bytes(B, T, D, K)
        ws3 = _tile_buffer(max(wsb3, 16), W)
        with _timed("assign_gemm_bwd_dw", (M, D, K)):
            lib.check(lib._lpm_assign_gemm_tiles_bwd_dw_bf16(ptr(xt), ptr(dlt16), B, T, D, K, ptr(dW), ptr(ws3), wsb3, st),
                      "lpm_assign_gemm_tiles_bwd_dw_bf16")
        d_in_gamma = d_in_beta = None
        if ctx.has_affine:
            cs = None if (use_bn and is_training) else dl.sum(0)
            d_in_gamma, d_in_beta = _empty((D,), W), _empty((D,), W)
            lib.check(lib._lpm_input_bn_grads(ptr(dcentres), ptr(centres) if has_w2 else None, ptr(W), ptr(dW), ptr(g0), ptr(cs),
                                              ptr(in_gamma.contiguous()), ptr(in_beta.contiguous()), B, D, K, ptr(d_in_gamma),
                                              ptr(d_in_beta), st), "lpm_input_bn_grads")
        dW2 = dcentres.reshape(1, D, K) if has_w2 else None
        return None, dW, dgamma, dbeta, None, None, dbias, dW2, None, None, None, d_in_gamma, d_in_beta, None, None, None

    @staticmethod
    def backward(ctx, dout):
        if ctx.storage == "bf16":
            return _NetVLAD._backward_bf16(ctx, dout)
        lib = _capi.load()
        B, T, D, K, flags, kmajor, use_bn, is_training, has_w2, tiles = ctx.dims
        x, W, logits, scale, shift, mean, var, gamma, centres, nrm, asum, colsq, csq, gsq, xt, xr, in_gamma, in_beta = ctx.saved_tensors
        dout = _f32(dout, "dout").contiguous()
        k3_tiles = _bwd_tiles_ok(lib, T, D, K)
        no_dx = ctx.no_dx
        if ctx.lazy and not no_dx:
            raise LpmError("netvlad: the backward of the lazily normalised descriptor needs frames without a gradient of their own "
                           "(input_affine): K3's k-major form has no input-gradient path")
        if ctx.nrm_raw and not k3_tiles:
            raise LpmError("netvlad: the forward left nrm un-normalised for the tile backward, which is no longer selected")
        if k3_tiles:
            dlt, dcentres, wspace, g0 = _aggregate_bwd_tiles(lib, dout, nrm, asum, colsq, csq, gsq, logits, scale, shift, x, xr,
                                                             centres, B, T, D, K, flags, kmajor, no_dx=no_dx, nrm_raw=ctx.nrm_raw,
                                                             raw_kmajor=ctx.lazy)
            dx = None
            if DEBUG_TAP is not None and D == 1024:
                head = wspace[0].view(torch.float32)       # K3's workspace: dots [B][16 slots][3][K] (k-major form: the first B*3*K) | u | v | ctil
                DEBUG_TAP.update(dout=dout.clone(), dlt_raw=dlt.clone(), dots=head[:B * 3 * K].clone() if ctx.lazy else None,
                                 u_v_ctil=head[B * 16 * 3 * K:B * 16 * 3 * K + 3 * B * K].clone(), xr=xr.clone(), logits=logits.clone(),
                                 scale=scale.clone(), shift=shift.clone(), nrm=nrm.clone(), colsq=colsq.clone(), dcentres=dcentres.clone(),
                                 centres=centres.clone() if centres is not None else None, g0=g0.clone() if g0 is not None else None)
        else:
            dlt, dx, dcentres = _aggregate_bwd(lib, dout, nrm, asum, colsq, csq, gsq, logits, scale, shift, x, centres, B, T, D,
                                               K, flags, kmajor)
        M = B * T
        dgamma = dbeta = dbias = None
        if use_bn and is_training:
            dgamma, dbeta = _empty((K,), x), _empty((K,), x)
            wsb = lib._lpm_bn_bwd_workspace_bytes(M, K)
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device)
            lib.check(lib._lpm_bn_bwd(ptr(dlt), ptr(logits), ptr(mean), ptr(var), ptr(gamma), BN_EPS, M, K, ptr(dlt),
                                      ptr(dgamma), ptr(dbeta), ptr(ws), wsb, stream_ptr()), "lpm_bn_bwd")
            dl = dlt
            if DEBUG_TAP is not None and D == 1024:
                DEBUG_TAP.update(dgamma=dgamma.clone(), dbeta=dbeta.clone(), dl=dl.clone(), bn_ws=ws.clone())
        elif use_bn:   # inference-mode statistics are constants
            lhat = (logits - mean) * torch.rsqrt(var + BN_EPS)
            dgamma, dbeta = (dlt * lhat).sum(0), dlt.sum(0)
            dl = dlt * scale
        else:
            dbias = dlt.sum(0)
            dl = dlt
        # backward of the soft-assignment GEMM; with K3 in tile form its dx term rides in K3's dx pass
        dlr = wtt = None
        if tiles:
            dW = _assign_gemm_dw_tiles(lib, x, xt, dl, B, T, D, K)
            if not no_dx:
                dlr, wtt = _assign_gemm_dx_operands(lib, W, dl, B, T, D, K)
        else:   # plain fp32 library GEMM (hipBLASLt through torch)
            dW = x.t().matmul(dl)
        if no_dx:
            # x = gamma_in * xhat + beta_in straight out of input_bn and nobody wants d/dx itself.  With V_b = sum_t a x (= U_b +
            # asum_b W2) and dx = sum_k a dU_b + dl W^T (App. F.2):
            #   d beta_in [c]            = sum_r dx[r,c]            = sum_b sum_k asum_b[k] dU_b[c,k] + sum_k W[c,k] colsum(dl)[k]
            #   gamma_in[c] d gamma_in[c] = sum_r dx[r,c] (x - beta) = sum_b sum_k dU_b[c,k] V_b[c,k] + sum_k W[c,k] dW[c,k] - beta_in d beta_in
            # and sum_b asum_b dU_b = -dcentres, sum_b sum_k dU_b U_b = g0.sum(0), sum_r x dl = dW: everything is already there
            # (dx, a [B*T, D] GEMM pair plus a pass over the frames, is not formed).  colsum(dl) = 0 after a training-mode
            # batch norm (its backward removes the batch mean).
            cs = None if (use_bn and is_training) else dl.sum(0)
            d_in_gamma, d_in_beta = _empty((D,), x), _empty((D,), x)
            lib.check(lib._lpm_input_bn_grads(ptr(dcentres), ptr(centres) if has_w2 else None, ptr(W), ptr(dW), ptr(g0), ptr(cs),
                                              ptr(in_gamma.contiguous()), ptr(in_beta.contiguous()), B, D, K, ptr(d_in_gamma),
                                              ptr(d_in_beta), stream_ptr()), "lpm_input_bn_grads")
            dW2 = dcentres.reshape(1, D, K) if has_w2 else None
            return None, dW, dgamma, dbeta, None, None, dbias, dW2, None, None, None, d_in_gamma, d_in_beta, None, None, None
        if k3_tiles:
            dx = _aggregate_bwd_tiles_dx(lib, wspace, dlr, wtt, x, B, T, D, K, out=_dx_slot_view(ctx.dx_slot, D))
            if not tiles:
                dx.addmm_(dl, W.t())
        elif tiles:
            lib.check(lib._lpm_assign_gemm_tiles_bwd_dx(ptr(dlr), ptr(wtt), B, T, D, K, ptr(dx), dx.stride(0), stream_ptr()),
                      "lpm_assign_gemm_tiles_bwd_dx")
        else:
            dx.addmm_(dl, W.t())
        dW2 = dcentres.reshape(1, D, K) if has_w2 else None
        return dx, dW, dgamma, dbeta, None, None, dbias, dW2, None, None, None, None, None, None, None, None


class _Materialise(torch.autograd.Function):
    """raw [B, K, D] x row_scale [B, K] -> the normalised descriptor as an ordinary tensor (ks > 0: raw is d-major, [B, D * ks]).  The
    gradient passes through UNCHANGED: by the contract of the lazily normalised form (netvlad(lazy=True), vlad_aggregate(lazy=True)) the
    gradient a consumer returns for `raw` IS the gradient with respect to the normalised descriptor -- the Jacobian of both
    normalisations lives in the pooling op's backward (K3)."""

    @staticmethod
    def forward(ctx, raw, row_scale, ks=0, stored=None):
        """stored: the sums when ``raw`` is only a handle (bf16 storage: _NetVLAD._forward_bf16, lazy)."""
        if stored is not None:
            raw = stored.float()
        if ks:
            B = raw.shape[0]
            return (raw.reshape(B, -1, ks) * row_scale.unsqueeze(1)).reshape(raw.shape)
        return raw * row_scale.unsqueeze(-1)

    @staticmethod
    def backward(ctx, d):
        return d, None, None, None


def row_scale_of(x):
    """The [B, K] row scale of a lazily normalised descriptor (None for an ordinary tensor)."""
    return getattr(x, "_lpm_row_scale", None)


def materialise(x):
    """An ordinary tensor for any consumer that does not apply the row scale itself."""
    rs = row_scale_of(x)
    return x if rs is None else _Materialise.apply(x, rs, int(getattr(x, "_lpm_scale_ks", 0)), getattr(x, "_lpm_raw", None))


def netvlad_lazy_ok(T, D, K):
    """Shapes for which netvlad(kmajor=True, lazy=True) exists: the LDS-shared K2 form and the tile form of K3."""
    lib = _capi.load()
    return (VLAD_PRECISION == "bf16x3" and VLAD_TILES3 and bool(lib._lpm_vlad_tiles3_supported(D, K)) and K <= 512
            and netvlad_input_shortcut_ok(T, D, K))


def netvlad(x, cluster_weights, cluster_weights2, max_frames, bn=None, bias=None, is_training=True, kmajor=False,
            input_affine=None, storage="f32", lazy=False, out_slot=None):
    """bn = (gamma, beta, moving_mean, moving_var) or None (then ``bias`` = cluster_biases).  input_affine = (gamma, beta) slices
    of the input batch norm whose output x is (x itself then needs no gradient): see _NetVLAD.forward.  storage="bf16": the
    descriptor comes back as bf16 [B, D*K] (see _NetVLAD._forward_bf16) -- or, with out_slot (an ops.OutputSlot of L = 1, F = D*K), as
    fp32 written straight into that column slot of the streams' joined buffer (ops.DescriptorSlots), returned as the slot's view.
    lazy (with kmajor; netvlad_lazy_ok shapes; frames without a gradient): the result is the LAZILY NORMALISED descriptor -- the
    un-normalised residual sums [B, K, D], written once by the aggregation kernel, carrying ``_lpm_row_scale`` [B, K] =
    1 / (n_k sqrt(g)); descriptor = result * scale per (clip, cluster) row (frame_level_models.py:2819-2822).  Consumers that know
    (ops.attention_block_x3: the operand split of its q/k/v GEMM and its residual layer norm) apply the scale where they read the
    rows; the gradient they return for it is the gradient with respect to the normalised descriptor.  Everybody else goes through
    ops.materialise.  No finalize pass, no transposes in the backward."""
    g, b, mm, mv = bn if bn is not None else (None, None, None, None)
    ig, ib = input_affine if input_affine is not None else (None, None)
    return _NetVLAD.apply(x, cluster_weights, g, b, mm, mv, bias, cluster_weights2, int(max_frames), bool(is_training),
                          bool(kmajor), ig, ib, storage, bool(lazy), out_slot)


def netvlad_input_shortcut_ok(T, D, K):
    """The input_bn gradient shortcut needs K1 and K3 in their tile (split-bf16) forms for this shape."""
    lib = _capi.load()
    return (ASSIGN_PRECISION == "bf16x3" and bool(lib._lpm_assign_gemm_tiles_supported(T, D, K)) and _bwd_tiles_ok(lib, T, D, K))


class _VladAggregate(torch.autograd.Function):
    """Similarities given (no softmax): NetVladAttenCluster tail, video_pooling_modules.py:1641-1658."""

    @staticmethod
    def forward(ctx, sims, x, centres, T, kmajor, lazy=False, grad_join=None):
        """grad_join (ops.GradJoin, accepted by the frame encoder's node): the gradient of x goes there instead of to autograd."""
        lib = _capi.load()
        ctx.join = grad_join if (grad_join is not None and grad_join.accepted) else None
        x = _rows(x, "inputs")
        M, D = x.shape
        K = centres.shape[1]
        B = M // T
        sims2 = _f32(sims, "cluster_similarities").reshape(M, K).contiguous()
        centres = _f32(centres, "cluster_centers").contiguous()
        flags = LPM_VLAD_RESIDUAL
        ctx.nrm_raw = _nrm_raw_ok(lib, T, D, K)
        if lazy:
            # the LAZILY NORMALISED d-major descriptor: K2 writes the un-normalised sums [B, D, K] once, lpm_vlad_row_scales turns the partial
            # norms into one factor per (clip, cluster); the consumer (ops.projection_parts) applies it where it reads the operand --
            # no finalize pass.  The backward is the ordinary one: K3 reads the un-normalised sums anyway (nrm_raw).
            if kmajor or not vlad_aggregate_lazy_ok(T, D, K):
                raise LpmError("vlad_aggregate: the lazily normalised d-major descriptor needs the LDS-shared K2 form and K3's tile form")
            st = stream_ptr()
            xt = _cached_tiles(x, B, T, D)
            if xt is None:
                xt = torch.empty(lib._lpm_xt_bytes(B, T, D) // 4, dtype=torch.int32, device=x.device)
                with _timed("split_frames", (B, T, D)):
                    lib.check(lib._lpm_split_frames(ptr(x), x.stride(0), B, T, D, ptr(xt), st), "lpm_split_frames")
            at = torch.empty(lib._lpm_at_bytes(B, T, K) // 4, dtype=torch.int32, device=x.device)
            with _timed("assign_tiles", (B, T, K)):
                lib.check(lib._lpm_assign_tiles(ptr(sims2), None, None, B, T, K, flags, ptr(at), st), "lpm_assign_tiles")
            raw = _empty((B, D * K), x)
            asum, colsq, csq, rs = (_empty((B, K), x) for _ in range(4))
            gsq = _empty((B,), x)
            # K = 256: the clip-wide items of vlad_clip.hip (frame tiles once per clip), leaving the sums d-major straight from the
            # accumulators; otherwise the 128 x 128 form
            Pc = lib._lpm_vlad_clip_slabs(D, K) if VLAD_CLIP else 0
            P = Pc if Pc else D // 128
            part = _empty((B, P, K), x)
            with _timed("vlad_aggregate_fwd", (B, T, D, K)):
                if Pc:
                    lib.check(lib._lpm_vlad_aggregate_clip_dmajor_fwd(ptr(at), ptr(xt), ptr(centres), B, T, D, K, flags, ptr(raw), ptr(asum),
                                                                      ptr(part), st), "lpm_vlad_aggregate_clip_dmajor_fwd")
                else:
                    lib.check(lib._lpm_vlad_aggregate_tiles3_fwd(ptr(at), ptr(xt), ptr(centres), B, T, D, K, flags, ptr(raw), ptr(asum),
                                                                 ptr(part), st), "lpm_vlad_aggregate_tiles3_fwd")
            with _timed("vlad_finalize", (B, D, K)):
                lib.check(lib._lpm_vlad_row_scales(ptr(part), P, B, K, ptr(rs), ptr(colsq), ptr(csq), ptr(gsq), st), "lpm_vlad_row_scales")
            ctx.dims = (B, T, D, K, flags, kmajor, sims.shape)
            ctx.save_for_backward(sims2, x, centres, raw.detach(), asum, colsq, csq, gsq)      # (raw IS the output: see _NetVLAD.forward)
            raw._lpm_row_scale, raw._lpm_scale_ks = rs, K
            return raw
        out, nrm, asum, colsq, csq, gsq, _ = _aggregate_fwd(lib, sims2, None, None, x, centres, B, T, D, K, flags, kmajor,
                                                            nrm_raw=ctx.nrm_raw, save_u=any(ctx.needs_input_grad))
        ctx.dims = (B, T, D, K, flags, kmajor, sims.shape)
        ctx.save_for_backward(sims2, x, centres, nrm, asum, colsq, csq, gsq)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _capi.load()
        B, T, D, K, flags, kmajor, sshape = ctx.dims
        sims2, x, centres, nrm, asum, colsq, csq, gsq = ctx.saved_tensors
        if ctx.nrm_raw and not _bwd_tiles_ok(lib, T, D, K):
            raise LpmError("vlad_aggregate: the forward left nrm un-normalised for the tile backward, which is no longer selected")
        if _bwd_tiles_ok(lib, T, D, K):
            dsims, dcentres, wspace, _ = _aggregate_bwd_tiles(lib, _f32(dout, "dout"), nrm, asum, colsq, csq, gsq, sims2, None, None, x,
                                                              None, centres, B, T, D, K, flags, kmajor, nrm_raw=ctx.nrm_raw)
            dx = _aggregate_bwd_tiles_dx(lib, wspace, None, None, x, B, T, D, K)
        else:
            dsims, dx, dcentres = _aggregate_bwd(lib, dout.contiguous(), nrm, asum, colsq, csq, gsq, sims2, None, None, x, centres,
                                                 B, T, D, K, flags, kmajor)
        if ctx.join is not None and dx is not None and ctx.needs_input_grad[1]:
            ctx.join.put(dx)                      # (the encoder's node, which runs after this one, adds it to the frames' other gradient)
            dx = None
        return dsims.reshape(sshape), dx, dcentres, None, None, None, None


def vlad_aggregate_lazy_ok(T, D, K):
    """Shapes for which vlad_aggregate(lazy=True) exists: the LDS-shared K2 form (D, K multiples of 128), K3's tile form, K a multiple of 32."""
    lib = _capi.load()
    return (VLAD_PRECISION == "bf16x3" and VLAD_TILES3 and bool(lib._lpm_vlad_tiles3_supported(D, K)) and _bwd_tiles_ok(lib, T, D, K)
            and K % 32 == 0 and K <= 1024)


class GradJoin:
    """One tensor read by TWO nodes of one module, handed to both explicitly (NetVladAttenCluster: the frames feed the frame encoder AND the
    aggregation, video_pooling_modules.py:1623-1658).  Autograd would add the two gradients in a pass of its own ([24 000, 1 024] at
    cfg-3: 49 us).  The encoder's output feeds the aggregation, so in every backward pass the aggregation's node runs FIRST: it ``put``s its
    gradient of the frames here instead of returning it, and the encoder's attention-half node (ops._AttnBlockBNX3) -- which ``accept``ed
    the join in its forward -- takes it as the second gradient of its residual tensor: added on the layer-norm backward's store, carried
    into the q/k/v input-gradient GEMM's beta = 1 operand.  Not accepted (the encoder took another path): the aggregation returns its
    gradient to autograd as always."""

    puts = 0                 # (diagnostics / tests: gradients that took this route in the process)

    def __init__(self):
        self.accepted = False
        self.dx = None
        self.taken = False

    def accept(self):
        self.accepted, self.dx, self.taken = True, None, False

    def put(self, dx):
        if self.taken:
            raise LpmError("GradJoin: the accepting node's backward ran BEFORE the gradient it was to take arrived -- the two readers are not "
                           "ordered as the module that joined them assumed; this gradient would have been lost")
        if self.dx is not None:
            raise LpmError("GradJoin: a second gradient arrived before the first was taken (two backward passes over one forward?)")
        self.dx = dx
        GradJoin.puts += 1

    def take(self):
        dx, self.dx, self.taken = self.dx, None, True
        return dx


GRAD_JOIN = os.environ.get("LPM_GRAD_JOIN", "1") != "0"        # "0": autograd adds the frames' two gradients (A/B)


def vlad_aggregate(sims, x, centres, max_frames, kmajor=False, lazy=False, grad_join=None):
    """lazy (d-major only; vlad_aggregate_lazy_ok shapes): the result is the LAZILY NORMALISED descriptor -- the un-normalised residual sums
    [B, D * K] carrying ``_lpm_row_scale`` [B, K] and ``_lpm_scale_ks`` = K: descriptor[b, d * K + k] = result[b, d * K + k] * scale[b, k]
    (video_pooling_modules.py:1655-1658).  ops.projection_parts applies the scale where it reads the operand; the gradient it returns is the
    gradient with respect to the normalised descriptor.  Everybody else goes through ops.materialise."""
    return _VladAggregate.apply(sims, x, centres, int(max_frames), bool(kmajor), bool(lazy), grad_join)


# ----------------------------------------------------------------------------------------------
# dense layers of the encoders on the bf16 matrix pipe at fp32-grade accuracy (split-bf16 operands)
# ----------------------------------------------------------------------------------------------
class OperandSite:
    """One operand of one dense layer (role "a": its input activation, "g": the gradient of its output) in the step that is running: the
    operand format its producer writes and what its consumers undo (csrc/operand_format.h).  f16: fp16 planes of x * ``scale`` (a power
    of two; ``inv`` = 1 / scale) -- three planes [hi | lo | hi] for an activation (LPM_OPERAND_FP16X3: the forward keeps all three
    terms), two [hi | lo] for a gradient (LPM_OPERAND_FP16X2: the backward products are two-term); otherwise split-bf16 x3, scale 1.
    Either way the producer records max |x| into the site's slot of OperandScales.amax (``fmt`` = the address of the
    LpmOperandFormat the library reads)."""
    __slots__ = ("f16", "scale", "inv", "struct", "fmt", "planes", "dtype", "kind")

    def __init__(self, f16, scale, amax_ptr, role="a"):
        self.f16 = bool(f16)
        self.scale = float(scale) if f16 else 1.0
        self.inv = 1.0 / self.scale
        self.kind = (_capi.LPM_OPERAND_FP16X2 if role == "g" else _capi.LPM_OPERAND_FP16X3) if f16 else _capi.LPM_OPERAND_BF16X3
        self.struct = _capi.OperandFormat(self.kind, self.scale, amax_ptr)
        self.fmt = C.c_void_p(C.addressof(self.struct))
        self.planes = 2 if (f16 and role == "g") else 3
        self.dtype = torch.float16 if f16 else torch.bfloat16


class OperandScales:
    """Delayed power-of-two scales for the fp16 two-product operand format (the recipe of fp8 training: the scale of
    step t comes from the max |x| the producers MEASURED at earlier steps).

    Every operand image of NetVladV1's encoder GEMMs has a site (``site(key)``); its producer -- a split pass, a layer norm, the
    attention kernels, a tile GEMM's epilogue -- records max |x| into the site's slot of ``amax`` in BOTH formats.  ``begin_step(t)``
    (trainer, before the forward): queues the read-back of step t - 1 (device -> pinned host copy + event, then zeroes the buffer:
    stream order puts it behind that step's kernels), makes sure the read-back of step t - 2 HAS ARRIVED (it waits if the host is more
    than two steps ahead of the device -- a step is GPU-bound, so normally it does not), and decides the step's format: fp16 when every
    site that exists was measured at step t - 2 or earlier, else split-bf16 x3 (the first two steps of a run, a model that grew a
    layer) -- a step never mixes formats.  The delay is FIXED: the scales of step t are a function of the maxima of steps t - 3 and
    t - 2 and of nothing else, so a run is bit for bit reproducible (tests/test_gpu_determinism.py) whatever the host's timing.
    The scale puts the larger of those two maxima at 2^TARGET: [2^10, 2^11), a factor 32 below fp16's largest value for growth
    between measurement and use (beyond it values saturate at +-65504, finite) and 2^24 above the smallest normal fp16 number; fp16
    subnormals are kept by the matrix cores (tools/fp16_probe.py), so (hi, lo) degrades gracefully into fixed point below that.
    ``calibrate`` (tests, short runs): measure synchronously from one forward + backward that the caller runs in split-bf16."""
    MAX_SITES = 64
    TARGET = 10
    SLOT = _capi.LPM_OPERAND_AMAX_SUB * _capi.LPM_OPERAND_AMAX_STRIDE       # floats per site: its sub-slots, one cache line apart

    def __init__(self, device):
        self.device = torch.device(device)
        self.amax = torch.zeros(self.MAX_SITES * self.SLOT, dtype=torch.float32, device=self.device)
        self.slots = {}
        self.first_step = []                 # slot -> id of the step it was first asked for in
        self.hist = [[0.0] * self.MAX_SITES, [0.0] * self.MAX_SITES]      # the last two harvested measurements per slot
        self.measured = -1                   # id of the newest step whose measurement the host holds
        self.step = -1                       # id of the running step
        self.pending = []                    # [(step id, pinned tensor, event)] read-backs in flight, oldest first
        self.RING = 4                        # pinned read-back buffers + events, allocated once (a pinned allocation costs ~1 ms)
        self._ring = None
        self._ring_next = 0
        self.fp16_now = False
        self.sites = {}                      # key -> OperandSite of the running step
        self.enabled = True
        self.steps_fp16 = 0
        self.saturation_margin = None        # diagnostics: min over sites of 65504 / (scale * newest amax)

    def _check_finite(self, vals, sid):
        """the fp16 split saturates -- a NaN or Inf entering an encoder GEMM would become a finite +-65504 and go on into Adam
        unseen (split-bf16 and the TF reference would carry it).  The producers' maxima keep it (operand_format.h: of_amax8 orders bit
        patterns, NaN above Inf), and the host refuses to go on when it arrives here -- one to three steps after the fact (the delay of
        the asynchronous read-back), naming the step and the operand sites."""
        import math
        bad = [k for k, sl in self.slots.items() if sl < len(vals) and (math.isnan(vals[sl]) or math.isinf(vals[sl]))]
        if bad:
            raise LpmError(f"OperandScales: NaN / Inf among the operands of the encoders' dense GEMMs in step {sid} "
                           f"(sites {[k[0] for k in bad]}: {len(bad)} of {len(self.slots)}); on fp16 planes such a value is stored as a finite "
                           f"+-65504 -- the steps since then are not to be trusted")

    def _harvest(self, wait=False):
        while self.pending:
            sid, host, ev = self.pending[0]
            if not wait and not ev.query():
                break
            if wait:
                ev.synchronize()
            vals = host.tolist()
            self._check_finite(vals, sid)
            self.hist[0], self.hist[1] = self.hist[1], vals
            self.measured = sid
            self.pending.pop(0)

    def begin_step(self):
        if self.step >= 0:                   # the step that just ended: its maxima go home behind its kernels
            if self._ring is None:
                self._ring = [(torch.empty(self.MAX_SITES, dtype=torch.float32, pin_memory=True), torch.cuda.Event()) for _ in range(self.RING)]
            host, ev = self._ring[self._ring_next]
            self._ring_next = (self._ring_next + 1) % self.RING
            host.copy_(self._compact(), non_blocking=True)
            ev.record()
            self.amax.zero_()
            self.pending.append((self.step, host, ev))
        while len(self.pending) > 1:         # everything but the read-back just queued: steps <= t - 2, in order (waits if it must)
            self._harvest_one_blocking()
        self.step += 1
        self.sites = {}
        self.fp16_now = bool(self.enabled and self.slots and all(fs <= self.measured for fs in self.first_step))
        if self.fp16_now:
            self.steps_fp16 += 1

    def _compact(self):
        """[MAX_SITES] = the maximum over each site's sub-slots (one small reduction on the device)."""
        return self.amax.view(self.MAX_SITES, _capi.LPM_OPERAND_AMAX_SUB, _capi.LPM_OPERAND_AMAX_STRIDE)[:, :, 0].amax(dim=1)

    def _harvest_one_blocking(self):
        sid, host, ev = self.pending.pop(0)
        ev.synchronize()
        vals = host.tolist()
        self._check_finite(vals, sid)
        self.hist[0], self.hist[1] = self.hist[1], vals
        self.measured = sid

    def calibrate_from_device(self):
        """Synchronous: the maxima recorded since the last begin_step become the measurement of the running step (tests / the first
        step of a run: Trainer.calibrate_operand_scales)."""
        torch.cuda.synchronize(self.device)
        self._harvest(wait=True)
        vals = self._compact().tolist()
        self._check_finite(vals, self.step)
        self.hist[0], self.hist[1] = vals, vals
        self.measured = self.step
        self.amax.zero_()

    def _scale_of(self, slot):
        import math
        a = max(self.hist[0][slot], self.hist[1][slot])
        if not (a > 0.0) or math.isinf(a) or math.isnan(a):
            return 1.0
        return 2.0 ** (self.TARGET - math.floor(math.log2(a)))

    def site(self, key):
        st = self.sites.get(key)
        if st is not None:
            return st
        slot = self.slots.get(key)
        if slot is None:
            slot = len(self.slots)
            if slot >= self.MAX_SITES:
                raise LpmError("OperandScales: more operand sites than slots")
            self.slots[key] = slot
            self.first_step.append(self.step)
        # (a site that appears for the first time INSIDE an fp16 step -- a code path the earlier steps did not take -- runs that step in
        # fp16 with scale 1: a step never mixes formats, and the site is measured from here on)
        f16 = self.fp16_now
        st = OperandSite(f16, self._scale_of(slot) if (f16 and self.first_step[slot] <= self.measured) else 1.0,
                         self.amax.data_ptr() + 4 * self.SLOT * slot, role=key[0])
        self.sites[key] = st
        return st

    def report(self):
        """{key: (newest measured max |x|, the scale it gives)} -- diagnostics / DESIGN's table."""
        return {k: (self.hist[1][sl], self._scale_of(sl)) for k, sl in self.slots.items()}


_ACTIVE_SCALES = None    # the OperandScales of the trainer whose step is running (train.Trainer.step; NetVladV1 with FLAGS.dense_arithmetic = "fp16x2"), else None


def _site(role, W):
    """The operand site (role "a": the layer's input activation, "g": the gradient of its output) of the dense layer with kernel W in the
    running step, or None outside a trainer's step / for a model that stays on split-bf16 x3."""
    sc = _ACTIVE_SCALES
    if sc is None or W is None:
        return None
    return sc.site((role, W.data_ptr()))


def _f16(site):
    return site is not None and site.f16


def _split_rows(x2d, bias=None, relu=False, grad=False, row_scale=None, site=None):
    """[M,K] fp32 -> [M,3K] bf16 = [hi | lo | hi] (optionally of relu(x + bias)); grad=True: the gradient plane order
    [hi | hi | lo] that pairs with w3k (rows [Wh|Wl|Wh]) and, row by row, with an activation image (see _dw_x3).
    site (OperandSite): the image in the site's format -- fp16: planes of x * site.scale, [hi | lo | hi] for an activation site,
    [hi | lo] for a gradient site -- and max |x| recorded.
    row_scale: x2d = the rows of a lazily normalised descriptor (netvlad(lazy=True)), scaled as they are read."""
    lib = _capi.load()
    M, K = x2d.shape
    f16 = _f16(site)
    out = torch.empty((M, (site.planes if site is not None else 3) * K), dtype=torch.float16 if f16 else torch.bfloat16, device=x2d.device)
    lib.check(lib._lpm_split_rows_fmt(ptr(x2d), x2d.stride(0), M, K, ptr(bias), 1 if relu else 0, 1 if grad else 0, ptr(row_scale), ptr(out),
                                      site.fmt if site is not None else None, stream_ptr()), "lpm_split_rows")
    return out


class WeightPack:
    """Every operand form of every dense-layer weight of a training step from ONE launch (lpm_weight_pack, weight_pack.hip).

    A weight changes once per step, in the optimiser; the forward and backward consume it as split-bf16 images (w3n, w3k) and fragment
    tiles (wt, wtt) that round 3 derived where they were used -- 14 small launches per cfg-2 step.  The trainer owns one WeightPack:
    ``begin_step`` (weights final, before the forward) derives all forms recorded so far in one launch; the consumers ask ``take``
    for them, which RECORDS what it is asked for -- a form that is not ready (the first step, a model that changed) is computed by the
    consumer's own call, as without the pack, and is part of the launch from the next step on.  Outside a trainer's step (predict(),
    the kernel tests) nothing is armed and every consumer computes its own."""

    def __init__(self):
        self.plan = {}          # key (address and shape of the source weights) -> {"srcs": [tensors], "need": set of forms}
        self.ready = {}         # key -> {form: tensor}, valid for the current step
        self.armed = False

    @staticmethod
    def _ok(w):
        return (torch.is_tensor(w) and w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous()
                and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0 and w.data_ptr() % 16 == 0)

    def begin_step(self):
        self.ready = {}
        self.armed = True
        if not self.plan or not WEIGHT_PACK:
            return
        lib = _capi.load()
        jobs, keep = [], []
        fp16_now = _ACTIVE_SCALES is not None and _ACTIVE_SCALES.fp16_now
        for key, e in self.plan.items():
            srcs, need = e["srcs"], e["need"]
            if not need or not all(self._ok(w) for w in srcs) or len({w.shape[0] for w in srcs}) != 1:
                continue
            # a weight asked for in both operand formats (the fp16 two-product steps of NetVladV1 follow a few split-bf16 ones): only the
            # forms of the format this step runs in
            need16 = {f for f in need if f.endswith("16")}
            f16 = bool(fp16_now and need16)
            need = need16 if f16 else (need - need16)
            if not need:
                continue
            sfx = "16" if f16 else ""
            # fp16 forms: n16 [Ntot, 3K] = [Wh^T|Wh^T|Wl^T], k16 [K, 2 Ntot] = [Wh|Wh], wt16 (hi, lo) tiles, wtt16 hi-plane tiles (half the bytes)
            pk, dt, tdiv = (2, torch.float16, 2) if f16 else (3, torch.bfloat16, 1)
            K = srcs[0].shape[0]
            Ntot = sum(w.shape[1] for w in srcs)
            dev = srcs[0].device
            out = {}
            if "n" + sfx in need:
                out["n" + sfx] = torch.empty((Ntot, 3 * K), dtype=dt, device=dev)
            if "k" + sfx in need:
                out["k" + sfx] = torch.empty((K, pk * Ntot), dtype=dt, device=dev)
            if "wt" + sfx in need:
                out["wt" + sfx] = torch.empty(lib._lpm_weight_tiles_bytes(K, Ntot) // 4, dtype=torch.int32, device=dev)
            if "wtt" + sfx in need and len(srcs) == 1:
                out["wtt" + sfx] = torch.empty(lib._lpm_weight_tiles_bytes(Ntot, K) // 4 // tdiv, dtype=torch.int32, device=dev)
            off = 0
            for w in srcs:
                j = _capi.WeightPackJob()
                j.w, j.K, j.N, j.ldw, j.Ntot, j.n_off = w.data_ptr(), K, w.shape[1], w.stride(0), Ntot, off
                j.w3n = out["n" + sfx].data_ptr() if "n" + sfx in out else None
                j.w3k = out["k" + sfx].data_ptr() if "k" + sfx in out else None
                j.wt = out["wt" + sfx].data_ptr() if "wt" + sfx in out else None
                j.wtt = out["wtt" + sfx].data_ptr() if "wtt" + sfx in out else None
                j.kind = _capi.LPM_OPERAND_FP16X3 if f16 else _capi.LPM_OPERAND_BF16X3
                jobs.append(j)
                off += w.shape[1]
            self.ready[key] = out
        st = stream_ptr()
        for i in range(0, len(jobs), _capi.WEIGHT_PACK_MAX_JOBS):
            chunk = jobs[i:i + _capi.WEIGHT_PACK_MAX_JOBS]
            arr = (_capi.WeightPackJob * len(chunk))(*chunk)
            lib.check(lib._lpm_weight_pack(C.cast(arr, C.c_void_p), len(chunk), st), "lpm_weight_pack")

    def end_step(self):
        self.armed = False
        self.ready = {}

    def take(self, srcs, forms):
        """The forms of the weight (or of the concatenation of the weights) ``srcs`` as a dict, or None -- then the caller computes them
        itself; either way the request is on record for the next ``begin_step``."""
        if not self.armed:
            return None
        if not all(self._ok(w) for w in srcs):
            return None
        # keyed by storage, not by object: autograd hands a saved weight back as a fresh tensor object over the same memory
        key = tuple((w.data_ptr(), w.shape[0], w.shape[1]) for w in srcs)
        e = self.plan.get(key)
        if e is None:
            e = self.plan[key] = {"srcs": [w.detach() for w in srcs], "need": set()}
        e["need"].update(forms)
        got = self.ready.get(key)
        if got is None or any(f not in got for f in forms):
            return None
        return {f: got[f] for f in forms}


# the split-K partial sums of the encoders' weight-gradient GEMMs added into their arena slots by lpm_sum_splits; "0": torch.sum per slot (A/B)
SUM_SPLITS = os.environ.get("LPM_SUM_SPLITS", "1") != "0"
_ACTIVE_PACK = None      # the WeightPack of the trainer whose step is running (train.Trainer.step), else None
# "0": every consumer derives its own operand forms, one launch each (A/B)
WEIGHT_PACK = os.environ.get("LPM_WEIGHT_PACK", "1") != "0"


_LIBRARY_SELECTION = None       # None: not attempted; True / False: recorded solutions active / not (mismatching validators, switched off, ...)
LIBRARY_GEMM_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_tunable", "gfx950_fp32_gemm.csv")


def enable_library_gemm_selection():
    """FLAGS.library_gemm_selection: hand PyTorch the per-shape hipBLASLt / rocBLAS solutions recorded for the fp32 GEMMs it runs on this path
    (TunableOp, tuning off -- nothing is searched or timed at run time).  A process that already configured TunableOp itself (the
    PYTORCH_TUNABLEOP_* environment, e.g. tools/tune_library_gemms.py regenerating the file) is left alone.  Once per process."""
    global _LIBRARY_SELECTION
    if _LIBRARY_SELECTION is not None:
        return _LIBRARY_SELECTION
    _LIBRARY_SELECTION = False
    from . import FLAGS
    if (not FLAGS.library_gemm_selection or os.environ.get("LPM_LIBRARY_GEMM_SELECTION") == "0" or not torch.cuda.is_available()
            or "PYTORCH_TUNABLEOP_ENABLED" in os.environ or not os.path.exists(LIBRARY_GEMM_FILE)):
        return False
    import torch.cuda.tunable as tunable
    try:
        tunable.enable(True)
        tunable.tuning_enable(False)
        tunable.record_untuned_enable(False)
        _LIBRARY_SELECTION = bool(tunable.read_file(LIBRARY_GEMM_FILE))
        if not _LIBRARY_SELECTION:
            tunable.enable(False)
    except Exception as e:      # an older / differently built PyTorch: the library's default choices are what every round before 4 ran on
        import warnings
        warnings.warn(f"library GEMM selection not available ({e}); continuing on the library's default solutions")
        _LIBRARY_SELECTION = False
    return _LIBRARY_SELECTION


def _packed(srcs, forms):
    return _ACTIVE_PACK.take(srcs, forms) if _ACTIVE_PACK is not None else None


def _weight_tiles(W, R, N, transposed, like, pack=True, f16=False):
    """lpm_split_weight_tiles(W, R, N, transposed) -- from the step's weight pack when it holds the form.  f16: fp16 tiles -- of W (the
    forward's operand) with (hi, lo) planes, of W^T (``transposed``: the input gradient's operand, a two-term product) the hi plane only:
    half the bytes."""
    lib = _capi.load()
    form = ("wtt" if transposed else "wt") + ("16" if f16 else "")
    got = _packed([W], [form]) if pack else None
    if got is not None:
        return got[form]
    kind = (_capi.LPM_OPERAND_FP16X2 if transposed else _capi.LPM_OPERAND_FP16X3) if f16 else _capi.LPM_OPERAND_BF16X3
    wt = _tile_buffer(lib._lpm_weight_tiles_bytes(R, N) // (2 if (f16 and transposed) else 1), like)
    lib.check(lib._lpm_split_weight_tiles_fmt(ptr(W), R, N, 1 if transposed else 0, ptr(wt), kind, stream_ptr()), "lpm_split_weight_tiles")
    return wt


def _split_weight_cat(Ws, need_t=True, f16=False):
    """_split_weight of the concatenation of Ws along the columns (q | k | v) -- without forming it when the weight pack has the images."""
    fn, fk = ("n16", "k16") if f16 else ("n", "k")
    got = _packed(list(Ws), [fn, fk] if need_t else [fn])
    if got is not None:
        return got[fn], got.get(fk)
    return _split_weight(torch.cat([_f32(w, "kernel") for w in Ws], dim=1), need_t, pack=False, f16=f16)


def _split_weight(W, need_t=True, pack=True, f16=False):
    """[K,N] fp32 -> w3n [N,3K] (rows [Wh^T|Wh^T|Wl^T]: y = X3 w3n^T) and w3k [K,3N] (rows [Wh|Wl|Wh]: dx = DY3 w3k^T), bf16.
    f16: fp16 planes -- wn [N,3K] = [Wh^T|Wh^T|Wl^T] (the forward keeps three terms), wk [K,2N] = [Wh|Wh] (the input gradient's two-term
    product: the weight rounded once)."""
    lib = _capi.load()
    fn, fk = ("n16", "k16") if f16 else ("n", "k")
    if pack:
        got = _packed([W], [fn, fk] if need_t else [fn])
        if got is not None:
            return got[fn], got.get(fk)
    K, N = W.shape
    pk, dt = (2, torch.float16) if f16 else (3, torch.bfloat16)
    w3n = torch.empty((N, 3 * K), dtype=dt, device=W.device)
    w3k = torch.empty((K, pk * N), dtype=dt, device=W.device) if need_t else None
    lib.check(lib._lpm_split_weight_fmt(ptr(W), K, N, ptr(w3n), ptr(w3k), _capi.LPM_OPERAND_FP16X3 if f16 else _capi.LPM_OPERAND_BF16X3,
                                        stream_ptr()), "lpm_split_weight")
    return w3n, w3k


def _mm3(a3, w3, acc=None, alpha=1.0):
    """a3 [M,3K] . w3 [N,3K]^T with fp32 accumulation and output (the weight image is stored transposed).  acc: an fp32
    [M,N] tensor the product is added to IN PLACE (the GEMM's beta = 1 instead of a separate add pass).  alpha (the fp16 two-product
    format: 1 / the data operand's scale, a power of two) rides in the GEMM."""
    if alpha == 1.0:
        if acc is None:
            return torch.mm(a3, w3.t(), out_dtype=torch.float32)
        return torch.addmm(acc, a3, w3.t(), out_dtype=torch.float32, out=acc)
    if acc is None:
        out = torch.empty((a3.shape[0], w3.shape[0]), dtype=torch.float32, device=a3.device)
        return torch.addmm(out, a3, w3.t(), out_dtype=torch.float32, beta=0, alpha=alpha, out=out)
    return torch.addmm(acc, a3, w3.t(), out_dtype=torch.float32, alpha=alpha, out=acc)


class _DenseX3(torch.autograd.Function):
    """y = x W with x W ~= [xh|xl|xh].[Wh;Wh;Wl]: library bf16 GEMMs (hipBLASLt), fp32 accumulation, ~4e-6 relative
    error (a plain bf16 GEMM: 2.5e-3, outside the parity bar; the fp32 GEMM: 1.5e-6 at 2-3x the time)."""

    @staticmethod
    def forward(ctx, x2d, W, x3=None):
        """x3 (block Functions only): the input already as its activation image (then x2d is None)."""
        W0 = W
        W = _f32(W, "dense kernel").contiguous()
        sa = ctx.site_a = _site("a", W0)          # (an image handed in was written in this site's format by its producer)
        if x3 is None:
            x2d = _rows(x2d, "dense input")
            x3 = _split_rows(x2d, site=sa)
        w3n, w3k = _split_weight(W, need_t=ctx.needs_input_grad[0], f16=_f16(sa))
        ctx.save_for_backward(x3, w3k)
        ctx.dims = (W.shape[0], W.shape[1])
        ctx.wrefs = (W0,)
        return _mm3(x3, w3n, alpha=sa.inv if sa is not None else 1.0)

    @staticmethod
    def backward(ctx, dy, dy3=None):
        """dy3 (block Functions only): dy already as its gradient image (in the format of this layer's "g" site)."""
        x3, w3k = ctx.saved_tensors
        K, N = ctx.dims
        sa, sg = ctx.site_a, _site("g", ctx.wrefs[0])
        if dy3 is None:
            dy3 = _split_rows(dy.contiguous(), grad=True, site=sg)
        dx = _mm3(dy3, w3k, alpha=sg.inv if sg is not None else 1.0) if ctx.needs_input_grad[0] else None
        dW = _dw_x3(x3, dy3, K, N, outs=[(ctx.wrefs[0], 0, N)], sa=sa, sg=sg)[0] if ctx.needs_input_grad[1] else None
        return dx, dW


def dense_x3(x2d, W):
    return _DenseX3.apply(x2d, W)


class _QKVX3(torch.autograd.Function):
    """q, k, v = x Wq, x Wk, x Wv of a self-attention block as ONE split-bf16 library GEMM against [Wq|Wk|Wv]: the input is
    split once, the GEMM has 3x the columns (1024-column GEMMs fill the chip badly: 0.70 vs 1.2 PFLOP/s executed), and in
    the backward dx comes out of a single GEMM over the concatenated reduction instead of three GEMMs + two adds.
    q, k, v are returned as column views of one [M, 3N] buffer (the attention kernels take a row stride)."""

    @staticmethod
    def forward(ctx, x2d, Wq, Wk, Wv, row_scale=None):
        x2d = _rows(x2d, "dense input")
        K, N = Wq.shape
        sa = ctx.site_a = _site("a", Wq)
        x3 = _split_rows(x2d, row_scale=row_scale, site=sa)
        w3n, w3k = _split_weight_cat([Wq, Wk, Wv], need_t=ctx.needs_input_grad[0], f16=_f16(sa))
        ctx.save_for_backward(x3, w3k)
        ctx.dims = (K, N)
        ctx.wrefs = (Wq, Wk, Wv)
        qkv = _mm3(x3, w3n, alpha=sa.inv if sa is not None else 1.0)
        return qkv[:, :N], qkv[:, N:2 * N], qkv[:, 2 * N:]

    @staticmethod
    def backward(ctx, dq, dk, dv, acc=None, dy3=None):
        """acc (block Functions only): dx is accumulated into it in place.  dy3: the gradient image of [dq | dk | dv] when
        the attention backward produced it directly (then dq, dk, dv are None)."""
        x3, w3k = ctx.saved_tensors
        K, N = ctx.dims
        M = x3.shape[0]
        sa, sg = ctx.site_a, _site("g", ctx.wrefs[0])
        if dy3 is None:
            esz = dq.element_size()
            adjacent = (dq.stride() == (3 * N, 1) and dk.stride() == (3 * N, 1) and dv.stride() == (3 * N, 1)
                        and dk.data_ptr() == dq.data_ptr() + N * esz and dv.data_ptr() == dq.data_ptr() + 2 * N * esz)
            if adjacent:      # the attention backward wrote all three into one buffer
                dqkv = torch.as_strided(dq, (M, 3 * N), (3 * N, 1))
            else:
                dqkv = torch.cat([dq, dk, dv], dim=1)
            dy3 = _split_rows(dqkv, grad=True, site=sg)
        dx = _mm3(dy3, w3k, acc, alpha=sg.inv if sg is not None else 1.0) if ctx.needs_input_grad[0] else None
        Wq, Wk, Wv = ctx.wrefs
        dWq, dWk, dWv = _dw_x3(x3, dy3, K, 3 * N, outs=[(Wq, 0, N), (Wk, N, N), (Wv, 2 * N, N)], sa=sa, sg=sg)
        return dx, dWq, dWk, dWv


def qkv_x3(x2d, Wq, Wk, Wv):
    return _QKVX3.apply(x2d, Wq, Wk, Wv)


def _grad_slot(W):
    """The trainer's arena slice for this weight's gradient (ParameterArena.mark_direct) if it may be written directly this step --
    it exists and nothing has been written to it yet -- else None: the caller returns the gradient to autograd as usual (a second use of
    the weight in one step accumulates there and ParameterArena.collect folds it in)."""
    view = getattr(W, "_lpm_grad_view", None) if (W is not None and DIRECT_WGRAD) else None
    if view is None or getattr(W, "_lpm_grad_written", False):
        return None
    return view


def _grad_done(W):
    W._lpm_grad_written = True
    ready = getattr(W, "_lpm_grad_ready", None)
    if ready is not None:
        ready()


class _LinearDirect(torch.autograd.Function):
    """y = x W (+ b) as plain fp32 library GEMMs (the MoE head's slim.fully_connected layers, video_level_models.py:86-114) whose weight
    gradient x^T dy goes straight into the trainer's arena slot when the weight has a free one (_grad_slot): for the two MoE matrices
    (10 M parameters at cfg-2) that is one GEMM writing in place instead of a GEMM + a 40 MB gather copy."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.wref = W
        ctx.has_bias = b is not None
        return torch.addmm(b, x, W) if b is not None else x.matmul(W)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dx = dy.matmul(W.t()) if ctx.needs_input_grad[0] else None
        dW = None
        if ctx.needs_input_grad[1]:
            slot = _grad_slot(ctx.wref)
            if slot is not None:
                torch.mm(x.t(), dy, out=slot)
                _grad_done(ctx.wref)
            else:
                dW = x.t().matmul(dy)
        db = dy.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dW, db


def linear_direct(x, W, b=None):
    return _LinearDirect.apply(x, W, b)


def _dw_x3(x3, dy3, K, N, outs=None, sa=None, sg=None, terms=None):
    """outs: [(weight, first column, columns)] -- the gradient's column blocks belong to these weights; a weight with a free arena slot
    (_grad_slot) receives its block straight from the split-K sum (no fresh tensor, no AccumulateGrad copy, no gather copy) and the
    returned tuple holds None in its place.  Without outs: the [K, N] gradient.  sa / sg: the operand sites of the two images (fp16
    planes: _dw_x2; terms: this site's number of products there, default DW_TERMS)."""
    if _f16(sa) != _f16(sg):
        raise LpmError("weight gradient: the activation image and the gradient image are in different operand formats")
    if _f16(sa):
        return _dw_x2(x3, dy3, K, N, outs, sa.inv * sg.inv, DW_TERMS if terms is None else terms)
    res = _dw_x3_impl(x3, dy3, K, N, outs)
    return res


# "1": FeedForwardNetwork's tile GEMMs read their data operand straight from its operand image (per-lane LDS-DMA gather of 32-byte row
# pieces) instead of from a row-tile copy made by a pass of its own (lpm_split_rows_tiles / lpm_image_row_tiles: 96 + 38 us per cfg-2
# step).  Built and measured in round 5, one box, alternating: 6.57 / 6.50 ms with the gather against 6.46 ms with the copies -- the
# gathered loads cost the two GEMMs what the passes cost; off by default, kept for the A/B
TILE_GEMM_FROM_IMAGE = os.environ.get("LPM_TILE_GEMM_FROM_IMAGE", "0") == "1"
# terms of the fp16 weight-gradient product: 2 = xh^T [dyh | dyl] (the gradient exact, the activation rounded once: 1.4e-4 per GEMM);
# 1 = xh^T dyh (both rounded once: 2e-4 per GEMM -- an error that stays in THIS weight's gradient and is not carried further down the
# backward, unlike an input gradient's).  LPM_DW_TERMS, A/B; measured in tests/test_gpu_fp16x2.py
DW_TERMS = int(os.environ.get("LPM_DW_TERMS", "1"))
# ... of ONE site, FeedForwardNetwork's first kernel (transformer_utils.py:701-704) -- the variable whose gradient sits closest to the
# north-star's 1e-3 at the untouched initialisation (1.0e-3 with two terms everywhere in round 4, 1.15e-3 with one).
# Measured in round 6 (tests/test_gpu_models.py::test_untouched_reference_initialisation[cfg2] prints it per setting)
DW_TERMS_FFN1 = int(os.environ.get("LPM_DW_TERMS_FFN1", str(DW_TERMS)))


DW_SLICES = int(os.environ.get("LPM_DW_SLICES", "0"))        # 0: the policy below; n: that many slices of the token reduction (A/B)


def _dw_x2(x2, dy2, K, N, outs, alpha, terms=None):
    """The fp16 weight gradient from an activation image x2 [M,3K] = [hi|lo|hi] (fp16) and a gradient image dy2 [M,2N] = [hi|lo].
    terms = 1 (the default, DW_TERMS): dW = alpha * xh^T dyh -- BOTH operands rounded once to fp16 (their hi planes, read in place with
    the images' row strides): 2.9e-4 relative L2 per GEMM, an error that stays in this weight's gradient.  terms = 2: dW = alpha * xh^T
    [dyh | dyl] -- the activation rounded once, the gradient exact to 22 bits (1.4e-4), twice the matrix-pipe work.  Either way ONE fp16
    library GEMM over S slices of the token reduction with a [K, terms N] output per slice, then lpm_sum_splits_scaled adds slices
    (and halves) and multiplies by alpha = 1 / (the two operands' scales)."""
    terms = DW_TERMS if terms is None else terms
    lib = _capi.load()
    M = x2.shape[0]
    S = DW_SLICES if DW_SLICES else (8 if K * N <= (1 << 20) else 4)     # (in-step at cfg-2, 2 -> 4 slices: 171 / 174 / 155 -> 146 / 151 / 116 us)
    while S > 1 and (M % S or M // S < 512):
        S //= 2
    xh = x2.view(S, M // S, x2.shape[1])[:, :, :K]
    halves = 2 if terms == 2 else 1
    dyv = dy2.view(S, M // S, 2 * N)
    part = torch.bmm(xh.transpose(1, 2), dyv if halves == 2 else dyv[:, :, :N], out_dtype=torch.float32)       # [S, K, halves N]
    slots = [_grad_slot(W) for W, _, _ in outs] if outs is not None else []
    if (outs is not None and all(sl is not None and sl.is_contiguous() for sl in slots) and len(outs) <= 3
            and all(nc == N // len(outs) and c0 == i * (N // len(outs)) for i, (_, c0, nc) in enumerate(outs)) and (N // len(outs)) % 4 == 0):
        sp = [ptr(sl) for sl in slots] + [None] * (3 - len(slots))
        lib.check(lib._lpm_sum_splits_scaled(ptr(part), S, K, N, halves, alpha, sp[0], sp[1], sp[2], len(slots), stream_ptr()), "lpm_sum_splits_scaled")
        for W, _, _ in outs:
            _grad_done(W)
        return tuple(None for _ in outs)
    full = torch.empty((K, N), dtype=torch.float32, device=x2.device)
    lib.check(lib._lpm_sum_splits_scaled(ptr(part), S, K, N, halves, alpha, ptr(full), None, None, 1, stream_ptr()), "lpm_sum_splits_scaled")
    if outs is None:
        return full
    res = []
    for (W, c0, nc), sl in zip(outs, slots):
        blk = full[:, c0:c0 + nc] if (c0, nc) != (0, N) else full
        if sl is None:
            res.append(blk)
        else:
            sl.copy_(blk)
            _grad_done(W)
            res.append(None)
    return tuple(res)


def _dw_x3_impl(x3, dy3, K, N, outs):
    """dW = x^T dy from an activation image x3 [M,3K] = [hi|lo|hi] and a gradient image dy3 [M,3N] = [hi|hi|lo]: seen as
    [3M,K] and [3M,N] their rows pair up plane by plane, so dW = xh^T dyh + xl^T dyh + xh^T dyl is one bf16 GEMM with a
    3M-deep reduction and fp32 accumulation.  hipBLASLt does not split a long reduction with a small output by itself
    (one 61440-deep GEMM: 0.45-0.94 PFLOP/s executed at cfg-2), so it is handed over as a batched GEMM over S slices of the
    reduction plus a sum (0.93-1.06 PFLOP/s, tools/bench_dw_gemms.py; the dW^T form is ~4 % faster still but hands autograd
    a transposed gradient that it then copies)."""
    M3 = 3 * x3.shape[0]
    S = 8 if K * N <= (1 << 20) else 4
    if M3 % S or M3 // S < 512:
        full = torch.mm(x3.view(M3, K).t(), dy3.view(M3, N), out_dtype=torch.float32)
        part = None
    else:
        xb, db = x3.view(S, M3 // S, K), dy3.view(S, M3 // S, N)
        part = torch.bmm(xb.transpose(1, 2), db, out_dtype=torch.float32)           # [S, K, N] partial sums
        full = None
    if outs is None:
        return full if full is not None else part.sum(0)
    slots = [_grad_slot(W) for W, _, _ in outs]
    if (SUM_SPLITS and part is not None and all(sl is not None and sl.is_contiguous() for sl in slots) and len(outs) <= 3
            and all(nc == N // len(outs) and c0 == i * (N // len(outs)) for i, (_, c0, nc) in enumerate(outs)) and (N // len(outs)) % 4 == 0):
        # every destination is a free arena slot and the column blocks tile the product evenly: ONE launch adds the split-K partial sums
        # into all of them (lpm_sum_splits) instead of one strided torch reduction per destination
        lib = _capi.load()
        sp = [ptr(sl) for sl in slots] + [None] * (3 - len(slots))
        lib.check(lib._lpm_sum_splits(ptr(part), part.shape[0], K, N, sp[0], sp[1], sp[2], len(slots), stream_ptr()), "lpm_sum_splits")
        for W, _, _ in outs:
            _grad_done(W)
        return tuple(None for _ in outs)
    if all(sl is None for sl in slots) and full is None:
        full = part.sum(0)
    res = []
    for (W, c0, nc), sl in zip(outs, slots):
        if sl is None:
            res.append(full[:, c0:c0 + nc] if (c0, nc) != (0, N) else full)
        else:
            if full is not None:
                sl.copy_(full[:, c0:c0 + nc])
            else:
                torch.sum(part[:, :, c0:c0 + nc], 0, out=sl)
            _grad_done(W)
            res.append(None)
    if full is None and any(sl is None for sl in slots):                              # mixed (never in practice): the rest from one sum
        full = part.sum(0)
        res = [full[:, c0:c0 + nc] if r is None and sl is None else r for r, sl, (W, c0, nc) in zip(res, slots, outs)]
    return tuple(res)


class _FFNX3(torch.autograd.Function):
    """relu(y W1 + b1) W2 of FeedForwardNetwork (transformer_utils.py:701-708) on split-bf16 operands with the first
    layer's bias + ReLU fused into the operand split of the second: the [M,4F] activation exists only as its bf16
    hi/lo image, and the backward's ReLU mask, bias gradient and operand split are one pass as well."""

    @staticmethod
    def forward(ctx, y2d, W1, b1, W2, y3=None):
        """y3: the [M, 3F] activation image of y2d when its producer already wrote it (the attention block's layer norm)."""
        y2d = _rows(y2d, "ffn input")
        W1_0, W2_0 = W1, W2
        W1, W2 = _f32(W1, "W1").contiguous(), _f32(W2, "W2").contiguous()
        lib = _capi.load()
        M, F = y2d.shape
        H = W1.shape[1]
        # operand sites (fp16 two-product format when the trainer's scales are calibrated): y feeds layer 1, the hidden activation layer 2
        s1, s2 = _site("a", W1_0), _site("a", W2_0)
        f16 = _f16(s1)
        ctx.sites = (s1, s2)
        dt = torch.float16 if f16 else torch.bfloat16
        a1 = s1.inv if s1 is not None else 1.0
        if y3 is None or tuple(y3.shape) != (M, 3 * F) or y3.dtype != dt:
            y3 = _split_rows(y2d, site=s1)
        w13n, w13k = _split_weight(W1, f16=f16)
        ctx.tiles = bool(FFN_TILES and y2d.stride(1) == 1 and lib._lpm_dense_tiles_supported(M, F, H) and lib._lpm_dense_tiles_supported(M, W2.shape[1], H))
        if ctx.tiles:
            # the first dense layer on the hand-written 256-row tile GEMM with the bias + ReLU + operand split in its epilogue: the
            # [M, 4F] pre-activation never exists in fp32 (335 MB written + read at cfg-2), no separate split pass
            st = stream_ptr()
            akind = _capi.LPM_OPERAND_FP16X3 if f16 else _capi.LPM_OPERAND_BF16X3
            if TILE_GEMM_FROM_IMAGE:     # the kernel gathers its A fragments from the image y3 row by row: no row-tile copy of y
                yr, ykind = y3, akind
            else:
                yr, ykind = _tile_buffer(lib._lpm_row_tiles_bytes(1, M, F), y2d), -1
                lib.check(lib._lpm_split_rows_tiles_fmt(ptr(y2d), y2d.stride(0), 1, M, F, ptr(yr), s1.fmt if s1 is not None else None, st),
                          "lpm_split_rows_tiles")
            w1t = _weight_tiles(W1, F, H, False, y2d, f16=f16)
            f3 = torch.empty((M, 3 * H), dtype=dt, device=y2d.device)
            lib.check(lib._lpm_dense_tiles_act_image_fwd_fmt(ptr(yr), ykind, ptr(w1t), ptr(b1.contiguous()), M, F, H, a1, ptr(f3),
                                                             s2.fmt if s2 is not None else None, st), "lpm_dense_tiles_act_image_fwd")
            w23n, _ = _split_weight(W2, need_t=False, f16=f16)
            ctx.save_for_backward(y3, f3, w13k, W2)
        else:
            pre1 = _mm3(y3, w13n, alpha=a1)
            f3 = _split_rows(pre1, bias=b1.contiguous(), relu=True, site=s2)
            del pre1
            w23n, w23k = _split_weight(W2, f16=f16)
            ctx.save_for_backward(y3, f3, w13k, w23k)
        ctx.dims = (W1.shape[0], W1.shape[1], W2.shape[1])
        ctx.wrefs = (W1_0, W2_0)
        return _mm3(f3, w23n, alpha=s2.inv if s2 is not None else 1.0)

    @staticmethod
    def backward(ctx, dout, acc=None, do3=None):
        """Block Functions only -- acc: dy is accumulated into it in place; do3: dout already as its gradient image."""
        lib = _capi.load()
        y3, f3, w13k, w23k = ctx.saved_tensors
        F, H, N = ctx.dims
        M = y3.shape[0]
        s1, s2 = ctx.sites
        g1, g2 = _site("g", ctx.wrefs[0]), _site("g", ctx.wrefs[1])       # the gradients of the two layers' outputs (do3 arrives in g2's format)
        f16 = _f16(s1)
        gkind = _capi.LPM_OPERAND_FP16X2 if f16 else _capi.LPM_OPERAND_BF16X3       # gradient images: [hi | lo] in fp16
        akind = _capi.LPM_OPERAND_FP16X3 if f16 else _capi.LPM_OPERAND_BF16X3       # the activation image f3 (ReLU mask): [hi | lo | hi]
        pl, dt = (2, torch.float16) if f16 else (3, torch.bfloat16)
        a2 = g2.inv if g2 is not None else 1.0
        if do3 is None:
            do3 = _split_rows(dout.contiguous(), grad=True, site=g2)
        dW2 = _dw_x3(f3, do3, H, N, outs=[(ctx.wrefs[1], 0, N)], sa=s2, sg=g2)[0]
        dp3 = torch.empty((M, pl * H), dtype=dt, device=f3.device)
        db1 = torch.empty((H,), dtype=torch.float32, device=f3.device)
        if ctx.tiles:
            # df = do W2^T on the tile GEMM; ReLU mask (the activation image's hi plane), bias gradient partial sums and the operand
            # split of the result in its epilogue: df never exists in fp32
            st = stream_ptr()
            W2 = w23k                                                     # (saved in its place: the fp32 weight [H, N])
            if TILE_GEMM_FROM_IMAGE:
                dor, dkind = do3, gkind
            else:
                dor, dkind = _tile_buffer(lib._lpm_row_tiles_bytes(1, M, N), f3), -1
                lib.check(lib._lpm_image_row_tiles_fmt(ptr(do3), M, N, 1, ptr(dor), gkind, st), "lpm_image_row_tiles")
            w2tt = _weight_tiles(W2, N, H, True, f3, f16=f16)
            wsb = lib._lpm_dense_tiles_relu_bwd_workspace_bytes(M, H)
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=f3.device)
            lib.check(lib._lpm_dense_tiles_relu_bwd_image_fmt(ptr(dor), dkind, ptr(w2tt), ptr(f3), akind, M, N, H, a2, ptr(dp3), ptr(db1), ptr(ws), wsb,
                                                              g1.fmt if g1 is not None else None, st), "lpm_dense_tiles_relu_bwd_image")
        else:
            df = _mm3(do3, w23k)                                          # [M, H]  (un-scaled: alpha rides in the split pass below)
            wsb = lib._lpm_split_rows_relu_bwd_workspace_bytes(M, H)
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=df.device)
            lib.check(lib._lpm_split_rows_relu_bwd_fmt(ptr(df), M, H, a2, ptr(f3), akind, ptr(dp3), ptr(db1), ptr(ws), wsb,
                                                       g1.fmt if g1 is not None else None, stream_ptr()), "lpm_split_rows_relu_bwd")
            del df
        dy = _mm3(dp3, w13k, acc, alpha=g1.inv if g1 is not None else 1.0)
        dW1 = _dw_x3(y3, dp3, F, H, outs=[(ctx.wrefs[0], 0, H)], sa=s1, sg=g1, terms=DW_TERMS_FFN1)[0]
        return dy, dW1, db1, dW2


def ffn_x3(y2d, W1, b1, W2):
    return _FFNX3.apply(y2d, W1, b1, W2)


# ----------------------------------------------------------------------------------------------
# a9: VLAD -> hidden projection (frame_level_models.py:2314-2319): [B, 270336] x [270336, H], weight-stream bound
# ----------------------------------------------------------------------------------------------
class _BiasAct(torch.autograd.Function):
    """act(y + bias) of tf.layers.dense IN PLACE on the GEMM's fresh output (one pass instead of add + relu); backward: the ReLU mask
    from the saved output and the bias gradient's column sums in one pass (instead of threshold + reduce)."""

    @staticmethod
    def forward(ctx, y, bias, relu):
        lib = _capi.load()
        C = y.shape[-1]
        M = y.numel() // C
        lib.check(lib._lpm_bias_act_fwd(ptr(y), ptr(bias.contiguous()), 1 if relu else 0, M, C, stream_ptr()), "lpm_bias_act_fwd")
        ctx.mark_dirty(y)
        ctx.relu = bool(relu)
        ctx.save_for_backward(y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _capi.load()
        y, = ctx.saved_tensors
        dy = dy.contiguous()
        C = dy.shape[-1]
        M = dy.numel() // C
        dx = torch.empty_like(dy) if ctx.relu else dy
        dbias = _empty((C,), dy)
        wsb = lib._lpm_bias_act_bwd_workspace_bytes(M, C)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=dy.device)
        lib.check(lib._lpm_bias_act_bwd(ptr(dy), ptr(y), 1 if ctx.relu else 0, M, C, ptr(dx) if ctx.relu else None, ptr(dbias), ptr(ws),
                                        wsb, stream_ptr()), "lpm_bias_act_bwd")
        return dx, dbias, None


def bias_act_ok(y, bias):
    return (y.is_cuda and y.dtype == torch.float32 and y.is_contiguous() and y.shape[-1] % 4 == 0 and y.data_ptr() % 16 == 0
            and bias is not None and bias.dtype == torch.float32 and y.numel() >= (1 << 16))


def bias_act(y, bias, relu):
    """y must be a tensor nobody else holds (the fresh output of a GEMM): it is overwritten."""
    return _BiasAct.apply(y, bias, bool(relu))


def skinny_weight_grad(x, dy, out=None):
    """dW [N1,N2] = x^T dy for a skinny batch (x [R,N1], dy [R,N2], R %% 16 == 0, N2 %% 32 == 0) on the bf16 pipe with
    split-bf16 operands: one pass, written straight into ``out``."""
    lib = _capi.load()
    R, N1 = x.shape
    N2 = dy.shape[1]
    st = stream_ptr()
    xt = _tile_buffer(lib._lpm_weight_tiles_bytes(R, N1), x)
    dyt = _tile_buffer(lib._lpm_weight_tiles_bytes(R, N2), x)
    lib.check(lib._lpm_split_weight_tiles(ptr(x), R, N1, 0, ptr(xt), st), "lpm_split_weight_tiles")
    lib.check(lib._lpm_split_weight_tiles(ptr(dy), R, N2, 0, ptr(dyt), st), "lpm_split_weight_tiles")
    if out is None:
        out = _empty((N1, N2), x)
    with _timed("skinny_weight_grad", (R, N1, N2)):
        lib.check(lib._lpm_skinny_weight_grad_tiles(ptr(xt), ptr(dyt), R, N1, N2, ptr(out), st), "lpm_skinny_weight_grad_tiles")
    return out


class _Projection(torch.autograd.Function):
    """Plain library GEMMs, arranged for a skinny-M / huge-K problem: the forward is split-K (a batched GEMM over
    K-slices + a tiny reduction: 0.21 ms instead of 0.77 ms for the one-shot GEMM hipBLASLt picks at M = 80), and the
    weight gradient (554 MB at cfg-2, 85 % of all gradient bytes) is written straight into the trainer's gradient
    arena when the weight carries ``_lpm_grad_view`` -- autograd never sees it (``_lpm_grad_ready`` is called instead
    of a post-accumulate hook)."""

    @staticmethod
    def forward(ctx, x, W):
        M, K = x.shape
        N = W.shape[1]
        S = next((s for s in (132, 128, 96, 64, 48, 32, 16, 8) if K % s == 0 and K // s >= 512), 1)
        ctx.save_for_backward(x, W)
        ctx.stream_kernels = (PROJ_STREAM and x.is_cuda and x.dtype == torch.float32 and W.dtype == torch.float32 and x.stride(1) == 1
                              and x.stride(0) >= K and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
                              and W.is_contiguous() and bool(_capi.load()._lpm_proj_supported(M, K, N)))
        if ctx.stream_kernels:
            # hand-written weight-stream kernel: W is read once as fp32 and split into bf16 planes in registers
            lib = _capi.load()
            y = _empty((M, N), x)
            wsb = lib._lpm_proj_fwd_workspace_bytes(M, K, N)
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device)
            with _timed("proj_fwd", (M, K, N)):
                lib.check(lib._lpm_proj_fwd(ptr(x), x.stride(0), ptr(W), M, K, N, ptr(y), ptr(ws), wsb, stream_ptr()), "lpm_proj_fwd")
            return y
        if S == 1 or M > 512:
            return x.matmul(W)
        return torch.bmm(x.view(M, S, K // S).transpose(0, 1), W.view(S, K // S, N)).sum(0)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0] and ctx.stream_kernels and W.shape[1] >= PROJ_DX_STREAM_MIN_N:
            lib = _capi.load()
            M, K = x.shape
            N = W.shape[1]
            dyt = _tile_buffer(lib._lpm_row_tiles_bytes(1, M, N), dy)
            lib.check(lib._lpm_split_rows_tiles(ptr(dy), N, 1, M, N, ptr(dyt), stream_ptr()), "lpm_split_rows_tiles")
            # the gradient gets the row stride of x: a padded descriptor buffer (ops.DescriptorSlots) keeps both off the channel aliasing
            dx = torch.empty_strided((M, K), (x.stride(0), 1), dtype=torch.float32, device=x.device)
            with _timed("proj_dx", (M, K, N)):
                lib.check(lib._lpm_proj_dx(ptr(dyt), ptr(W), M, K, N, ptr(dx), dx.stride(0), stream_ptr()), "lpm_proj_dx")
        elif ctx.needs_input_grad[0]:
            dx = dy.matmul(W.t())
        if not ctx.needs_input_grad[1]:
            return dx, None
        skinny = x.shape[0] % 16 == 0 and dy.shape[1] % 32 == 0 and x.is_contiguous()
        factored = getattr(W, "_lpm_factored", None)
        if factored is not None and factored.armed and factored.strict and not (skinny and x.is_cuda and not factored.puts):
            # data parallel: the route selects between two different collectives (all-gather of the factors / all-reduce of bucket 0)
            # and was agreed across the towers in Trainer.build -- a rank must never switch on local grounds (it would hang the job)
            raise LpmError(f"hidden projection backward: the towers agreed on the factored gradient route at build time, but this "
                           f"rank's step does not fit it (clips {x.shape[0]} not a multiple of 16, non-contiguous input, or the "
                           f"weight used twice): use a fixed per-rank batch or FLAGS.hidden1_factored_update = False")
        if factored is not None and factored.armed and factored.puts and getattr(factored, "early_done", False):
            # the first use's product has already been consumed by the update that ran INSIDE this backward (Trainer,
            # FLAGS.hidden1_early_update) -- the weight, its moments and its compute copy are being rewritten on the update stream.  A
            # second product could only be dropped (its arena slice is skipped) and its dx above has read a weight in flux.
            raise LpmError("hidden projection backward: hidden1_weights was used twice in one step, but its update already ran inside "
                           "this backward (FLAGS.hidden1_early_update): set FLAGS.hidden1_early_update = False for such a model")
        if factored is not None and factored.armed and skinny and x.is_cuda and not factored.puts:
            # The trainer's optimiser consumes this gradient as the PRODUCT x^T dy (FactoredGradient): only the two operands leave.
            # (A batch that is not a multiple of 16 clips, or a second use of the weight, takes the generic route below and the
            # trainer, finding nothing pending, runs the generic update.)
            factored.put(x, dy)
            return dx, None
        view = getattr(W, "_lpm_grad_view", None)
        if view is None:
            return dx, (skinny_weight_grad(x, dy) if skinny else x.t().matmul(dy))
        # The trainer owns this gradient's storage (a slice of the flat gradient arena): write it there and hand autograd
        # nothing -- returning the tensor would make AccumulateGrad copy all 554 MB of it into a fresh .grad.
        if getattr(W, "_lpm_grad_written", False):          # second use of the weight in one step: accumulate
            view += skinny_weight_grad(x, dy) if skinny else x.t().matmul(dy)
        elif skinny:
            skinny_weight_grad(x, dy, out=view)             # bf16 pipe, split-bf16 operands, one 554 MB pass
        else:
            torch.mm(x.t(), dy, out=view)
        W._lpm_grad_written = True
        ready = getattr(W, "_lpm_grad_ready", None)
        if ready is not None:
            ready()                                         # e.g. start this bucket's all-reduce under the rest of backward
        return dx, None


def projection(x, W):
    return _Projection.apply(x, W)


class _ProjectionParts(torch.autograd.Function):
    """y = [x1 * scale | x2] . W (tf.concat + tf.matmul, frame_level_models.py:2445 + :2319) with x1 the lazily normalised d-major
    descriptor of the video stream (vlad_aggregate(lazy=True): un-normalised sums + one scale per (clip, cluster)) and x2 the audio
    stream's descriptor: lpm_proj_fwd_parts reads both where they are and applies the scale as it loads the operand; the weight gradient's
    X factor is written as tiles by lpm_split_weight_tiles_parts the same way.  The input gradient dx = dy . W^T is ONE [M, Kd] buffer whose
    column blocks go back as views (K3 reads a strided gradient in place); for x1 it is the gradient with respect to the NORMALISED
    descriptor (the lazily normalised form's contract)."""

    @staticmethod
    def forward(ctx, x1h, stored, scale, ks, x2, W):
        """x1h: the tensor autograd sees (its gradient is dx1); stored: the sums when x1h is only a handle (bf16 storage), else None."""
        lib = _capi.load()
        x1 = stored if stored is not None else x1h
        M, n1a = x1.shape
        n1b = x2.shape[1] if x2 is not None else 0
        Kd, N = n1a + n1b, W.shape[1]
        if not projection_parts_ok(x1, scale, ks, x2, W):
            raise LpmError("projection_parts: shapes / layouts outside lpm_proj_fwd_parts' conditions (use ops.materialise + ops.projection)")
        ctx.save_for_backward(x1, scale, x2, W)
        ctx.ks = int(ks)
        y = _empty((M, N), W)
        wsb = lib._lpm_proj_fwd_workspace_bytes(M, Kd, N)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=x1.device)
        # bf16 storage with a compute copy of the weight attached (ops.ComputeCopy): both passes read the copy
        cc = getattr(W, "_lpm_w16", None) if PROJ_W16 else None
        ctx.w16 = cc.tensor(W) if (cc is not None and x1.dtype == torch.bfloat16) else None
        with _timed("proj_fwd", (M, Kd, N)):
            if ctx.w16 is not None:
                lib.check(lib._lpm_proj_fwd_parts_w16(ptr(x1), x1.stride(0), n1a, ptr(scale), int(ks), ptr(x2), x2.stride(0) if x2 is not None else 0,
                                                      ptr(ctx.w16), M, Kd, N, ptr(y), ptr(ws), wsb, stream_ptr()), "lpm_proj_fwd_parts_w16")
            else:
                lib.check(lib._lpm_proj_fwd_parts(ptr(x1), x1.stride(0), n1a, int(x1.dtype == torch.bfloat16), ptr(scale), int(ks), ptr(x2),
                                                  x2.stride(0) if x2 is not None else 0, ptr(W), M, Kd, N, ptr(y), ptr(ws), wsb, stream_ptr()),
                          "lpm_proj_fwd_parts")
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _capi.load()
        x1, scale, x2, W = ctx.saved_tensors
        M, n1a = x1.shape
        n1b = x2.shape[1] if x2 is not None else 0
        Kd, N = n1a + n1b, W.shape[1]
        dy = dy.contiguous()
        st = stream_ptr()
        dx1 = dx2 = None
        want_dx = ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[4])
        factored = getattr(W, "_lpm_factored", None)
        # the input gradient from INSIDE the weight's update pass (FLAGS.hidden1_fold_input_gradient): the trainer offered it for this step
        # (``fold_dx``), the forward read the compute copy, the shape fits -- then the factors are handed over first and dx comes back
        fold = (want_dx and ctx.needs_input_grad[5] and factored is not None and factored.armed and not factored.puts
                and getattr(factored, "fold_dx", False) and getattr(ctx, "w16", None) is not None and M % 16 == 0
                and bool(lib._lpm_factored_fold_supported(M, Kd, N)))
        if fold:
            dx = _empty((M, Kd), W)
            factored.dx_out = dx
            xt = _tile_buffer(lib._lpm_weight_tiles_bytes(M, Kd), x1)
            lib.check(lib._lpm_split_weight_tiles_parts(ptr(x1), x1.stride(0), n1a, int(x1.dtype == torch.bfloat16), ptr(scale), ctx.ks, ptr(x2),
                                                        x2.stride(0) if x2 is not None else 0, M, Kd, ptr(xt), st), "lpm_split_weight_tiles_parts")
            factored.put_tiles(xt, dy, M, Kd)                     # -> Trainer._factored_put: clip + Adam + dx in one pass over the weight
            factored.dx_out = None
            if factored.dx_done:
                return dx[:, :n1a], None, None, None, (dx[:, n1a:] if x2 is not None else None), None
            want_dx, dx = True, None                             # (the trainer did not run the update now: dx by its own pass below)
        if want_dx:
            if N >= PROJ_DX_STREAM_MIN_N:
                dyt = _tile_buffer(lib._lpm_row_tiles_bytes(1, M, N), dy)
                lib.check(lib._lpm_split_rows_tiles(ptr(dy), N, 1, M, N, ptr(dyt), st), "lpm_split_rows_tiles")
                dx = _empty((M, Kd), W)
                with _timed("proj_dx", (M, Kd, N)):
                    if getattr(ctx, "w16", None) is not None and N % 64 == 0:
                        lib.check(lib._lpm_proj_dx_w16(ptr(dyt), ptr(ctx.w16), M, Kd, N, ptr(dx), dx.stride(0), st), "lpm_proj_dx_w16")
                    else:
                        lib.check(lib._lpm_proj_dx(ptr(dyt), ptr(W), M, Kd, N, ptr(dx), dx.stride(0), st), "lpm_proj_dx")
            else:
                dx = dy.matmul(W.t())
            dx1 = dx[:, :n1a]
            dx2 = dx[:, n1a:] if x2 is not None else None
        if not ctx.needs_input_grad[5]:
            return dx1, None, None, None, dx2, None

        def x_tiles():
            xt = _tile_buffer(lib._lpm_weight_tiles_bytes(M, Kd), x1)
            lib.check(lib._lpm_split_weight_tiles_parts(ptr(x1), x1.stride(0), n1a, int(x1.dtype == torch.bfloat16), ptr(scale), ctx.ks, ptr(x2),
                                                        x2.stride(0) if x2 is not None else 0, M, Kd, ptr(xt), st), "lpm_split_weight_tiles_parts")
            return xt

        skinny = M % 16 == 0 and N % 32 == 0
        if fold:                                                 # (the factors are with the trainer already; only dx was missing)
            return dx1, None, None, None, dx2, None
        if factored is not None and factored.armed and factored.strict and not (skinny and not factored.puts):
            raise LpmError("hidden projection backward: the towers agreed on the factored gradient route at build time, but this rank's "
                           "step does not fit it (clips not a multiple of 16, or the weight used twice)")
        if factored is not None and factored.armed and skinny and not factored.puts:
            factored.put_tiles(x_tiles(), dy, M, Kd)
            return dx1, None, None, None, dx2, None
        if skinny:
            dyt2 = _tile_buffer(lib._lpm_weight_tiles_bytes(M, N), dy)
            lib.check(lib._lpm_split_weight_tiles(ptr(dy), M, N, 0, ptr(dyt2), st), "lpm_split_weight_tiles")
            view = getattr(W, "_lpm_grad_view", None)
            fresh = view is None or getattr(W, "_lpm_grad_written", False)
            out = _empty((Kd, N), W) if fresh else view
            with _timed("skinny_weight_grad", (M, Kd, N)):
                lib.check(lib._lpm_skinny_weight_grad_tiles(ptr(x_tiles()), ptr(dyt2), M, Kd, N, ptr(out), st), "lpm_skinny_weight_grad_tiles")
            dW = out
        else:
            xm = _Materialise.forward(None, x1.float(), scale, ctx.ks)
            dW = (torch.cat([xm, x2], 1) if x2 is not None else xm).t().matmul(dy)
            view = getattr(W, "_lpm_grad_view", None)
            fresh = True
        if view is None:
            return dx1, None, None, None, dx2, dW
        if getattr(W, "_lpm_grad_written", False):
            view += dW
        elif fresh:
            view.copy_(dW)
        W._lpm_grad_written = True
        ready = getattr(W, "_lpm_grad_ready", None)
        if ready is not None:
            ready()
        return dx1, None, None, None, dx2, None


def projection_parts_ok(x1, scale, ks, x2, W):
    """lpm_proj_fwd_parts' conditions (include/lpm_hip.h) for these tensors."""
    if not (PROJ_STREAM and x1.is_cuda and x1.dtype in (torch.float32, torch.bfloat16) and W.dtype == torch.float32 and W.is_contiguous()
            and x1.dim() == 2 and scale is not None and scale.is_contiguous() and scale.dtype == torch.float32):
        return False
    M, n1a = x1.shape
    n1b = 0
    if x2 is not None:
        if not (x2.dim() == 2 and x2.shape[0] == M and x2.dtype == torch.float32 and x2.stride(1) == 1 and x2.stride(0) % 4 == 0
                and x2.data_ptr() % 16 == 0 and x2.shape[1] % 16 == 0):
            return False
        n1b = x2.shape[1]
    ks = int(ks)
    return (x1.stride(1) == 1 and x1.stride(0) % 4 == 0 and x1.data_ptr() % 16 == 0 and ks > 0 and ks % 4 == 0 and n1a % 32 == 0
            and n1a % ks == 0 and tuple(scale.shape) == (M, ks) and W.shape[0] == n1a + n1b
            and bool(_capi.load()._lpm_proj_supported(M, n1a + n1b, W.shape[1])))


def projection_parts(x1, x2, W):
    """[descriptor(x1) | x2] . W for a lazily normalised d-major x1 (see _ProjectionParts); x2 may be None."""
    return _ProjectionParts.apply(x1, getattr(x1, "_lpm_raw", None), row_scale_of(x1), int(getattr(x1, "_lpm_scale_ks", 0)), x2, W)


class _BatchNormRows(torch.autograd.Function):
    """Training-mode slim.batch_norm over the rows of a channel-last tensor: column statistics, normalisation and the moving
    averages in three launches; backward = lpm_bn_bwd (two column reductions + one elementwise pass)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, biased_moving):
        lib = _capi.load()
        x2 = _f32(x, "batch_norm input").contiguous().view(-1, x.shape[-1])
        M, C = x2.shape
        y = torch.empty_like(x2)
        mean, var = _empty((C,), x2), _empty((C,), x2)
        wsb = lib._lpm_bn_rows_workspace_bytes(M, C)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=x2.device)
        lib.check(lib._lpm_bn_rows_fwd(ptr(x2), M, C, ptr(gamma), ptr(beta), BN_EPS, BN_DECAY, 1 if biased_moving else 0, ptr(y),
                                       ptr(mean), ptr(var), ptr(moving_mean), ptr(moving_var), ptr(ws), wsb, stream_ptr()),
                  "lpm_bn_rows_fwd")
        ctx.save_for_backward(x2, mean, var, gamma)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        lib = _capi.load()
        x2, mean, var, gamma = ctx.saved_tensors
        M, C = x2.shape
        dy2 = dy.contiguous().view(M, C)
        dx = torch.empty_like(x2)
        dgamma, dbeta = _empty((C,), x2), _empty((C,), x2)
        wsb = lib._lpm_bn_bwd_workspace_bytes(M, C)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=x2.device)
        lib.check(lib._lpm_bn_bwd(ptr(dy2), ptr(x2), ptr(mean), ptr(var), ptr(gamma), BN_EPS, M, C, ptr(dx), ptr(dgamma), ptr(dbeta),
                                  ptr(ws), wsb, stream_ptr()), "lpm_bn_bwd")
        return dx.view(ctx.shape), dgamma, dbeta, None, None, None


class _BatchNormRowsAct(torch.autograd.Function):
    """slim.batch_norm(act(x + bias)) over the rows of a channel-last tensor, x the RAW output of the dense layer in front: the bias
    add and ReLU ride in the batch norm's statistics / apply passes and in both passes of its backward (the activation is never
    stored; the backward returns the gradient of x with the ReLU mask applied and the bias gradient)."""

    @staticmethod
    def forward(ctx, x, bias, relu, gamma, beta, moving_mean, moving_var, biased_moving):
        lib = _capi.load()
        x2 = _f32(x, "batch_norm input").contiguous().view(-1, x.shape[-1])
        M, C = x2.shape
        bias = bias.contiguous()
        y = torch.empty_like(x2)
        mean, var = _empty((C,), x2), _empty((C,), x2)
        wsb = lib._lpm_bn_rows_workspace_bytes(M, C)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=x2.device)
        lib.check(lib._lpm_bn_rows_act_fwd(ptr(x2), ptr(bias), 1 if relu else 0, M, C, ptr(gamma), ptr(beta), BN_EPS, BN_DECAY,
                                           1 if biased_moving else 0, ptr(y), ptr(mean), ptr(var), ptr(moving_mean), ptr(moving_var), ptr(ws),
                                           wsb, stream_ptr()), "lpm_bn_rows_act_fwd")
        ctx.save_for_backward(x2, bias, mean, var, gamma)
        ctx.shape, ctx.relu = x.shape, bool(relu)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        lib = _capi.load()
        x2, bias, mean, var, gamma = ctx.saved_tensors
        M, C = x2.shape
        dy2 = dy.contiguous().view(M, C)
        dx = torch.empty_like(x2)
        dgamma, dbeta, dbias = _empty((C,), x2), _empty((C,), x2), _empty((C,), x2)
        wsb = lib._lpm_bn_act_bwd_workspace_bytes(M, C)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=x2.device)
        lib.check(lib._lpm_bn_act_bwd(ptr(dy2), ptr(x2), ptr(bias), 1 if ctx.relu else 0, ptr(mean), ptr(var), ptr(gamma), BN_EPS, M, C,
                                      ptr(dx), ptr(dgamma), ptr(dbeta), ptr(dbias), ptr(ws), wsb, stream_ptr()), "lpm_bn_act_bwd")
        return dx.view(ctx.shape), dbias, None, dgamma, dbeta, None, None, None


class _FFNModX3(torch.autograd.Function):
    """FeedForwardNetworkMod up to its second dense layer (transformer_utils.py:741-756): pre2 = BN(relu(y W1 + b1)) W2 as ONE node.
    The [M, 4F] tensor between the dense layers exists in fp32 only as the first GEMM's raw output (which the batch norm's statistics
    and backward need); the batch norm writes its result ONLY as the second GEMM's operand image (lpm_bn_rows_act_image_fwd) and its
    backward writes the gradient of the raw output ONLY as the first layer's gradient image (lpm_bn_act_bwd_image): two 393 MB fp32
    tensors and two operand-split passes per encoder and step are gone (NetVladV2 at cfg-3)."""

    @staticmethod
    def forward(ctx, y2d, W1, b1, gamma, beta, moving_mean, moving_var, W2, y3=None):
        """y3: the operand image of y2d when its producer (the layer norm in front) wrote one."""
        lib = _capi.load()
        y2d = _rows(y2d, "ffn input")
        W1_0, W2_0 = W1, W2
        W1, W2 = _f32(W1, "W1").contiguous(), _f32(W2, "W2").contiguous()
        M, F = y2d.shape
        C = W1.shape[1]
        s1, s2 = _site("a", W1_0), _site("a", W2_0)        # operand sites (fp16 planes when the trainer's scales are calibrated)
        f16 = _f16(s1)
        ctx.sites = (s1, s2)
        dt = torch.float16 if f16 else torch.bfloat16
        if y3 is None or y3.dtype != dt:
            y3 = _split_rows(y2d, site=s1)
        w13n, w13k = _split_weight(W1, f16=f16)
        pre = _mm3(y3, w13n, alpha=s1.inv if s1 is not None else 1.0)      # raw output of the first dense layer [M, C]
        b1 = b1.contiguous()
        f3 = torch.empty((M, 3 * C), dtype=dt, device=y2d.device)
        mean, var = _empty((C,), pre), _empty((C,), pre)
        wsb = lib._lpm_bn_rows_workspace_bytes(M, C)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=pre.device)
        lib.check(lib._lpm_bn_rows_act_image_fwd_fmt(ptr(pre), ptr(b1), 1, M, C, ptr(gamma), ptr(beta), BN_EPS, BN_DECAY, 1, ptr(f3), ptr(mean),
                                                     ptr(var), ptr(moving_mean), ptr(moving_var), ptr(ws), wsb,
                                                     s2.fmt if s2 is not None else None, stream_ptr()), "lpm_bn_rows_act_image_fwd")
        w23n, w23k = _split_weight(W2, f16=f16)
        ctx.save_for_backward(y3, pre, b1, mean, var, gamma, f3, w13k, w23k)
        ctx.dims = (F, C, W2.shape[1])
        ctx.wrefs = (W1_0, W2_0)
        return _mm3(f3, w23n, alpha=s2.inv if s2 is not None else 1.0)

    @staticmethod
    def backward(ctx, dout):
        lib = _capi.load()
        y3, pre, b1, mean, var, gamma, f3, w13k, w23k = ctx.saved_tensors
        F, C, N = ctx.dims
        M = y3.shape[0]
        s1, s2 = ctx.sites
        g1, g2 = _site("g", ctx.wrefs[0]), _site("g", ctx.wrefs[1])
        f16 = _f16(s1)
        do3 = _split_rows(dout.contiguous(), grad=True, site=g2)
        dW2 = _dw_x3(f3, do3, C, N, outs=[(ctx.wrefs[1], 0, N)], sa=s2, sg=g2)[0]
        df = _mm3(do3, w23k, alpha=g2.inv if g2 is not None else 1.0)      # gradient of the batch norm's output [M, C]
        dp3 = torch.empty((M, (2 if f16 else 3) * C), dtype=torch.float16 if f16 else torch.bfloat16, device=df.device)
        dgamma, dbeta, db1 = _empty((C,), df), _empty((C,), df), _empty((C,), df)
        wsb = lib._lpm_bn_act_bwd_workspace_bytes(M, C)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=df.device)
        lib.check(lib._lpm_bn_act_bwd_image_fmt(ptr(df), ptr(pre), ptr(b1), 1, ptr(mean), ptr(var), ptr(gamma), BN_EPS, M, C, ptr(dp3), ptr(dgamma),
                                                ptr(dbeta), ptr(db1), ptr(ws), wsb, g1.fmt if g1 is not None else None, stream_ptr()),
                  "lpm_bn_act_bwd_image")
        del df
        dy = _mm3(dp3, w13k, alpha=g1.inv if g1 is not None else 1.0) if ctx.needs_input_grad[0] else None
        dW1 = _dw_x3(y3, dp3, F, C, outs=[(ctx.wrefs[0], 0, C)], sa=s1, sg=g1)[0]
        return dy, dW1, db1, dgamma, dbeta, None, None, dW2, None


_ZEROS = {}       # (device, length) -> a zero vector, never written


class _BNDenseX3(torch.autograd.Function):
    """slim.batch_norm(x) . W as ONE node (MultiHeadAttentionBN's attention_bn -> output_transform, transformer_utils.py:666-677): the batch
    norm writes its result ONLY as the GEMM's operand image (lpm_bn_rows_act_image_fwd with a zero bias and no activation) -- the
    normalised tensor never exists in fp32 and there is no operand-split pass; the backward is the dense layer's followed by lpm_bn_bwd."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, W):
        lib = _capi.load()
        x2 = _f32(x, "batch_norm input").contiguous()
        W0 = W
        W = _f32(W, "dense kernel").contiguous()
        M, C = x2.shape
        sa = ctx.site_a = _site("a", W0)
        f16 = _f16(sa)
        f3 = torch.empty((M, 3 * C), dtype=torch.float16 if f16 else torch.bfloat16, device=x2.device)
        mean, var = _empty((C,), x2), _empty((C,), x2)
        zero = _ZEROS.get((x2.device, C))
        if zero is None:
            zero = _ZEROS[(x2.device, C)] = torch.zeros(C, dtype=torch.float32, device=x2.device)       # (the kernel's bias operand: none here)
        wsb = lib._lpm_bn_rows_workspace_bytes(M, C)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=x2.device)
        lib.check(lib._lpm_bn_rows_act_image_fwd_fmt(ptr(x2), ptr(zero), 0, M, C, ptr(gamma), ptr(beta), BN_EPS, BN_DECAY, 1, ptr(f3), ptr(mean),
                                                     ptr(var), ptr(moving_mean), ptr(moving_var), ptr(ws), wsb,
                                                     sa.fmt if sa is not None else None, stream_ptr()), "lpm_bn_rows_act_image_fwd")
        w3n, w3k = _split_weight(W, f16=f16)
        ctx.save_for_backward(x2, mean, var, gamma, f3, w3k)
        ctx.dims = (C, W.shape[1])
        ctx.wrefs = (W0,)
        return _mm3(f3, w3n, alpha=sa.inv if sa is not None else 1.0)

    @staticmethod
    def backward(ctx, dout, do3=None):
        """do3 (block Functions only): dout already as its gradient image, in the format of this layer's "g" site."""
        lib = _capi.load()
        x2, mean, var, gamma, f3, w3k = ctx.saved_tensors
        C, N = ctx.dims
        M = x2.shape[0]
        sa, sg = ctx.site_a, _site("g", ctx.wrefs[0])
        if do3 is None:
            do3 = _split_rows(dout.contiguous(), grad=True, site=sg)
        dW = _dw_x3(f3, do3, C, N, outs=[(ctx.wrefs[0], 0, N)], sa=sa, sg=sg)[0]
        df = _mm3(do3, w3k, alpha=sg.inv if sg is not None else 1.0)       # gradient of the batch norm's output [M, C]
        dx = torch.empty_like(x2)
        dgamma, dbeta = _empty((C,), x2), _empty((C,), x2)
        wsb = lib._lpm_bn_bwd_workspace_bytes(M, C)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=x2.device)
        lib.check(lib._lpm_bn_bwd(ptr(df), ptr(x2), ptr(mean), ptr(var), ptr(gamma), BN_EPS, M, C, ptr(dx), ptr(dgamma), ptr(dbeta),
                                  ptr(ws), wsb, stream_ptr()), "lpm_bn_bwd")
        return dx, dgamma, dbeta, None, None, dW


# "0": layers.batch_norm + layers.dense as two nodes with an operand-split pass in between (A/B)
BN_DENSE_FUSED = os.environ.get("LPM_BN_DENSE_FUSED", "1") != "0"


def bn_dense_x3_ok(x, units):
    rows = x.numel() // x.shape[-1]
    return (BN_DENSE_FUSED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[-1] % 8 == 0 and units % 8 == 0 and rows >= 1024)


def bn_dense_x3(x, gamma, beta, moving_mean, moving_var, W, bias=None):
    """x [..., C] -> slim.batch_norm(x) . W (+ bias) [..., N] (training mode; moving statistics updated in place, biased variance: the
    rank-3 path).  The bias add runs in place on the GEMM's own output (ops.bias_act), before that output is viewed in x's shape."""
    rows = x.numel() // x.shape[-1]
    out = _BNDenseX3.apply(x.reshape(rows, x.shape[-1]), gamma, beta, moving_mean, moving_var, W)
    if bias is not None:
        out = bias_act(out, bias, False) if (BIAS_ACT_FUSED and bias_act_ok(out, bias)) else out + bias
    return out.reshape(*x.shape[:-1], W.shape[1])


def ffn_mod_x3_ok(x, filter_size, final_size):
    rows = x.numel() // x.shape[-1]
    return (FFN_MOD_FUSED and BN_ACT_FUSED and x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 8 == 0 and filter_size % 8 == 0
            and final_size % 8 == 0 and rows >= 1024 and bool(_capi.load()._lpm_bn_act_bwd_supported(rows, filter_size)))


def ffn_mod_x3(x, W1, b1, gamma, beta, moving_mean, moving_var, W2):
    """x [..., F] -> BN(relu(x W1 + b1)) W2 [..., N] (training mode; the batch norm's moving statistics are updated in place, biased
    variance: the rank-3 path of slim.batch_norm)."""
    rows = x.numel() // x.shape[-1]
    out = _FFNModX3.apply(x.reshape(rows, x.shape[-1]), W1, b1, gamma, beta, moving_mean, moving_var, W2, _take_image(x, rows))
    return out.reshape(*x.shape[:-1], W2.shape[1])


def _take_image(y, rows):
    """The operand image a producer attached to y (``y._lpm_y3``: ops._ResidualLayerNorm(image=True)) -- only if y is still the tensor it
    was taken from (same storage, not modified in place since); consumed either way."""
    tag = getattr(y, "_lpm_y3", None)
    if tag is None:
        return None
    img, dptr, ver = tag
    try:
        del y._lpm_y3
    except AttributeError:
        pass
    ok = dptr == y.data_ptr() and ver == y._version and y.is_contiguous() and tuple(img.shape) == (rows, 3 * y.shape[-1])
    return img if ok else None


def batch_norm_rows_act_ok(x, bias):
    C = x.shape[-1]
    return (BN_ACT_FUSED and x.is_cuda and x.dtype == torch.float32 and bias is not None and C % 4 == 0
            and bool(_capi.load()._lpm_bn_act_bwd_supported(x.numel() // C, C)))


def batch_norm_rows_act(x, bias, relu, gamma, beta, moving_mean, moving_var, biased_moving_variance):
    return _BatchNormRowsAct.apply(x, bias, bool(relu), gamma, beta, moving_mean, moving_var, bool(biased_moving_variance))


def batch_norm_rows(x, gamma, beta, moving_mean, moving_var, biased_moving_variance):
    """Training-mode batch norm of a channel-last tensor (channels = last axis, statistics over all other axes); updates the
    moving statistics in place.  x.shape[-1] %% 4 == 0."""
    return _BatchNormRows.apply(x, gamma, beta, moving_mean, moving_var, bool(biased_moving_variance))


# ----------------------------------------------------------------------------------------------
# residual add + layer_norm (TF1 defaults: moments over all non-batch axes)
# ----------------------------------------------------------------------------------------------
LN_EPS = 1e-12
LN_FEATURES = (128, 256, 512, 1024)


class OutputSlot:
    """A [B, L, F] clip-slot of a wider [B, total] buffer (ops.DescriptorSlots): handed to a Function as a plain Python
    object so that autograd does not track the buffer; ``view()`` is the strided tensor the kernel writes."""

    def __init__(self, base, offset, L, F):
        self.base, self.offset, self.L, self.F = base, int(offset), int(L), int(F)

    def view(self):
        B, tot = self.base.shape
        return torch.as_strided(self.base, (B, self.L, self.F), (tot, self.F, 1), self.offset)


def _batch_strided(t, L, F):
    """t as the layer-norm kernels can read / write it: [B, L, F] with contiguous clips at a 16-byte-aligned batch stride."""
    return (t.dim() == 3 and t.stride(2) == 1 and t.stride(1) == F and t.stride(0) >= L * F and t.stride(0) % 4 == 0
            and t.data_ptr() % 16 == 0)


class _ResidualLayerNorm(torch.autograd.Function):
    """y = layer_norm(act(a + bias) + r): TF1 joint moments, with the producing dense layer's bias add / ReLU fused in."""

    @staticmethod
    def forward(ctx, a, r, gamma, beta, bias, relu, out=None, r_scale=None, image=False, mask=None, mask_scale=1.0, site=None):
        """r_scale [B * L] (block Functions only): r holds the rows of a lazily normalised descriptor, scaled as they are read.
        image: y is ALSO written as the [B*L, 3F] bf16 activation image of the dense layer that reads it next and attached to the
        result as ``y._lpm_y3`` (ops._FFNX3 / ops.ffn_mod_x3 take it instead of splitting y again).
        mask [B, L, F] uint8 / bool + mask_scale: a dropout between the dense layer and the layer norm (NetVladV2): the keep mask and
        1 / keep probability apply to act(a + bias) before the residual is added; the backward returns the gradient of a through it."""
        lib = _capi.load()
        a = _f32(a, "layer_norm input").contiguous()
        B, L, F = a.shape
        r = r.contiguous() if r is not None else None
        bias = bias.contiguous() if bias is not None else None
        y = out.view() if out is not None else torch.empty_like(a)
        if tuple(y.shape) != (B, L, F) or not _batch_strided(y, L, F):
            raise LpmError("layer_norm: output slot does not match [B, L, F] with contiguous clips")
        z = torch.empty_like(a) if (r is not None or bias is not None) else a
        stats = _empty((B, 2), a)
        wsb = lib._lpm_layer_norm_workspace_bytes(B, F)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=a.device)
        # site (block Functions only): the input site of the dense layer that reads y next -- the image in its operand format
        y3 = torch.empty((B * L, 3 * F), dtype=torch.float16 if _f16(site) else torch.bfloat16, device=a.device) if image else None
        if mask is not None:
            if r_scale is not None:
                raise LpmError("layer_norm: a dropout mask and a residual row scale do not combine")
            mask = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.contiguous()
            if mask.dtype != torch.uint8 or tuple(mask.shape) != (B, L, F):
                raise LpmError("layer_norm: the dropout keep mask must be uint8 / bool [B, L, F]")
            z = torch.empty_like(a) if z is a else z
            lib.check(lib._lpm_layer_norm_act_mask_image_fwd_fmt(ptr(a), ptr(bias), 1 if relu else 0, ptr(mask), float(mask_scale), ptr(r), ptr(gamma),
                                                                 ptr(beta), B, L, F, LN_EPS, ptr(y), y.stride(0), ptr(y3), ptr(z), ptr(stats),
                                                                 ptr(ws), wsb, site.fmt if site is not None else None, stream_ptr()),
                      "lpm_layer_norm_act_mask_image_fwd")
        elif image:
            lib.check(lib._lpm_layer_norm_act_image_fwd_fmt(ptr(a), ptr(bias), 1 if relu else 0, ptr(r), ptr(r_scale), ptr(gamma), ptr(beta), B, L,
                                                            F, LN_EPS, ptr(y), y.stride(0), ptr(y3), ptr(z) if z is not a else None, ptr(stats),
                                                            ptr(ws), wsb, site.fmt if site is not None else None, stream_ptr()),
                      "lpm_layer_norm_act_image_fwd")
        elif r_scale is not None:
            lib.check(lib._lpm_layer_norm_act_fwd_rs(ptr(a), ptr(bias), 1 if relu else 0, ptr(r), ptr(r_scale), ptr(gamma), ptr(beta), B, L, F,
                                                     LN_EPS, ptr(y), y.stride(0), ptr(z) if z is not a else None, ptr(stats), ptr(ws),
                                                     wsb, stream_ptr()), "lpm_layer_norm_act_fwd_rs")
        else:
            lib.check(lib._lpm_layer_norm_act_fwd(ptr(a), ptr(bias), 1 if relu else 0, ptr(r), ptr(gamma), ptr(beta), B, L, F, LN_EPS,
                                                  ptr(y), y.stride(0), ptr(z) if z is not a else None, ptr(stats), ptr(ws), wsb,
                                                  stream_ptr()), "lpm_layer_norm_act_fwd")
        ctx.has_r, ctx.relu, ctx.has_bias = r is not None, bool(relu), bias is not None
        ctx.mask, ctx.mask_scale = mask, float(mask_scale)
        ctx.save_for_backward(z, stats, gamma, a if relu else None, bias)
        if y3 is not None:
            # handed to the next block with the identity of the tensor it images: a consumer must see the same storage, untouched
            y._lpm_y3 = (y3, y.data_ptr(), y._version)
        return y

    @staticmethod
    def backward(ctx, dy, dr_extra=None, da_image=False, site=None):
        """Block Functions only -- dr_extra: a second consumer's gradient of the residual tensor, added to the residual's
        gradient on its way out of the kernel; da_image: da is returned as the [B*L, 3F] bf16 gradient image the next GEMMs
        read (ops._split_rows(grad=True)) instead of in fp32."""
        lib = _capi.load()
        z, stats, gamma, a, bias = ctx.saved_tensors
        B, L, F = z.shape
        if not _batch_strided(dy, L, F):          # a column slice of a wider gradient buffer is read in place
            dy = dy.contiguous()
        dz = torch.empty_like(z)
        mask = getattr(ctx, "mask", None)
        # site (block Functions only): the gradient site of the dense layer whose output a is -- the image in its operand format
        img = (torch.empty((B * L, (2 if _f16(site) else 3) * F), dtype=torch.float16 if _f16(site) else torch.bfloat16, device=z.device)
               if da_image else None)
        da = torch.empty_like(z) if ((ctx.relu or dr_extra is not None or mask is not None) and not da_image) else None
        dgamma, dbeta = _empty((F,), z), _empty((F,), z)
        dbias = _empty((F,), z) if ctx.has_bias else None
        wsb = lib._lpm_layer_norm_workspace_bytes(B, F)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=z.device)
        if mask is not None:
            lib.check(lib._lpm_layer_norm_act_mask_bwd_fmt(ptr(dy), dy.stride(0), ptr(z), ptr(stats), ptr(gamma), ptr(a), ptr(bias),
                                                           1 if ctx.relu else 0, ptr(mask), ctx.mask_scale, B, L, F, ptr(dz), ptr(da), ptr(dgamma),
                                                           ptr(dbeta), ptr(dbias), ptr(dr_extra), ptr(img), ptr(ws), wsb,
                                                           site.fmt if site is not None else None, stream_ptr()),
                      "lpm_layer_norm_act_mask_bwd")
        else:
            lib.check(lib._lpm_layer_norm_act_bwd_fmt(ptr(dy), dy.stride(0), ptr(z), ptr(stats), ptr(gamma), ptr(a), ptr(bias),
                                                      1 if ctx.relu else 0, B, L, F, ptr(dz), ptr(da), ptr(dgamma), ptr(dbeta), ptr(dbias),
                                                      ptr(dr_extra), ptr(img), ptr(ws), wsb, site.fmt if site is not None else None,
                                                      stream_ptr()), "lpm_layer_norm_act_bwd")
        first = img if da_image else (da if da is not None else dz)
        return first, (dz if ctx.has_r else None), dgamma, dbeta, dbias, None, None, None, None, None, None, None


def residual_layer_norm(a, r, gamma, beta, bias=None, relu=False, image=False, mask=None, mask_scale=1.0, next_kernel=None):
    """layer_norm(act(a + bias) + r) with TF1 joint moments; a, r: [B, L, F]; act = relu when ``relu`` (needs ``bias``).
    image: the result ALSO leaves as the operand image of the dense layer that reads it next, attached as ``y._lpm_y3`` (ops.ffn_mod_x3
    takes it instead of splitting y again)."""
    site = _site("a", next_kernel) if (image and LN_IMAGE and next_kernel is not None) else None
    return _ResidualLayerNorm.apply(a, r, gamma, beta, bias, bool(relu), None, None, bool(image) and LN_IMAGE, mask, float(mask_scale), site)


# NetVladV2: tf.layers.dropout between output_transform and the layer norm rides in the layer norm's passes; "0": separate passes (A/B)
LN_DROPOUT_FUSED = os.environ.get("LPM_LN_DROPOUT_FUSED", "1") != "0"


# ----------------------------------------------------------------------------------------------
# K4: attention core
# ----------------------------------------------------------------------------------------------
def _mha_dims(q, num_heads):
    if q.dim() != 3 or q.stride(2) != 1 or q.stride(0) != q.shape[1] * q.stride(1):
        raise LpmError("attention operands must be [B, L, h*d] with contiguous rows")
    B, L, F = q.shape
    if F % num_heads:
        raise LpmError("hidden size not divisible by num_heads")
    return B, L, F // num_heads


def _bn_pass_precision(which, L=None):
    """MHA_BN_PRECISION = "mixed": per-pass arithmetic of the logits_bn attention, as 'fwd,stats,main' in MHA_BN_MIXED ("auto": by the
    number of keys L, see MHA_BN_X3_MIN_KEYS)."""
    if MHA_BN_PRECISION != "mixed":
        return MHA_BN_PRECISION
    fwd, stats, main = MHA_BN_MIXED.split(",")
    prec = {"fwd": fwd, "stats": stats, "main": main}[which]
    if prec == "auto":
        prec = "bf16x3" if (L is not None and L >= MHA_BN_X3_MIN_KEYS) else "f32"
    return prec


def _mha_fwd_fn(lib, bn=False, which="fwd", L=None):
    prec = _bn_pass_precision(which, L) if bn else MHA_PRECISION
    if prec == "bf16x3":
        return lib._lpm_mha_fwd_x3
    if prec == "f32":
        return lib._lpm_mha_fwd
    raise LpmError(f"unknown attention precision {prec!r} (bf16x3 | f32)")


def _mha_bwd_fn(lib, bn=False, which="main", L=None):
    prec = _bn_pass_precision(which, L) if bn else MHA_PRECISION
    if prec == "bf16x3":
        return lib._lpm_mha_bwd_x3
    if prec == "f32":
        return lib._lpm_mha_bwd
    raise LpmError(f"unknown attention precision {prec!r} (bf16x3 | f32)")


def _qkv_operands(q, k, v):
    """q, k, v as the kernels take them: [B, L, h*d] with unit column stride and ONE common row stride (column views of a
    fused [B, L, 3*h*d] projection qualify); anything else is made contiguous."""
    q, k, v = (_f32(t, "qkv") for t in (q, k, v))

    def ok(t):
        return t.dim() == 3 and t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1) and t.stride(1) % 4 == 0 \
            and t.data_ptr() % 16 == 0
    if ok(q) and ok(k) and ok(v) and q.stride(1) == k.stride(1) == v.stride(1):
        return q, k, v
    return q.contiguous(), k.contiguous(), v.contiguous()


def _dqkv_buffers(q):
    """dq, dk, dv as column views of one [B, L, 3*h*d] buffer (what ops._QKVX3.backward consumes without a copy)."""
    B, L, F = q.shape
    buf = torch.empty((B, L, 3 * F), dtype=torch.float32, device=q.device)
    return buf[..., :F], buf[..., F:2 * F], buf[..., 2 * F:]


class _MHACore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, num_heads, scale, image=False, site=None):
        """image (block Functions only, split-bf16 arithmetic): the result is written ONLY as the [B*L, 3*h*d] bf16 activation image
        of the output projection GEMM (and kept in that form for the backward).  site: that GEMM's input site -- the image in its
        operand format (fp16: the same three planes, fp16 words of o * scale)."""
        lib = _capi.load()
        q, k, v = _qkv_operands(q, k, v)
        B, L, d = _mha_dims(q, num_heads)
        lse = _empty((B, num_heads, L), q)
        ctx.dims = (B, L, num_heads, d, scale)
        ctx.o_image = bool(image)
        ctx.o_site = site
        if image:
            o = torch.empty((B * L, 3 * num_heads * d), dtype=torch.float16 if _f16(site) else torch.bfloat16, device=q.device)
            with _timed("mha_fwd", (B, L, num_heads, d)):
                lib.check(lib._lpm_mha_fwd_x3_image_fmt(ptr(q), ptr(k), ptr(v), q.stride(1), B, L, num_heads, d, scale, ptr(o), ptr(lse),
                                                        site.fmt if site is not None else None, stream_ptr()), "lpm_mha_fwd_x3_image")
        else:
            o = torch.empty(q.shape, dtype=torch.float32, device=q.device)
            with _timed("mha_fwd", (B, L, num_heads, d)):
                lib.check(_mha_fwd_fn(lib)(ptr(q), ptr(k), ptr(v), q.stride(1), B, L, num_heads, d, scale, None, None, ptr(o),
                                           o.stride(1), ptr(lse), stream_ptr()), "lpm_mha_fwd")
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, do, image=False, site=None):
        """image (block Functions only, split-bf16 arithmetic): return the q/k/v gradients as the [B*L, 9*h*d] bf16 gradient
        image of [dq | dk | dv] that ops._QKVX3.backward feeds its GEMMs, written by the kernels themselves.  site: the q/k/v layer's
        gradient site -- the image in its operand format (fp16 two-product: [B*L, 6*h*d] fp16 = [hi(3N) | lo(3N)])."""
        lib = _capi.load()
        B, L, h, d, scale = ctx.dims
        q, k, v, o, lse = ctx.saved_tensors
        do = do.contiguous()
        o_image = getattr(ctx, "o_image", False)
        o_site = getattr(ctx, "o_site", None)
        if image and (site is not None or o_site is not None):
            if not o_image:
                raise LpmError("mha backward: operand sites need the image-form attention output")
            dy3 = torch.empty((B * L, (6 if _f16(site) else 9) * h * d), dtype=torch.float16 if _f16(site) else torch.bfloat16, device=q.device)
            lib.check(lib._lpm_mha_bwd_x3_image_fmt(ptr(q), ptr(k), ptr(v), q.stride(1), ptr(o), o_site.fmt if o_site is not None else None,
                                                    ptr(do), do.stride(1), ptr(lse), B, L, h, d, scale, ptr(dy3),
                                                    site.fmt if site is not None else None, stream_ptr()), "lpm_mha_bwd_x3_image")
            return dy3
        if image:
            dy3 = torch.empty((B * L, 9 * h * d), dtype=torch.bfloat16, device=q.device)
            lib.check(lib._lpm_mha_bwd_x3_image(ptr(q), ptr(k), ptr(v), q.stride(1), ptr(o), 1 if o_image else 0, ptr(do), do.stride(1),
                                                ptr(lse), B, L, h, d, scale, ptr(dy3), stream_ptr()), "lpm_mha_bwd_x3_image")
            return dy3
        if o_image:
            raise LpmError("mha backward: an image-form attention output needs the image-form backward")
        dq, dk, dv = _dqkv_buffers(q)
        lib.check(_mha_bwd_fn(lib)(ptr(q), ptr(k), ptr(v), q.stride(1), ptr(o), ptr(do), o.stride(1), ptr(lse), B, L, h, d,
                                   scale, None, None, ptr(dq), ptr(dk), ptr(dv), dq.stride(1), None, None, None, stream_ptr()),
                  "lpm_mha_bwd")
        return dq, dk, dv, None, None, None


def mha_core(q, k, v, num_heads, scale):
    """softmax(scale * q k^T) v per head on [B, L, h*d] projections (transformer_utils.py:564-581)."""
    return _MHACore.apply(q, k, v, int(num_heads), float(scale), False)


# ----------------------------------------------------------------------------------------------
# a6 as two block Functions: the sub-layers run as plain routines (their own forward / backward code, a stand-in ctx) so
# that the block's backward decides where gradients of a shared tensor meet -- in a GEMM's beta = 1 or on a layer-norm
# kernel's store instead of in three [B*L, F] add passes per encoder that autograd would insert.
# ----------------------------------------------------------------------------------------------
class _SubCtx:
    def __init__(self):
        self.needs_input_grad = (True,) * 12
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def _pack_subs(ctx, subs):
    flat, counts = [], []
    for c in subs:
        flat += list(c.saved_tensors)
        counts.append(len(c.saved_tensors))
        c.saved_tensors = ()
    ctx.save_for_backward(*flat)
    ctx.subs, ctx.counts = subs, counts


def _unpack_subs(ctx):
    flat, i = ctx.saved_tensors, 0
    for c, n in zip(ctx.subs, ctx.counts):
        c.saved_tensors = flat[i:i + n]
        i += n
    return ctx.subs


class _AttnBlockX3(torch.autograd.Function):
    """y = layer_norm(MHA(x, x) Wo + bo + x)  (transformer_utils.py:403-407 with :552-586): fused q/k/v GEMM, K4, output
    GEMM, fused bias + residual layer norm.  Backward: the residual's gradient dz is the beta = 1 operand of the q/k/v
    input-gradient GEMM."""

    @staticmethod
    def forward(ctx, x, Wq, Wk, Wv, Wo, bo, gamma, beta, num_heads, scale, row_scale=None, next_kernel=None):
        """row_scale [B, L]: x is a lazily normalised descriptor (netvlad(lazy=True)) -- its two readers, the q/k/v operand split and the
        residual layer norm, scale the rows as they read them; the returned gradient is with respect to the normalised x."""
        x = _f32(x, "attention block input").contiguous()
        B, L, F = x.shape
        N = Wq.shape[1]
        rs = row_scale.reshape(-1).contiguous() if row_scale is not None else None
        cq, cm, co, cl = _SubCtx(), _SubCtx(), _SubCtx(), _SubCtx()
        q, k, v = _QKVX3.forward(cq, x.view(B * L, F), Wq, Wk, Wv, row_scale=rs)
        from . import FLAGS
        if MHA_PRECISION == "bf16x3" and FLAGS.mha_gradient_image:     # the attention result only as the output GEMM's operand image
            o3 = _MHACore.forward(cm, q.view(B, L, N), k.view(B, L, N), v.view(B, L, N), num_heads, scale, image=True, site=_site("a", Wo))
            att = _DenseX3.forward(co, None, Wo, x3=o3)
        else:
            o = _MHACore.forward(cm, q.view(B, L, N), k.view(B, L, N), v.view(B, L, N), num_heads, scale)
            att = _DenseX3.forward(co, o.view(B * L, N), Wo)
        # (the feed-forward block behind this one reads y as a GEMM operand: the layer norm writes that image on its way out, in the
        # operand format of that block's first dense layer -- ``next_kernel``)
        y = _ResidualLayerNorm.forward(cl, att.view(B, L, Wo.shape[1]), x, gamma, beta, bo, False, None, rs, image=LN_IMAGE,
                                       site=_site("a", next_kernel) if next_kernel is not None else None)
        ctx.wq = Wq
        _pack_subs(ctx, (cq, cm, co, cl))
        ctx.shape = (B, L, F, N)
        return y

    @staticmethod
    def backward(ctx, dy):
        cq, cm, co, cl = _unpack_subs(ctx)
        B, L, F, N = ctx.shape
        from . import FLAGS
        _, dz, dgamma, dbeta, dbo = _ResidualLayerNorm.backward(cl, dy)[:5]         # no ReLU: da is dz
        do, dWo = _DenseX3.backward(co, dz.view(B * L, F))   # (dz also as an image from the kernel: measured neutral, not kept)
        if cm.o_image:
            dy3 = _MHACore.backward(cm, do.view(B, L, N), image=True, site=_site("g", ctx.wq))   # the kernels write the GEMM operand image
            dx, dWq, dWk, dWv = _QKVX3.backward(cq, None, None, None, acc=dz.view(B * L, F), dy3=dy3)
        else:
            dq, dk, dv = _MHACore.backward(cm, do.view(B, L, N))[:3]
            dx, dWq, dWk, dWv = _QKVX3.backward(cq, dq.view(B * L, N), dk.view(B * L, N), dv.view(B * L, N), acc=dz.view(B * L, F))
        return dx.view(B, L, F), dWq, dWk, dWv, dWo, dbo, dgamma, dbeta, None, None, None, None


def attention_block_x3(x, Wq, Wk, Wv, Wo, bo, gamma, beta, num_heads, scale, next_kernel=None):
    """x may be a lazily normalised descriptor (ops.netvlad(lazy=True)): its row scale is applied where the block reads the rows.
    next_kernel: the kernel of the dense layer that reads the block's result next (FeedForwardNetwork's first layer): the operand
    image the layer norm hands it is written in that layer's operand format."""
    return _AttnBlockX3.apply(x, Wq, Wk, Wv, Wo, bo, gamma, beta, int(num_heads), float(scale), row_scale_of(x), next_kernel)


def _ln_pair_forward(c1, c2, a, r, g1, be1, bias, g2, be2, out=None):
    """layer_norm(layer_norm(relu(a + bias) + r; g1, be1) + r; g2, be2) through lpm_layer_norm_pair_fwd; fills the two sub-contexts
    exactly as two _ResidualLayerNorm.forward calls would, so the backward is unchanged."""
    lib = _capi.load()
    a = _f32(a, "layer_norm input").contiguous()
    B, L, F = a.shape
    r, bias = r.contiguous(), bias.contiguous()
    y = out.view() if out is not None else torch.empty_like(a)
    if tuple(y.shape) != (B, L, F) or not _batch_strided(y, L, F):
        raise LpmError("layer_norm: output slot does not match [B, L, F] with contiguous clips")
    z1, z2 = torch.empty_like(a), torch.empty_like(a)
    stats1, stats2 = _empty((B, 2), a), _empty((B, 2), a)
    wsb = lib._lpm_layer_norm_workspace_bytes(B, F)
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=a.device)
    lib.check(lib._lpm_layer_norm_pair_fwd(ptr(a), ptr(bias), 1, ptr(r), ptr(g1), ptr(be1), ptr(g2), ptr(be2), B, L, F, LN_EPS, ptr(y),
                                           y.stride(0), ptr(z1), ptr(stats1), ptr(z2), ptr(stats2), ptr(ws), wsb, stream_ptr()),
              "lpm_layer_norm_pair_fwd")
    c1.has_r, c1.relu, c1.has_bias = True, True, True
    c1.save_for_backward(z1, stats1, g1, a, bias)
    c2.has_r, c2.relu, c2.has_bias = True, False, False
    c2.save_for_backward(z2, stats2, g2, None, None)
    return y


class _FFNBlockX3(torch.autograd.Function):
    """out = layer_norm(n + y),  n = layer_norm(relu(relu(y W1 + b1) W2 + b2) + y)  (transformer_utils.py:409-411 with
    :696-715).  y feeds the first GEMM and both residuals; backward: the outer layer norm's dz2 is the inner one's incoming
    gradient AND its residual's second gradient (added on the inner kernel's store), and the sum is the beta = 1 operand of
    the first GEMM's input-gradient GEMM."""

    @staticmethod
    def forward(ctx, y, W1, b1, W2, b2, g1, be1, g2, be2, out=None):
        y = _f32(y, "ffn block input").contiguous()
        B, L, F = y.shape
        cf, c1, c2 = _SubCtx(), _SubCtx(), _SubCtx()
        # the layer norm's operand image of y (ops._ResidualLayerNorm(image=True)), only if y is still the tensor it was taken from:
        # modified in place, re-allocated or replaced in between, the image is dropped and y is split again
        y3 = None
        tag = getattr(y, "_lpm_y3", None)
        if tag is not None:
            img, dptr, ver = tag
            if dptr == y.data_ptr() and ver == y._version:
                y3 = img
            try:
                del y._lpm_y3                   # consumed: the image lives as long as this block's saved tensors, not as long as y
            except AttributeError:
                pass
        pre = _FFNX3.forward(cf, y.view(B * L, F), W1, b1, W2, y3=y3)
        from . import FLAGS
        if FLAGS.ln_pair_forward:      # three passes for the two layer norms: n itself is never stored
            out = _ln_pair_forward(c1, c2, pre.view(B, L, F), y, g1, be1, b2, g2, be2, out)
        else:
            n = _ResidualLayerNorm.forward(c1, pre.view(B, L, F), y, g1, be1, b2, True)
            out = _ResidualLayerNorm.forward(c2, n, y, g2, be2, None, False, out)
        _pack_subs(ctx, (cf, c1, c2))
        ctx.shape = (B, L, F)
        return out

    @staticmethod
    def backward(ctx, dout):
        cf, c1, c2 = _unpack_subs(ctx)
        B, L, F = ctx.shape
        dz2, _, dg2, dbe2 = _ResidualLayerNorm.backward(c2, dout)[:4]             # gradient of n; y receives the same
        from . import FLAGS
        if FLAGS.ln_gradient_image:               # the ReLU-masked da1 only feeds the FFN's GEMMs: it leaves the kernel as their image
            do3, dzy, dg1, dbe1, db2 = _ResidualLayerNorm.backward(c1, dz2, dr_extra=dz2, da_image=True, site=_site("g", cf.wrefs[1]))[:5]
            dy, dW1, db1, dW2 = _FFNX3.backward(cf, None, acc=dzy.view(B * L, F), do3=do3)
        else:
            da1, dzy, dg1, dbe1, db2 = _ResidualLayerNorm.backward(c1, dz2, dr_extra=dz2)[:5]   # dzy = dz1 + dz2
            dy, dW1, db1, dW2 = _FFNX3.backward(cf, da1.view(B * L, F), acc=dzy.view(B * L, F))
        return dy.view(B, L, F), dW1, db1, dW2, db2, dg1, dbe1, dg2, dbe2, None


def ffn_block_x3(y, W1, b1, W2, b2, g1, be1, g2, be2, out=None):
    """out: an ops.OutputSlot -- the block's result is written straight into that clip-slot of a wider buffer."""
    return _FFNBlockX3.apply(y, W1, b1, W2, b2, g1, be1, g2, be2, out)


class DescriptorSlots:
    """The pooled descriptors of the streams side by side in ONE [B, sum_i L_i * F_i] buffer (tf.concat(..., 1) at
    frame_level_models.py:2309): every stream's encoder writes its slot in place, ``join`` hands the buffer to the
    projection and routes the column slices of its gradient back -- no concat copy forward, no slice copies backward."""

    def __init__(self, batch, dims, like):
        self.dims = [(int(L), int(F)) for L, F in dims]
        offs, cur = [], 0
        for L, F in self.dims:
            offs.append(cur)
            cur += L * F
        self.total = cur
        self.base = torch.empty((batch, cur), dtype=torch.float32, device=like.device)
        self.slots = [OutputSlot(self.base, o, L, F) for o, (L, F) in zip(offs, self.dims)]
        self.offsets = offs

    def join(self, *parts):
        return _JoinSlots.apply(self, *parts)


class _JoinSlots(torch.autograd.Function):
    @staticmethod
    def forward(ctx, slots, *parts):
        for p, slot in zip(parts, slots.slots):
            v = slot.view()
            if p.data_ptr() != v.data_ptr() or tuple(p.shape) != tuple(v.shape) or p.stride() != v.stride():
                raise LpmError("DescriptorSlots.join: a part was not written into its slot")
        ctx.slots = slots
        return slots.base.view_as(slots.base)

    @staticmethod
    def backward(ctx, d):
        sl = ctx.slots
        B = d.shape[0]
        out = []
        for o, (L, F) in zip(sl.offsets, sl.dims):
            out.append(d[:, o:o + L * F].view(B, L, F))
        return (None, *out)


class _MHACoreBN(torch.autograd.Function):
    """MultiHeadAttentionBN core: batch_norm over the key-position channel of the rank-4 logits, then
    softmax . v (transformer_utils.py:652-661).  Training statistics over (B, h, query)."""

    @staticmethod
    def forward(ctx, q, k, v, gamma, beta, moving_mean, moving_var, num_heads, is_training):
        lib = _capi.load()
        q, k, v = _qkv_operands(q, k, v)
        B, L, d = _mha_dims(q, num_heads)
        h = num_heads
        moments = None
        if is_training:
            partial = _empty((B * h, 2, L), q)
            # (lpm_mha_logit_stats_moments reads q and k as float4: 16-byte aligned views only; anything else takes the general kernel)
            if MHA_BN_MOMENTS and d in (8, 16) and q.data_ptr() % 16 == 0 and k.data_ptr() % 16 == 0 and (q.stride(1) * 4) % 16 == 0:
                # the statistics from the d x d moments of q, which the backward's repair of dk needs again (kept: 272 floats per head)
                moments = _empty((B * h, d * d + d), q)
                lib.check(lib._lpm_mha_logit_stats_moments(ptr(q), ptr(k), q.stride(1), B, L, h, d, ptr(partial), ptr(moments), stream_ptr()),
                          "lpm_mha_logit_stats_moments")
            else:
                lib.check(lib._lpm_mha_logit_stats(ptr(q), ptr(k), q.stride(1), B, L, h, d, ptr(partial), stream_ptr()),
                          "lpm_mha_logit_stats")
            mean, var, kscale, kshift = bn_fold(partial, B * h, L, B * h * L, gamma, beta, moving_mean, moving_var)
        else:
            kscale, kshift = folded_eval_affine(gamma, beta, moving_mean, moving_var)
            kscale, kshift = kscale.contiguous(), kshift.contiguous()
            mean, var = moving_mean.detach().clone(), moving_var.detach().clone()
        o = torch.empty(q.shape, dtype=torch.float32, device=q.device)
        lse = _empty((B, h, L), q)
        lib.check(_mha_fwd_fn(lib, bn=True, L=L)(ptr(q), ptr(k), ptr(v), q.stride(1), B, L, h, d, 1.0, ptr(kscale), ptr(kshift), ptr(o),
                                   o.stride(1), ptr(lse), stream_ptr()), "lpm_mha_fwd")
        ctx.dims = (B, L, h, d, is_training)
        ctx.moments = moments
        ctx.save_for_backward(q, k, v, o, lse, kscale, kshift, mean, var, gamma)
        return o

    @staticmethod
    def backward(ctx, do, image=False, site=None):
        """image (ops._QKVAttnBNX3 only): the q / k / v gradients as the [B*L, 9 h d] bf16 / [B*L, 6 h d] fp16 gradient image of
        [dq | dk | dv] that ops._QKVX3.backward feeds its GEMMs (site: the q/k/v layer's gradient site), written by the three kernels of
        the one-pass backward themselves; returned in dq's place with dk = dv = None.  Where the one-pass form does not apply the plain
        gradients come back and the caller splits them."""
        lib = _capi.load()
        B, L, h, d, is_training = ctx.dims
        q, k, v, o, lse, kscale, kshift, mean, var, gamma = ctx.saved_tensors
        do = do.contiguous()
        st = stream_ptr()
        onepass = (MHA_BN_ONEPASS and is_training and d in (8, 16) and _bn_pass_precision("stats", L) == "bf16x3"
                   and _bn_pass_precision("main", L) == "bf16x3")
        if onepass and image:
            N = h * d
            f16 = _f16(site)
            fmt = site.fmt if site is not None else None
            dy3 = torch.empty((B * L, (6 if f16 else 9) * N), dtype=torch.float16 if f16 else torch.bfloat16, device=q.device)
            dkp = torch.empty((B * L, N), dtype=torch.float32, device=q.device)      # dk before the statistics' share is taken out
            partial = _empty((B * h, 2, L), q)
            lib.check(lib._lpm_mha_bwd_x3_bn_image_fmt(ptr(q), ptr(k), ptr(v), q.stride(1), ptr(o), ptr(do), o.stride(1), ptr(lse), B, L, h, d, 1.0,
                                                       ptr(kscale), ptr(kshift), ptr(dkp), None, None, ptr(partial), ptr(dy3), fmt, st),
                      "lpm_mha_bwd_x3_bn_image(dk, dv, statistics)")
            dgamma, dbeta, corr_a, corr_b = (_empty((L,), q) for _ in range(4))
            lib.check(lib._lpm_mha_bn_corrections(ptr(partial), B * h, L, ptr(mean.contiguous()), ptr(var.contiguous()), ptr(kscale), BN_EPS,
                                                  B * h * L, ptr(dgamma), ptr(dbeta), ptr(corr_a), ptr(corr_b), st), "lpm_mha_bn_corrections")
            lib.check(lib._lpm_mha_bwd_x3_bn_image_fmt(ptr(q), ptr(k), ptr(v), q.stride(1), ptr(o), ptr(do), o.stride(1), ptr(lse), B, L, h, d, 1.0,
                                                       ptr(kscale), ptr(kshift), None, ptr(corr_a), ptr(corr_b), None, ptr(dy3), fmt, st),
                      "lpm_mha_bwd_x3_bn_image(dq)")
            lib.check(lib._lpm_mha_bn_dk_correct_image(ptr(q), ptr(k), q.stride(1), B, L, h, d, 1.0, ptr(corr_a), ptr(corr_b), ptr(dkp),
                                                       ptr(ctx.moments), ptr(dy3), fmt, st), "lpm_mha_bn_dk_correct_image")
            return dy3, None, None, dgamma, dbeta, None, None, None, None
        if onepass:
            # ONE pass over the scores for dk, dv AND the statistics: the batch statistics' share of ds is affine in the raw score, so its
            # part of dk is a d-vector and a d x d matrix per (batch, head) away (lpm_mha_bn_dk_correct); dq's pass applies it in place
            partial = _empty((B * h, 2, L), q)
            dq, dk, dv = _dqkv_buffers(q)
            lib.check(lib._lpm_mha_bwd_x3(ptr(q), ptr(k), ptr(v), q.stride(1), ptr(o), ptr(do), o.stride(1), ptr(lse), B, L, h, d, 1.0,
                                          ptr(kscale), ptr(kshift), None, ptr(dk), ptr(dv), dq.stride(1), None, None, ptr(partial), st),
                      "lpm_mha_bwd_x3(dk, dv, statistics)")
            dgamma, dbeta, corr_a, corr_b = (_empty((L,), q) for _ in range(4))
            lib.check(lib._lpm_mha_bn_corrections(ptr(partial), B * h, L, ptr(mean.contiguous()), ptr(var.contiguous()), ptr(kscale), BN_EPS,
                                                  B * h * L, ptr(dgamma), ptr(dbeta), ptr(corr_a), ptr(corr_b), st), "lpm_mha_bn_corrections")
            # (the repair of dk on an auxiliary stream beside dq's pass was measured slower: 10.62 vs 10.44 ms per step at cfg-3)
            lib.check(lib._lpm_mha_bwd_x3(ptr(q), ptr(k), ptr(v), q.stride(1), ptr(o), ptr(do), o.stride(1), ptr(lse), B, L, h, d, 1.0,
                                          ptr(kscale), ptr(kshift), ptr(dq), None, None, dq.stride(1), ptr(corr_a), ptr(corr_b), None, st),
                      "lpm_mha_bwd_x3(dq)")
            lib.check(lib._lpm_mha_bn_dk_correct(ptr(q), ptr(k), q.stride(1), B, L, h, d, 1.0, ptr(corr_a), ptr(corr_b), ptr(dk),
                                                 dk.stride(1), ptr(ctx.moments), st), "lpm_mha_bn_dk_correct")
            return dq, dk, dv, dgamma, dbeta, None, None, None, None
        # pass 1: column sums of dz and dz*s over (B, h, query)
        partial = _empty((B * h, 2, L), q)
        lib.check(_mha_bwd_fn(lib, bn=True, which="stats")(ptr(q), ptr(k), ptr(v), q.stride(1), ptr(o), ptr(do), o.stride(1), ptr(lse), B, L,
                                                           h, d, 1.0, ptr(kscale), ptr(kshift), None, None, None, q.stride(1), None, None,
                                                           ptr(partial), st), "lpm_mha_bwd(stats)")
        # dbeta = sum dz, dgamma = sum dz * s_hat and (training) the two correction vectors of the main pass, fp64, in one launch
        dgamma, dbeta = _empty((L,), q), _empty((L,), q)
        corr_a, corr_b = (_empty((L,), q), _empty((L,), q)) if is_training else (None, None)
        lib.check(lib._lpm_mha_bn_corrections(ptr(partial), B * h, L, ptr(mean.contiguous()), ptr(var.contiguous()), ptr(kscale), BN_EPS,
                                              B * h * L, ptr(dgamma), ptr(dbeta), ptr(corr_a), ptr(corr_b), st), "lpm_mha_bn_corrections")
        dq, dk, dv = _dqkv_buffers(q)
        lib.check(_mha_bwd_fn(lib, bn=True)(ptr(q), ptr(k), ptr(v), q.stride(1), ptr(o), ptr(do), o.stride(1), ptr(lse), B, L, h, d, 1.0,
                                   ptr(kscale), ptr(kshift), ptr(dq), ptr(dk), ptr(dv), dq.stride(1), ptr(corr_a), ptr(corr_b),
                                   None, st), "lpm_mha_bwd")
        return dq, dk, dv, dgamma, dbeta, None, None, None, None


def mha_core_bn(q, k, v, num_heads, gamma, beta, moving_mean, moving_var, is_training=True):
    return _MHACoreBN.apply(q, k, v, gamma, beta, moving_mean, moving_var, int(num_heads), bool(is_training))


# LPM_MHA_BN_GRAD_IMAGE=0: the q/k/v layer and the logits_bn attention as two autograd nodes again (fp32 dq | dk | dv + an operand-split pass)
MHA_BN_GRAD_IMAGE = os.environ.get("LPM_MHA_BN_GRAD_IMAGE", "1") != "0"


class _QKVAttnBNX3(torch.autograd.Function):
    """MultiHeadAttentionBN up to the combined heads (transformer_utils.py:559-561 + :652-661) as ONE node: the fused q/k/v GEMM and the
    logits_bn attention.  Forward: the two Functions' forwards unchanged.  Backward: the attention kernels write [dq | dk | dv] as the
    gradient image of the q/k/v layer's GEMMs -- at cfg-3 the fp32 [24000, 3072] gradient and the lpm_split_rows pass over it (190 us of
    an 8.85 ms step for the frames, 2 x 295 MB) never exist.  Same values bit for bit: the image is the split of the same fp32 numbers."""

    @staticmethod
    def forward(ctx, x, Wq, Wk, Wv, gamma, beta, moving_mean, moving_var, num_heads):
        x = _f32(x, "attention input").contiguous()
        B, L, F = x.shape
        N = Wq.shape[1]
        cq, cm = _SubCtx(), _SubCtx()
        cq.needs_input_grad = (ctx.needs_input_grad[0],) + (True,) * 11
        q, k, v = _QKVX3.forward(cq, x.view(B * L, F), Wq, Wk, Wv)
        o = _MHACoreBN.forward(cm, q.view(B, L, N), k.view(B, L, N), v.view(B, L, N), gamma, beta, moving_mean, moving_var, num_heads, True)
        ctx.wq = Wq
        _pack_subs(ctx, (cq, cm))
        ctx.shape = (B, L, F, N)
        return o

    @staticmethod
    def backward(ctx, do):
        cq, cm = _unpack_subs(ctx)
        B, L, F, N = ctx.shape
        got = _MHACoreBN.backward(cm, do, image=True, site=_site("g", ctx.wq))
        dgamma, dbeta = got[3], got[4]
        if got[1] is None:                                   # the kernels wrote the GEMMs' operand image
            dx, dWq, dWk, dWv = _QKVX3.backward(cq, None, None, None, dy3=got[0])
        else:
            dx, dWq, dWk, dWv = _QKVX3.backward(cq, got[0].reshape(B * L, N), got[1].reshape(B * L, N), got[2].reshape(B * L, N))
        return (dx.view(B, L, F) if dx is not None else None), dWq, dWk, dWv, dgamma, dbeta, None, None, None


DROPOUT_MASK_KERNEL = os.environ.get("LPM_DROPOUT_MASK_KERNEL", "1") != "0"       # "0": torch's bernoulli_ draws the keep masks (A/B)


def dropout_keep_mask(shape, keep_prob, device):
    """uint8 keep mask of tf.layers.dropout (1 = kept, probability keep_prob) for the layer norm that applies it (transformer_utils.py:450).
    On the GPU one store-stream launch (lpm_dropout_keep_mask) seeded from torch's CPU generator -- torch.manual_seed makes runs repeatable;
    elsewhere / switched off: torch's bernoulli_."""
    n = 1
    for d in shape:
        n *= int(d)
    if DROPOUT_MASK_KERNEL and device.type == "cuda" and n % 16 == 0 and 0.0 < keep_prob <= 1.0:
        seed = int(torch.empty((), dtype=torch.int64).random_().item()) & ((1 << 63) - 1)
        mask = torch.empty(tuple(shape), dtype=torch.uint8, device=device)
        lib = _capi.load()
        lib.check(lib._lpm_dropout_keep_mask(ptr(mask), n, float(keep_prob), seed, stream_ptr()), "lpm_dropout_keep_mask")
        return mask
    return torch.empty(tuple(shape), dtype=torch.uint8, device=device).bernoulli_(keep_prob)


# LPM_ATTN_BLOCK_BN=0: MultiHeadAttentionBN + the encoder's first layer norm as separate autograd nodes (A/B)
ATTN_BLOCK_BN = os.environ.get("LPM_ATTN_BLOCK_BN", "1") != "0"
ATTN_BLOCK_BN_LN_IMAGE = os.environ.get("LPM_ATTN_BLOCK_LN_IMAGE", "1") != "0"      # "0": the masked gradient in fp32 + an operand-split pass (A/B)


class _AttnBlockBNX3(torch.autograd.Function):
    """TransformerEncoderMod up to its first layer norm (transformer_utils.py:444-454 with :589-677) as ONE node:
    y = layer_norm(dropout(attention_bn(MHA_logits_bn(x, x)) Wo + bo) + x).  Forward: the four Functions' forwards unchanged (q/k/v GEMM,
    logits_bn attention, attention_bn + output transform, bias + dropout + residual layer norm).  Backward: what the node buys is WHERE the
    two gradients of x meet -- the residual's dz is the beta = 1 operand of the q/k/v input-gradient GEMM instead of a [B L, F] add pass that
    autograd would insert (49 us for the frames of cfg-3) -- and the attention's [dq | dk | dv] image as in ops._QKVAttnBNX3."""

    @staticmethod
    def forward(ctx, x, Wq, Wk, Wv, lgamma, lbeta, lmm, lmv, g2, b2, mm2, mv2, Wo, bo, gamma, beta, num_heads, mask, mask_scale, image, site,
                grad_join=None):
        ctx.join = None
        if grad_join is not None and GRAD_JOIN and ctx.needs_input_grad[0] and x.dtype == torch.float32:
            grad_join.accept()                    # a second reader of x (ops.vlad_aggregate) will leave its gradient there
            ctx.join = grad_join
        x = _f32(x, "attention block input").contiguous()
        B, L, F = x.shape
        N = Wq.shape[1]
        cq, cm, co, cl = _SubCtx(), _SubCtx(), _SubCtx(), _SubCtx()
        cq.needs_input_grad = (ctx.needs_input_grad[0],) + (True,) * 11
        q, k, v = _QKVX3.forward(cq, x.view(B * L, F), Wq, Wk, Wv)
        o = _MHACoreBN.forward(cm, q.view(B, L, N), k.view(B, L, N), v.view(B, L, N), lgamma, lbeta, lmm, lmv, num_heads, True)
        att = _BNDenseX3.forward(co, o.view(B * L, N), g2, b2, mm2, mv2, Wo)
        y = _ResidualLayerNorm.forward(cl, att.view(B, L, Wo.shape[1]), x, gamma, beta, bo, False, None, None, image=image, mask=mask,
                                       mask_scale=mask_scale, site=site)
        ctx.wq = Wq
        _pack_subs(ctx, (cq, cm, co, cl))
        ctx.shape = (B, L, F, N)
        return y

    @staticmethod
    def backward(ctx, dy):
        cq, cm, co, cl = _unpack_subs(ctx)
        B, L, F, N = ctx.shape
        extra = ctx.join.take() if ctx.join is not None else None        # the aggregation's gradient of the same frames (ops.GradJoin)
        if extra is not None:
            extra = extra.reshape(B, L, F).contiguous()
        if ATTN_BLOCK_BN_LN_IMAGE:
            # the gradient of att through the dropout mask leaves the layer norm as the operand image of the output transform's GEMMs
            img, dz, dgamma, dbeta, dbo = _ResidualLayerNorm.backward(cl, dy, dr_extra=extra, da_image=True, site=_site("g", co.wrefs[0]))[:5]
            do, dg2, db2, _, _, dWo = _BNDenseX3.backward(co, None, do3=img)
        else:
            first, dz, dgamma, dbeta, dbo = _ResidualLayerNorm.backward(cl, dy, dr_extra=extra)[:5]
            do, dg2, db2, _, _, dWo = _BNDenseX3.backward(co, first.view(B * L, -1))
        got = _MHACoreBN.backward(cm, do.view(B, L, N), image=True, site=_site("g", ctx.wq))
        acc = dz.view(B * L, F) if cq.needs_input_grad[0] else None
        if got[1] is None:
            dx, dWq, dWk, dWv = _QKVX3.backward(cq, None, None, None, acc=acc, dy3=got[0])
        else:
            dx, dWq, dWk, dWv = _QKVX3.backward(cq, got[0].reshape(B * L, N), got[1].reshape(B * L, N), got[2].reshape(B * L, N), acc=acc)
        return ((dx.view(B, L, F) if dx is not None else None), dWq, dWk, dWv, got[3], got[4], None, None, dg2, db2, None, None, dWo, dbo,
                dgamma, dbeta, None, None, None, None, None, None)


def attention_block_bn_x3(x, Wq, Wk, Wv, logits_bn, attention_bn, Wo, bo, gamma, beta, num_heads, mask, mask_scale, image=False, next_kernel=None,
                          grad_join=None):
    """logits_bn / attention_bn: (gamma, beta, moving_mean, moving_variance) of the two batch norms.  mask: the dropout KEEP mask
    (uint8 / bool [B, L, F]); image / next_kernel as ops.residual_layer_norm."""
    site = _site("a", next_kernel) if (image and LN_IMAGE and next_kernel is not None) else None
    return _AttnBlockBNX3.apply(x, Wq, Wk, Wv, *logits_bn, *attention_bn, Wo, bo, gamma, beta, int(num_heads), mask, float(mask_scale),
                                bool(image) and LN_IMAGE, site, grad_join)


def qkv_attention_bn_ok(x, hidden, num_heads):
    """Whether MultiHeadAttentionBN's self-attention front runs as ops.qkv_attention_bn_x3 for this [B, L, F] input (training only)."""
    return bool(MHA_BN_GRAD_IMAGE and MHA_BN_ONEPASS and x.is_cuda and x.dim() == 3 and hidden % num_heads == 0 and hidden // num_heads in (8, 16)
                and x.shape[1] <= 512 and _bn_pass_precision("stats", x.shape[1]) == "bf16x3" and _bn_pass_precision("main", x.shape[1]) == "bf16x3")


def qkv_attention_bn_x3(x, Wq, Wk, Wv, gamma, beta, moving_mean, moving_var, num_heads):
    return _QKVAttnBNX3.apply(x, Wq, Wk, Wv, gamma, beta, moving_mean, moving_var, int(num_heads))


# ----------------------------------------------------------------------------------------------
# a14 + a15: clip + Adam over flat arenas
# ----------------------------------------------------------------------------------------------
class _MoeCrossEntropy(torch.autograd.Function):
    """(gate_act, expert_act, labels) -> (predictions, loss): MoeModel's mixture tail + CrossEntropyLoss in one kernel each way."""

    @staticmethod
    def forward(ctx, gate_act, expert_act, labels, num_mixtures, eps):
        lib = _capi.load()
        gate_act, expert_act = _f32(gate_act, "gate activations").contiguous(), _f32(expert_act, "expert activations").contiguous()
        B = gate_act.shape[0]
        V = gate_act.shape[1] // (num_mixtures + 1)
        if gate_act.shape[1] != V * (num_mixtures + 1) or expert_act.shape != (B, V * num_mixtures):
            raise LpmError("moe_cross_entropy: activations do not match num_mixtures")
        pred = _empty((B, V), gate_act)
        loss = part = None
        if labels is not None:
            labels = labels.to(device=gate_act.device, dtype=torch.float32).contiguous()
            loss = _empty((), gate_act)
            part = _empty((lib._lpm_moe_ce_nblk(B, V),), gate_act)
        lib.check(lib._lpm_moe_ce_fwd(ptr(gate_act), ptr(expert_act), ptr(labels), B, V, num_mixtures, eps, ptr(pred), ptr(loss),
                                      ptr(part), stream_ptr()), "lpm_moe_ce_fwd")
        ctx.dims = (B, V, num_mixtures, eps)
        ctx.save_for_backward(gate_act, expert_act, labels)
        ctx.set_materialize_grads(False)      # an unused output (the predictions, in training) arrives as None, not as zeros
        if labels is None:
            ctx.mark_non_differentiable()
            return pred, None
        return pred, loss

    @staticmethod
    def backward(ctx, dpred, dloss):
        lib = _capi.load()
        B, V, m, eps = ctx.dims
        gate_act, expert_act, labels = ctx.saved_tensors
        dgate, dexpert = torch.empty_like(gate_act), torch.empty_like(expert_act)
        if dloss is None and labels is not None:
            dloss = torch.zeros((), dtype=torch.float32, device=gate_act.device)
        dloss = dloss.contiguous() if dloss is not None else None
        dpred = dpred.contiguous() if dpred is not None else None
        lib.check(lib._lpm_moe_ce_bwd(ptr(gate_act), ptr(expert_act), ptr(labels), ptr(dloss), ptr(dpred), B, V, m, eps, ptr(dgate),
                                      ptr(dexpert), stream_ptr()), "lpm_moe_ce_bwd")
        return dgate, dexpert, None, None, None


def moe_cross_entropy(gate_act, expert_act, labels, num_mixtures, eps=10e-6):
    """MoeModel mixture tail (video_level_models.py:116-126) + CrossEntropyLoss (losses.py:41-51) -> (predictions, loss);
    labels None -> (predictions, None)."""
    return _MoeCrossEntropy.apply(gate_act, expert_act, labels, int(num_mixtures), float(eps))


class ComputeCopy:
    """The bf16 compute copy of a weight beside its fp32 master (SURVEY section 7 hard part 2: "keep master fp32 + bf16 compute copy";
    BASELINE configs[4]).  The owner (train.Trainer) attaches it to the variable as ``W._lpm_w16``; FactoredGradient.clip_adam keeps it
    current from the update pass's epilogue (lpm_factored_clip_adam_copy), ops._ProjectionParts reads it.  Whatever else writes the
    master -- a checkpoint load, an update that did not go through that kernel -- makes the copy stale: torch's version counter catches
    writes through torch, ``invalidate()`` is for writes through raw pointers; a stale copy is rebuilt (one pass over the weight) at its
    next use, never read."""

    def __init__(self, W, also=()):
        """also: other tensors over the same memory whose in-place writes must be seen as well (the trainer's flat parameter arena: a
        variable is a view of it with a version counter of its own -- found by tools/determinism_check.py, which resets the ARENA)."""
        self.buf = torch.empty(W.shape, dtype=torch.bfloat16, device=W.device)
        self.also = tuple(also)
        self.version = None
        self.stale = True
        self.refreshes = 0               # full rebuilds (diagnostics: a training run shows one -- the first use)

    def invalidate(self):
        self.stale = True

    def _versions(self, W):
        return (W._version,) + tuple(t._version for t in self.also)

    def tensor(self, W):
        v = self._versions(W)
        if self.stale or self.version != v:
            self.buf.copy_(W.detach())                   # fp32 -> bf16, round to nearest even (the epilogue's rounding)
            self.version, self.stale = v, False
            self.refreshes += 1
        return self.buf

    def current(self, W):
        """The buffer for the update pass to write, or None when the copy is stale anyway (it will be rebuilt from the new master)."""
        return None if (self.stale or self.version != self._versions(W)) else self.buf


PROJ_W16 = os.environ.get("LPM_PROJ_W16", "1") != "0"       # "0": the projection ignores an attached compute copy (A/B)


class FactoredGradient:
    """The gradient of the hidden projection's weight as the product it is: dW = X^T DY with X [R, N1] the projection's input and
    DY [R, N2] the gradient of its output (R = clips).  ``put`` (called by _Projection.backward while ``armed``) keeps the two
    operands as split-bf16 weight tiles -- 86 MB + 0.2 MB at cfg-2 instead of the 554 MB gradient --, ``on_put`` lets the trainer
    start the towers' all-gather of the tiles (concatenating tile buffers along their leading step axis IS the product over all
    towers' clips: utils.combine_gradients' SUM), and ``clip_adam`` runs lpm_factored_clip_adam on the variable's arena slices."""

    def __init__(self, on_put=None, strict=False):
        self.armed = False
        self.strict = strict              # data parallel: the route is a cross-rank agreement, a step that does not fit it raises
        self.on_put = on_put
        self.clear()

    def clear(self):
        self.early_done = False           # the trainer's early update consumed this step's product inside backward (Trainer._factored_put)
        self.dx_out = None                # ops._ProjectionParts.backward: "form my input gradient here if the update runs now" ...
        self.dx_done = False              # ... and the trainer's answer (Trainer._factored_put)
        self.xt = self.dyt = None
        self.x = self.dy = None           # the fp32 factors themselves (this rank's; None once tiles of several towers were gathered)
        self.R = self.N1 = self.N2 = 0
        self.x_in_tiles = False           # put_tiles: X was never an fp32 matrix of its own
        self.puts = 0

    @property
    def pending(self):
        return self.xt is not None

    def put(self, x, dy):
        if self.puts:
            raise LpmError("FactoredGradient: the weight was used twice in one step; its gradient is then a sum of two products")
        lib = _capi.load()
        R, N1 = x.shape
        N2 = dy.shape[1]
        st = stream_ptr()
        xt = _tile_buffer(lib._lpm_weight_tiles_bytes(R, N1), x)
        dyt = _tile_buffer(lib._lpm_weight_tiles_bytes(R, N2), x)
        lib.check(lib._lpm_split_weight_tiles(ptr(x), R, N1, 0, ptr(xt), st), "lpm_split_weight_tiles")
        lib.check(lib._lpm_split_weight_tiles(ptr(dy.contiguous()), R, N2, 0, ptr(dyt), st), "lpm_split_weight_tiles")
        self.xt, self.dyt, self.R, self.N1, self.N2 = xt, dyt, R, N1, N2
        self.x, self.dy = x, dy
        self.puts += 1
        if self.on_put is not None:
            self.on_put(self)

    def put_tiles(self, xt, dy, R, N1):
        """``put`` for an X that exists only as its weight tiles (ops._ProjectionParts: the lazily normalised descriptor is scaled as the
        tiles are written); the quadratic-form norm reads X from the tiles."""
        if self.puts:
            raise LpmError("FactoredGradient: the weight was used twice in one step; its gradient is then a sum of two products")
        lib = _capi.load()
        N2 = dy.shape[1]
        dyt = _tile_buffer(lib._lpm_weight_tiles_bytes(R, N2), dy)
        lib.check(lib._lpm_split_weight_tiles(ptr(dy.contiguous()), R, N2, 0, ptr(dyt), stream_ptr()), "lpm_split_weight_tiles")
        self.xt, self.dyt, self.R, self.N1, self.N2 = xt, dyt, R, N1, N2
        self.x, self.dy = None, dy
        self.x_in_tiles = True
        self.puts += 1
        if self.on_put is not None:
            self.on_put(self)

    def materialise(self):
        """dW [N1, N2] from the operands held (tests / diagnostics): what the generic path would have written."""
        lib = _capi.load()
        out = torch.empty((self.N1, self.N2), dtype=torch.float32, device=self.xt.device)
        lib.check(lib._lpm_skinny_weight_grad_tiles(ptr(self.xt), ptr(self.dyt), self.R, self.N1, self.N2, ptr(out), stream_ptr()),
                  "lpm_skinny_weight_grad_tiles")
        return out

    def fold_supported(self):
        """Whether the update pass can return the projection's input gradient for the operands held (lpm_factored_fold_supported)."""
        return bool(self.pending and self.dy is not None and _capi.load()._lpm_factored_fold_supported(self.R, self.N1, self.N2))

    def clip_adam(self, param, m, v, clip_norm, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, scratch=None, param_bf16=None, dx=None):
        """param / m / v: the variable's [N1 * N2] slices of the arenas.  -> scratch (its last four floats: clip factor, norm, -, -).
        param_bf16: the variable's bf16 compute copy (ComputeCopy.buf), rewritten from the update pass's epilogue.
        dx [R, N1] (with param_bf16; ``fold_supported``): receives the projection's INPUT gradient DY W_old^T, formed inside the update
        pass from the weights it streams (lpm_factored_clip_adam_copy_dx) -- the caller is the projection's backward itself."""
        lib = _capi.load()
        if dx is not None and param_bf16 is None:
            raise LpmError("FactoredGradient.clip_adam: the input gradient rides in the update pass of a variable with a compute copy only")
        if param_bf16 is not None:
            return self._clip_adam_copy(lib, param, m, v, param_bf16, clip_norm, lr, step, beta1, beta2, eps, scratch, dx)
        nb = lib._lpm_factored_clip_adam_scratch_bytes(self.N1, self.N2)
        if scratch is None or scratch.numel() * 4 < nb:
            scratch = torch.empty(nb // 4, dtype=torch.float32, device=param.device)
        with _timed("factored_clip_adam", (self.R, self.N1, self.N2)):
            quad = (FACTORED_NORM_QUADFORM and self.x is not None and self.x.shape[0] == self.R and self.R <= 128 and self.x.stride(1) == 1
                    and self.x.dtype == torch.float32)
            tiles_only = (FACTORED_NORM_QUADFORM and self.x is None and self.x_in_tiles and self.dy is not None and self.dy.shape[0] == self.R
                          and self.R <= 128 and os.environ.get("LPM_FQ_TILES", "1") != "0")
            if tiles_only:
                # X lives in its tiles only (put_tiles): the quadratic forms read it from there (fa_quadform_kernel<true>; the fp32
                # pointer is the A/B form's operand and is not dereferenced)
                G = torch.mm(self.dy, self.dy.t())
                gdt = _tile_buffer(lib._lpm_row_tiles_bytes(1, self.R, self.R), G)
                lib.check(lib._lpm_split_rows_tiles(ptr(G), self.R, 1, self.R, self.R, ptr(gdt), stream_ptr()), "lpm_split_rows_tiles")
                lib.check(lib._lpm_factored_clip_adam_q(ptr(self.xt), ptr(self.dyt), ptr(self.xt), self.N1, ptr(gdt), self.R, self.N1,
                                                        self.N2, ptr(param), ptr(m), ptr(v), float(clip_norm), float(lr), beta1, beta2, eps,
                                                        int(step), ptr(scratch), nb, stream_ptr()), "lpm_factored_clip_adam_q")
            elif quad:
                # the norm from the quadratic forms x_n1^T (DY DY^T) x_n1 (lpm_factored_clip_adam_q): no first GEMM pass over R x N1 x N2
                G = torch.mm(self.dy, self.dy.t())      # [R, R] fp32 (its entries then enter the MFMAs as bf16 hi + lo: 2^-17 relative)
                gdt = _tile_buffer(lib._lpm_row_tiles_bytes(1, self.R, self.R), G)
                lib.check(lib._lpm_split_rows_tiles(ptr(G), self.R, 1, self.R, self.R, ptr(gdt), stream_ptr()), "lpm_split_rows_tiles")
                lib.check(lib._lpm_factored_clip_adam_q(ptr(self.xt), ptr(self.dyt), ptr(self.x), self.x.stride(0), ptr(gdt), self.R, self.N1,
                                                        self.N2, ptr(param), ptr(m), ptr(v), float(clip_norm), float(lr), beta1, beta2, eps,
                                                        int(step), ptr(scratch), nb, stream_ptr()), "lpm_factored_clip_adam_q")
            else:
                lib.check(lib._lpm_factored_clip_adam(ptr(self.xt), ptr(self.dyt), self.R, self.N1, self.N2, ptr(param), ptr(m), ptr(v),
                                                      float(clip_norm), float(lr), beta1, beta2, eps, int(step), ptr(scratch), nb,
                                                      stream_ptr()), "lpm_factored_clip_adam")
        return scratch


def _factored_clip_adam_copy(self, lib, param, m, v, param_bf16, clip_norm, lr, step, beta1, beta2, eps, scratch, dx=None):
    nb = lib._lpm_factored_clip_adam_scratch_bytes(self.N1, self.N2)
    if scratch is None or scratch.numel() * 4 < nb:
        scratch = torch.empty(nb // 4, dtype=torch.float32, device=param.device)
    if param_bf16.dtype != torch.bfloat16 or param_bf16.numel() != self.N1 * self.N2 or not param_bf16.is_contiguous():
        raise LpmError("FactoredGradient.clip_adam: the compute copy must be a contiguous bf16 tensor of the weight's size")
    with _timed("factored_clip_adam", (self.R, self.N1, self.N2)):
        x = ldx = gdt = None
        quad = (FACTORED_NORM_QUADFORM and self.x is not None and self.x.shape[0] == self.R and self.R <= 128 and self.x.stride(1) == 1
                and self.x.dtype == torch.float32)
        tiles_only = (FACTORED_NORM_QUADFORM and self.x is None and self.x_in_tiles and self.dy is not None and self.dy.shape[0] == self.R
                      and self.R <= 128 and os.environ.get("LPM_FQ_TILES", "1") != "0")
        if tiles_only or quad:
            G = torch.mm(self.dy, self.dy.t())
            gdt = _tile_buffer(lib._lpm_row_tiles_bytes(1, self.R, self.R), G)
            lib.check(lib._lpm_split_rows_tiles(ptr(G), self.R, 1, self.R, self.R, ptr(gdt), stream_ptr()), "lpm_split_rows_tiles")
            x, ldx = (self.xt, self.N1) if tiles_only else (self.x, self.x.stride(0))
        if dx is not None:
            if self.dy is None or tuple(dx.shape) != (self.R, self.N1) or dx.dtype != torch.float32 or dx.stride(1) != 1:
                raise LpmError("FactoredGradient.clip_adam: dx must be an fp32 [R, N1] matrix and this rank's DY must be at hand")
            dy16 = self.dy.to(torch.bfloat16).contiguous()               # round to nearest even: the hi plane of DY's split
            lib.check(lib._lpm_factored_clip_adam_copy_dx(ptr(self.xt), ptr(self.dyt), ptr(x), ldx or 0, ptr(gdt), self.R, self.N1, self.N2,
                                                          ptr(param), ptr(m), ptr(v), ptr(param_bf16), ptr(dy16), ptr(dx), dx.stride(0),
                                                          float(clip_norm), float(lr), beta1, beta2, eps, int(step), ptr(scratch), nb,
                                                          stream_ptr()), "lpm_factored_clip_adam_copy_dx")
            return scratch
        lib.check(lib._lpm_factored_clip_adam_copy(ptr(self.xt), ptr(self.dyt), ptr(x), ldx or 0, ptr(gdt), self.R, self.N1, self.N2,
                                                   ptr(param), ptr(m), ptr(v), ptr(param_bf16), float(clip_norm), float(lr), beta1, beta2, eps,
                                                   int(step), ptr(scratch), nb, stream_ptr()), "lpm_factored_clip_adam_copy")
    return scratch


FactoredGradient._clip_adam_copy = _factored_clip_adam_copy


def clip_adam_step(param, grad, m, v, offsets, ntensors, clip_norm, lr, step, beta1=0.9, beta2=0.999, eps=1e-8,
                   scratch: Optional[torch.Tensor] = None, l2: Optional[torch.Tensor] = None):
    """l2 [ntensors] (device, fp32; optional): per-variable L2-penalty coefficients whose gradient, coefficient * w, the two passes add on
    the fly (lpm_multi_tensor_clip_adam_l2) -- the caller then does NOT add it to ``grad`` itself."""
    lib = _capi.load()
    total = param.numel()
    if scratch is None:
        scratch = torch.empty(lib._lpm_clip_adam_scratch_bytes(total, ntensors) // 4, dtype=torch.float32, device=param.device)
    with _timed("clip_adam", (total, ntensors)):
        if l2 is not None:
            if l2.dtype != torch.float32 or l2.numel() != ntensors or not l2.is_contiguous():
                raise LpmError("clip_adam_step: l2 must be a contiguous fp32 vector with one coefficient per variable")
            lib.check(lib._lpm_multi_tensor_clip_adam_l2(ptr(param), ptr(grad), ptr(m), ptr(v), ptr(offsets), ptr(l2), ntensors, total,
                                                         float(clip_norm), float(lr), beta1, beta2, eps, int(step), ptr(scratch),
                                                         stream_ptr()), "lpm_multi_tensor_clip_adam_l2")
        else:
            lib.check(lib._lpm_multi_tensor_clip_adam(ptr(param), ptr(grad), ptr(m), ptr(v), ptr(offsets), ntensors, total,
                                                      float(clip_norm), float(lr), beta1, beta2, eps, int(step), ptr(scratch),
                                                      stream_ptr()), "lpm_multi_tensor_clip_adam")
    return scratch
