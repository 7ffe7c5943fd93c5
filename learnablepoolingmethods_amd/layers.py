"""TF1 / slim layer semantics the hot path depends on, as thin torch code.
These are the non-hot ops (dense GEMMs go to hipBLASLt through torch.matmul; small [B,H]-sized batch
norms and layer norms are elementwise plumbing).  The hot ops live in ops.py / csrc/."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import FLAGS, ops
from . import variables as vs

BN_EPS = 1e-3
BN_DECAY = 0.999
LN_EPS = 1e-12


def l2_normalize(x: torch.Tensor, dim: int, eps: float = 1e-12) -> torch.Tensor:
    """tf.nn.l2_normalize: x * rsqrt(max(sum x^2, eps)) (train.py:264)."""
    return x * torch.rsqrt(torch.clamp((x * x).sum(dim=dim, keepdim=True), min=eps))


def bn_variables(scope: str, channels: int, device):
    """slim.batch_norm(center=True, scale=True) variables: beta, gamma, moving_mean, moving_variance."""
    with vs.variable_scope(scope):
        beta = vs.get_variable("beta", [channels], vs.zeros_initializer(), device=device)
        gamma = vs.get_variable("gamma", [channels], vs.ones_initializer(), device=device)
        mm = vs.get_variable("moving_mean", [channels], vs.zeros_initializer(), trainable=False, device=device)
        mv = vs.get_variable("moving_variance", [channels], vs.ones_initializer(), trainable=False, device=device)
    return gamma, beta, mm, mv


def batch_norm(x: torch.Tensor, is_training: bool, scope: str, pre_bias: torch.Tensor = None, pre_relu: bool = False) -> torch.Tensor:
    """slim.batch_norm on a tensor whose last axis is the channel (frame_level_models.py:2355;
    transformer_utils.py:666,747,760).  Moving variance gets the unbiased estimate on the fused
    (rank-2/4) path, the biased one otherwise.  pre_bias / pre_relu: x is the raw output of a dense layer whose bias add (and ReLU) the
    caller deferred (dense(..., defer_bias=True)): the normalised tensor is act(x + pre_bias)."""
    C = x.shape[-1]
    gamma, beta, mm, mv = bn_variables(scope, C, x.device)
    if pre_bias is not None:
        if is_training and x.dim() == 3 and ops.batch_norm_rows_act_ok(x, pre_bias):
            return ops.batch_norm_rows_act(x, pre_bias, pre_relu, gamma, beta, mm, mv, biased_moving_variance=True)
        if ops.BIAS_ACT_FUSED and x.is_contiguous() and ops.bias_act_ok(x, pre_bias):
            x = ops.bias_act(x * 1.0 if x._base is not None else x, pre_bias, pre_relu)      # (a view of the GEMM's output: not in place on it)
        else:
            x = x + pre_bias
            if pre_relu:
                x = torch.relu(x)
    if is_training and x.dim() == 2 and x.is_cuda and x.shape[0] > 1:
        # the fused rank-2 path in one library kernel each way: batch statistics, normalisation and the moving-average
        # update (unbiased variance into the moving average, exactly TF's fused batch norm) instead of ~12 small kernels
        return F.batch_norm(x, mm, mv, gamma, beta, True, 1.0 - BN_DECAY, BN_EPS)
    if is_training and x.dim() == 3 and x.is_cuda and C % 4 == 0:
        # the V2 encoder's [B, L, C] tensors (up to 393 MB): hand-written channel-last kernels, 3 launches forward and 3
        # backward instead of ~30 elementwise / reduction passes; rank 3 takes TF's non-fused path (biased moving variance)
        return ops.batch_norm_rows(x, gamma, beta, mm, mv, biased_moving_variance=True)
    if is_training:
        red = tuple(range(x.dim() - 1))
        n = x.numel() // C
        mean = x.mean(dim=red)
        var = ((x - mean) ** 2).mean(dim=red)
        with torch.no_grad():
            uv = var * (n / max(n - 1, 1)) if x.dim() in (2, 4) else var
            mm.mul_(BN_DECAY).add_(mean.detach(), alpha=1 - BN_DECAY)
            mv.mul_(BN_DECAY).add_(uv.detach(), alpha=1 - BN_DECAY)
    else:
        mean, var = mm, mv
    return (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta


def layer_norm_variables(scope: str, features: int, device):
    """tf.contrib.layers.layer_norm's variables in its creation order: beta (zeros), gamma (ones) -> (gamma, beta)."""
    with vs.variable_scope(scope):
        beta = vs.get_variable("beta", [features], vs.zeros_initializer(), device=device)
        gamma = vs.get_variable("gamma", [features], vs.ones_initializer(), device=device)
    return gamma, beta


def layer_norm(x: torch.Tensor, scope: str = "LayerNorm", residual: torch.Tensor = None, bias: torch.Tensor = None,
               relu: bool = False, image: bool = False, mask: torch.Tensor = None, mask_scale: float = 1.0, next_kernel=None) -> torch.Tensor:
    """tf.contrib.layers.layer_norm(act(x + bias) [+ residual]) with TF1 defaults: moments over ALL non-batch axes,
    gamma/beta [last], eps 1e-12 (transformer_utils.py:407,411,454,713).  ``bias`` / ``relu`` are the tail of the dense
    layer that produced x (tf.layers.dense(use_bias=True[, activation=relu])), handed over so that [B,L,F] tensors on the
    GPU take ONE fused bias + activation + residual + layer-norm kernel pair (csrc/layer_norm.hip)."""
    with vs.variable_scope(scope):
        beta = vs.get_variable("beta", [x.shape[-1]], vs.zeros_initializer(), device=x.device)
        gamma = vs.get_variable("gamma", [x.shape[-1]], vs.ones_initializer(), device=x.device)
    if x.is_cuda and x.dim() == 3 and x.shape[-1] in ops.LN_FEATURES:
        # image, mask / mask_scale (a dropout keep mask applied to act(x + bias) before the residual): see ops.residual_layer_norm
        # next_kernel: the kernel of the dense layer that reads the result next -- the operand image is written in that layer's format
        return ops.residual_layer_norm(x, residual, gamma, beta, bias=bias, relu=relu, image=image, mask=mask, mask_scale=mask_scale,
                                       next_kernel=next_kernel)
    if bias is not None:
        x = x + bias
    if relu:
        x = torch.relu(x)
    if mask is not None:
        x = x * mask.to(x.dtype) * mask_scale
    if residual is not None:
        x = x + residual
    y = F.layer_norm(x, tuple(x.shape[1:]), None, None, LN_EPS)
    return y * gamma + beta


def use_split_gemm(x: torch.Tensor, rows: int, units: int) -> bool:
    """Dense layers big enough to pay for the operand split run as split-bf16 library GEMMs (ops._DenseX3)."""
    return (FLAGS.dense_precision == "bf16x3" and x.is_cuda and rows >= 1024 and x.shape[-1] % 8 == 0 and units % 8 == 0)


def dense_variables(name: str, fan_in: int, units: int, use_bias: bool, device):
    """tf.layers.dense variables: <name>/kernel (glorot uniform), <name>/bias (zeros)."""
    with vs.variable_scope(name):
        kernel = vs.get_variable("kernel", [fan_in, units], vs.glorot_uniform_initializer(), device=device)
        bias = vs.get_variable("bias", [units], vs.zeros_initializer(), device=device) if use_bias else None
    return kernel, bias


def qkv_projections(queries: torch.Tensor, keys: torch.Tensor, units: int):
    """The bias-free q / k / v projections of an attention block (transformer_utils.py:559-561).  Self-attention on the
    split-bf16 path runs them as one fused GEMM (ops.qkv_x3); variable names and values are those of three tf.layers.dense."""
    rows = queries.numel() // queries.shape[-1]
    if queries is keys and use_split_gemm(queries, rows, units):
        wq, _ = dense_variables("q", queries.shape[-1], units, False, queries.device)
        wk, _ = dense_variables("k", queries.shape[-1], units, False, queries.device)
        wv, _ = dense_variables("v", queries.shape[-1], units, False, queries.device)
        q, k, v = ops.qkv_x3(queries.reshape(rows, queries.shape[-1]), wq, wk, wv)
        shape = (*queries.shape[:-1], units)
        return q.view(shape), k.view(shape), v.view(shape)
    return (dense(queries, units, use_bias=False, name="q"), dense(keys, units, use_bias=False, name="k"),
            dense(keys, units, use_bias=False, name="v"))


def dense(x: torch.Tensor, units: int, use_bias: bool, name: str, activation=None, defer_bias: bool = False):
    """tf.layers.dense: contracts the last axis; glorot-uniform kernel, zero bias.  defer_bias: return (x W, bias) and leave
    the bias add (and activation) to the caller -- the fused layer_norm that follows takes them."""
    kernel, bias = dense_variables(name, x.shape[-1], units, use_bias, x.device)
    rows = x.numel() // x.shape[-1]
    split = use_split_gemm(x, rows, units)
    y = ops.dense_x3(x.reshape(rows, x.shape[-1]), kernel) if split else x.matmul(kernel)
    if defer_bias:
        return y.reshape(*x.shape[:-1], units), bias
    if use_bias and activation in (None, torch.relu) and ops.BIAS_ACT_FUSED and ops.bias_act_ok(y, bias):
        # one in-place pass over the GEMM's fresh output (one pass in the backward as well) instead of add + relu (threshold + reduce)
        return ops.bias_act(y, bias, activation is torch.relu).reshape(*x.shape[:-1], units)
    y = y.reshape(*x.shape[:-1], units)
    if use_bias:
        y = y + bias
    return activation(y) if activation is not None else y
