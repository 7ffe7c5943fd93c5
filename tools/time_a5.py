"""The whole a5 function (frame_level_models.py:2798-2822: BN-affine + softmax -> residual aggregation -> both normalisations) at a
BASELINE shape, as the chain of launches the product runs for it: lpm_assign_tiles + K2 (+ finalize, two-pass form), timed with
events over a loop; fused vs two-pass, training (U stored for the backward) vs inference.
  python tools/time_a5.py [B T D K kmajor]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import ops, _capi
from learnablepoolingmethods_amd._capi import LPM_VLAD_SOFTMAX, LPM_VLAD_RESIDUAL
dev = torch.device("cuda:0")
a = [int(v) for v in sys.argv[1:]]
B, T, D, K = (a + [80, 300, 1024, 256])[:4] if len(a) >= 4 else (80, 300, 1024, 256)
kmajor = bool(a[4]) if len(a) > 4 else True
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B * T, D, device=dev, generator=g)
logits = torch.randn(B * T, K, device=dev, generator=g)
scale = 1 + 0.1 * torch.randn(K, device=dev, generator=g)
shift = 0.1 * torch.randn(K, device=dev, generator=g)
cen = torch.randn(D, K, device=dev, generator=g) / 32
lib = _capi.load()
flags = LPM_VLAD_SOFTMAX | LPM_VLAD_RESIDUAL
alg = 4 * (B * T * K + B * T * D + D * K + B * D * K)
xt = torch.empty(lib._lpm_xt_bytes(B, T, D) // 4, dtype=torch.int32, device=dev)
lib.check(lib._lpm_split_frames(_capi.ptr(x), x.stride(0), B, T, D, _capi.ptr(xt), _capi.stream_ptr()), "split")
ops._XT_CACHE.clear()
orig = ops._cached_tiles
ops._cached_tiles = lambda *_: xt          # the frame tiles exist already (frame_sample_bn writes them in the step)
for fused in (True, False):
    for save_u in (True, False):
        ops.VLAD_FUSED = fused
        def once():
            return ops._aggregate_fwd(lib, logits, scale, shift, x, cen, B, T, D, K, flags, kmajor, nrm_raw=True, save_u=save_u)
        for _ in range(5):
            once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            once()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"a5 B={B} T={T} D={D} K={K} kmajor={kmajor} fused={fused} store_u={save_u}: {us:.1f} us per call  "
              f"({alg / 1e6:.1f} MB algorithmic -> {alg / us / 1e6:.2f} TB/s = {alg / us / 1e6 / 8:.3f} of 8 TB/s)")
