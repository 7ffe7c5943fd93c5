"""Where a NaN of ops.projection_parts' weight gradient comes from (case 16 x 128 x 128, no second block)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from learnablepoolingmethods_amd import ops, _capi
from learnablepoolingmethods_amd._capi import ptr, stream_ptr
dev = torch.device("cuda:0")
B, D, K, NA, H = 16, 128, 128, 0, 512
for trial in range(3):
    g = torch.Generator().manual_seed(B + K + NA)
    raw = (torch.randn(B, D * K, generator=g) * 3).to(dev)
    scale = (torch.rand(B, K, generator=g) + 0.5).to(dev) / (D * K) ** 0.5
    Kd = D * K
    W = (torch.randn(Kd, H, generator=g) / Kd ** 0.5).to(dev).requires_grad_(True)
    dy = torch.randn(B, H, generator=g).to(dev)
    x1 = raw.clone().requires_grad_(True)
    x1._lpm_row_scale, x1._lpm_scale_ks = scale, K
    view = torch.full((Kd, H), float("nan"), device=dev)
    W._lpm_grad_view = view
    y = ops.projection_parts(x1, None, W)
    y.backward(dy)
    torch.cuda.synchronize()
    bad = torch.isnan(view)
    print("trial", trial, "nan count", int(bad.sum()), "rows", bad.any(1).nonzero().flatten()[:8].tolist(), "cols", bad.any(0).nonzero().flatten()[:8].tolist(),
          "y nan", int(torch.isnan(y).sum()), "dx nan", int(torch.isnan(x1.grad).sum()))
    lib = _capi.load()
    xt = torch.empty(lib._lpm_weight_tiles_bytes(B, Kd) // 4, dtype=torch.int32, device=dev)
    lib.check(lib._lpm_split_weight_tiles_parts(ptr(x1.detach()), Kd, Kd, 0, ptr(scale), K, None, 0, B, Kd, ptr(xt), stream_ptr()), "parts")
    xm = (raw.view(B, D, K) * scale.unsqueeze(1)).reshape(B, Kd).contiguous()
    xt2 = torch.empty_like(xt)
    lib.check(lib._lpm_split_weight_tiles(ptr(xm), B, Kd, 0, ptr(xt2), stream_ptr()), "plain")
    torch.cuda.synchronize()
    print("   tiles differ in", int((xt != xt2).sum()), "of", xt.numel(), "words")

lib = _capi.load()
for (R, N1, N2) in [(16, 16384, 512), (16, 4224, 64), (16, 4224, 512), (32, 16384, 512), (16, 16384, 128)]:
    x = torch.randn(R, N1, device=dev); dyy = torch.randn(R, N2, device=dev)
    out = torch.full((N1, N2), float("nan"), device=dev)
    ops.skinny_weight_grad(x, dyy, out=out)
    torch.cuda.synchronize()
    print((R, N1, N2), "nan", int(torch.isnan(out).sum()), "err", float((out - x.t() @ dyy).abs().max()))
