"""K1 on plain bf16 tiles alone at BASELINE configs[4]'s video shape (128 x 300 frames, 1024 -> 512): HIP-event time per launch.
usage: python tools/k1_bf16_loop.py [launches]   (LPM_K1_WIDE=0: the flat 96-row form; LPM_K1_WIDE_DBG=n: timing experiments)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import _capi, ops
from learnablepoolingmethods_amd._capi import ptr, stream_ptr

lib = _capi.load()
dev = torch.device("cuda:0")
B, T, D, K = 128, 300, 1024, 512
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
raw = torch.randn(B, T, 1152, device=dev)
nf = torch.full((B,), T, dtype=torch.int32, device=dev)
y = ops.frame_sample_bn(raw, nf, T, storage="bf16", materialize=False)
xr = ops._cached_tiles(y[:, :D], B, T, D, rows=True, storage="bf16")
W = torch.randn(D, K, device=dev) / 32
wt = torch.empty(lib._lpm_weight_tiles_bytes(D, K) // 8, dtype=torch.int32, device=dev)
st = stream_ptr()
lib.check(lib._lpm_split_weight_tiles_bf16(ptr(W), D, K, 0, ptr(wt), st), "w")
nblk = lib._lpm_assign_gemm_tiles_nblk(B, T)
logits = torch.empty(B * T, K, dtype=torch.bfloat16, device=dev)
partial = torch.empty(nblk, 2, K, device=dev)
fn = lambda: lib.check(lib._lpm_assign_gemm_tiles_fwd_bf16(ptr(xr), ptr(wt), B, T, D, K, ptr(logits), ptr(partial), st), "k1")
for _ in range(5):
    fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    fn()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(f"K1 bf16 {B}x{T} {D}->{K}  WIDE={os.environ.get('LPM_K1_WIDE', '1')} DBG={os.environ.get('LPM_K1_WIDE_DBG', '0')}: {us:.1f} us / launch back to back = "
      f"{2.0 * B * T * D * K / us * 1e-6:.0f} TFLOP/s = {2.0 * B * T * D * K / us * 1e-6 / 2500:.3f} of the bf16 peak")
if "--clock" in sys.argv:
    import numpy as np
    NS, PERIOD = 60000, 100
    samples = torch.zeros(2 * NS, dtype=torch.int64, device=dev)
    marks = torch.zeros(2 * 4, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(side):
        lib.check(lib._lpm_clock_sampler(ptr(samples), NS, PERIOD, side.cuda_stream), "lpm_clock_sampler")
    torch.cuda._sleep(int(2.0e6 * 2))
    lib.check(lib._lpm_clock_marker(ptr(marks), 0, st), "lpm_clock_marker")
    for _ in range(max(n, 300)):
        fn()
    lib.check(lib._lpm_clock_marker(ptr(marks), 1, st), "lpm_clock_marker")
    torch.cuda.synchronize()
    s = samples.cpu().numpy().reshape(-1, 2).astype(np.int64)
    m = marks.cpu().numpy().reshape(-1, 2).astype(np.int64)
    s = s[s[:, 0] > 0]
    t = s[:, 0] / 100.0
    f = np.diff(s[:, 1]) / np.maximum(np.diff(s[:, 0]), 1) * 100.0
    mid = (t[:-1] + t[1:]) / 2
    a, b = m[0, 0] / 100.0, m[1, 0] / 100.0
    inside = (mid > a + 0.2 * (b - a)) & (mid < b)
    before = (mid < a) & (mid > a - 1500)
    print(f"   shader clock before {np.nanmean(f[before]) if before.any() else float('nan'):6.0f} MHz, inside the train {np.nanmean(f[inside]):6.0f} MHz "
          f"(min {np.nanmin(f[inside]):6.0f}, {inside.sum()} samples); {(b - a) / max(n, 300):.1f} us per launch in the train")
