"""The shader clock WHILE K1 runs: lpm_clock_sampler (one wave on a stream of its own, a sample of the constant 100 MHz counter and the shader-clock
counter every microsecond) beside a train of back-to-back K1 launches at cfg-2's shape, markers in the same time base around the train.
Prints the mean shader clock before / inside / after the train for random and for all-zero operands (the matrix pipe's power depends on the data).
  python tools/k1_clock.py [launches, default 300]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from learnablepoolingmethods_amd import _capi
from learnablepoolingmethods_amd._capi import ptr, stream_ptr

lib = _capi.load()
dev = torch.device("cuda:0")
B, T, D, K = 80, 300, 1024, 256
M = B * T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for zero in (False, True):
    x = torch.zeros(M, D, device=dev) if zero else torch.randn(M, D, device=dev)
    W = torch.zeros(D, K, device=dev) if zero else torch.randn(D, K, device=dev) / 32
    logits = torch.empty(M, K, device=dev)
    st = stream_ptr()
    xr = torch.empty(lib._lpm_row_tiles_bytes(B, T, D) // 4, dtype=torch.int32, device=dev)
    wt = torch.empty(lib._lpm_weight_tiles_bytes(D, K) // 4, dtype=torch.int32, device=dev)
    p1 = torch.empty(lib._lpm_assign_gemm_tiles_nblk(B, T), 2, K, device=dev)
    lib._lpm_split_rows_tiles(ptr(x), x.stride(0), B, T, D, ptr(xr), st)
    lib._lpm_split_weight_tiles(ptr(W), D, K, 0, ptr(wt), st)
    for _ in range(20):
        lib.check(lib._lpm_assign_gemm_tiles_fwd(ptr(xr), ptr(wt), B, T, D, K, ptr(logits), ptr(p1), st), "k1")
    torch.cuda.synchronize()
    NS, PERIOD = 60000, 100                      # 1 us per sample, 60 ms
    samples = torch.zeros(2 * NS, dtype=torch.int64, device=dev)
    marks = torch.zeros(2 * 4, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(side):
        lib.check(lib._lpm_clock_sampler(ptr(samples), NS, PERIOD, side.cuda_stream), "lpm_clock_sampler")
    torch.cuda._sleep(int(2.0e6 * 2))            # ~2 ms of idle in front of the train
    lib.check(lib._lpm_clock_marker(ptr(marks), 0, st), "lpm_clock_marker")
    for _ in range(n):
        lib.check(lib._lpm_assign_gemm_tiles_fwd(ptr(xr), ptr(wt), B, T, D, K, ptr(logits), ptr(p1), st), "k1")
    lib.check(lib._lpm_clock_marker(ptr(marks), 1, st), "lpm_clock_marker")
    torch.cuda.synchronize()
    s = samples.cpu().numpy().reshape(-1, 2).astype(np.int64)
    m = marks.cpu().numpy().reshape(-1, 2).astype(np.int64)
    s = s[s[:, 0] > 0]
    t = s[:, 0] / 100.0
    f = np.diff(s[:, 1]) / np.maximum(np.diff(s[:, 0]), 1) * 100.0
    mid = (t[:-1] + t[1:]) / 2
    a, b = m[0, 0] / 100.0, m[1, 0] / 100.0
    inside = (mid > a + 0.2 * (b - a)) & (mid < b)          # (the first fifth: the clock is still on its way down)
    before = (mid < a) & (mid > a - 1500)
    print(f"{'all-zero' if zero else 'random  '} operands: {n} launches in {b - a:8.1f} us = {(b - a) / n:6.2f} us each; shader clock "
          f"before {np.nanmean(f[before]) if before.any() else float('nan'):6.0f} MHz, inside the train {np.nanmean(f[inside]):6.0f} MHz "
          f"(min {np.nanmin(f[inside]):6.0f}, {inside.sum()} samples; sampling gaps > 5 us: {(np.diff(t) > 5).sum()})")
