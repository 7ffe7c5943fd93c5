"""The projection passes on the bf16 compute copy alone (lpm_proj_fwd_parts_w16, lpm_proj_dx_w16): time and TB/s of weight for
cfg-5's shape (128 clips, Kd = 540 672, N = 1 024) and for the same bytes as N = 512 (Kd doubled: one column block, whole rows per piece).
  python tools/time_proj_w16.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import _capi
from learnablepoolingmethods_amd._capi import ptr, stream_ptr

lib = _capi.load()
dev = torch.device("cuda:0")
M = 128


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for Kd, N in ((540672, 1024), (1081344, 512)):
    n1a, ks = Kd - 16384, 512
    x1 = torch.randn(M, n1a, device=dev).to(torch.bfloat16)
    scale = torch.rand(M, ks, device=dev) + 0.5
    x2 = torch.randn(M, Kd - n1a, device=dev)
    W16 = (torch.randn(Kd, N, device=dev) / 100).to(torch.bfloat16)
    y = torch.empty(M, N, device=dev)
    wsb = lib._lpm_proj_fwd_workspace_bytes(M, Kd, N)
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
    st = stream_ptr()
    dy = torch.randn(M, N, device=dev)
    dyt = torch.empty(lib._lpm_row_tiles_bytes(1, M, N) // 4, dtype=torch.int32, device=dev)
    lib.check(lib._lpm_split_rows_tiles(ptr(dy), N, 1, M, N, ptr(dyt), st), "tiles")
    dx = torch.empty(M, Kd, device=dev)
    gb = Kd * N * 2 / 1e9
    t = timeit(lambda: lib.check(lib._lpm_proj_fwd_parts_w16(ptr(x1), n1a, n1a, ptr(scale), ks, ptr(x2), x2.stride(0), ptr(W16), M, Kd, N, ptr(y),
                                                           ptr(ws), wsb, st), "fwd"))
    print(f"Kd={Kd} N={N}: forward {t:7.1f} us = {gb / t * 1e3:5.2f} TB/s of weight")
    t = timeit(lambda: lib.check(lib._lpm_proj_dx_w16(ptr(dyt), ptr(W16), M, Kd, N, ptr(dx), Kd, st), "dx"))
    print(f"Kd={Kd} N={N}: dx      {t:7.1f} us = {gb / t * 1e3:5.2f} TB/s of weight")
    t = timeit(lambda: W16.float())          # a plain streaming pass over the same bytes (reads 2 B, writes 4 B per weight)
    print(f"Kd={Kd} N={N}: torch bf16 -> fp32 copy {t:7.1f} us = {3 * gb / t * 1e3:5.2f} TB/s total")
    del x1, x2, W16, dx
