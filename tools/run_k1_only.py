"""Time the soft-assignment GEMM (K1) and its backward alone at cfg-2's shape: tile (bf16x3) and fp32 forms, HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import _capi
from learnablepoolingmethods_amd._capi import ptr, stream_ptr

lib = _capi.load()
dev = torch.device("cuda:0")
B, T, D, K, ld = 80, 300, 1024, 256, 1152
M = B * T
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
full = torch.randn(M, ld, device=dev)
x = full[:, :D]
W = torch.randn(D, K, device=dev) / 32
dl = torch.randn(M, K, device=dev)
dx = torch.zeros(M, D, device=dev)
dW = torch.empty(D, K, device=dev)
logits = torch.empty(M, K, device=dev)
st = stream_ptr()


def buf(n):
    return torch.empty(n // 4, dtype=torch.int32, device=dev)


xr, wt = buf(lib._lpm_row_tiles_bytes(B, T, D)), buf(lib._lpm_weight_tiles_bytes(D, K))
dlr, wtt = buf(lib._lpm_row_tiles_bytes(B, T, K)), buf(lib._lpm_weight_tiles_bytes(K, D))
xt, dlt = buf(lib._lpm_xt_bytes(B, T, D)), buf(lib._lpm_xt_bytes(B, T, K))
wsb = lib._lpm_assign_gemm_tiles_bwd_dw_workspace_bytes(B, T, D, K)
ws = buf(wsb)
p1 = torch.empty(lib._lpm_assign_gemm_tiles_nblk(B, T), 2, K, device=dev)
p0 = torch.empty(lib._lpm_assign_gemm_nblk(M), 2, K, device=dev)


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


fl = 2.0 * M * D * K
rows = [
    ("split_rows_tiles(x)", lambda: lib._lpm_split_rows_tiles(ptr(x), x.stride(0), B, T, D, ptr(xr), st)),
    ("split_weight_tiles(W)", lambda: lib._lpm_split_weight_tiles(ptr(W), D, K, 0, ptr(wt), st)),
    ("assign_gemm_tiles_fwd", lambda: lib._lpm_assign_gemm_tiles_fwd(ptr(xr), ptr(wt), B, T, D, K, ptr(logits), ptr(p1), st)),
    ("assign_gemm_fwd (fp32 MFMA)", lambda: lib._lpm_assign_gemm_fwd(ptr(x), x.stride(0), ptr(W), M, D, K, 0, ptr(logits), ptr(p0), st)),
    ("split_frames(x)", lambda: lib._lpm_split_frames(ptr(x), x.stride(0), B, T, D, ptr(xt), st)),
    ("split_frames(dl)", lambda: lib._lpm_split_frames(ptr(dl), K, B, T, K, ptr(dlt), st)),
    ("split_rows_tiles(dl)", lambda: lib._lpm_split_rows_tiles(ptr(dl), K, B, T, K, ptr(dlr), st)),
    ("split_weight_tiles(W^T)", lambda: lib._lpm_split_weight_tiles(ptr(W), K, D, 1, ptr(wtt), st)),
    ("assign_gemm_tiles_bwd_dw", lambda: lib._lpm_assign_gemm_tiles_bwd_dw(ptr(xt), ptr(dlt), B, T, D, K, ptr(dW), ptr(ws), wsb, st)),
    ("assign_gemm_tiles_bwd_dx", lambda: lib._lpm_assign_gemm_tiles_bwd_dx(ptr(dlr), ptr(wtt), B, T, D, K, ptr(dx), D, st)),
    ("torch dW = x^T dl (fp32)", lambda: x.t().matmul(dl)),
    ("torch dx += dl W^T (fp32)", lambda: dx.addmm_(dl, W.t())),
]
for name, fn in rows:
    t = timeit(fn)
    print(f"{name:32s} {t:8.1f} us   ({fl / t / 1e6:7.1f} TF if a full GEMM)")
