"""FeedForwardNetwork's first dense layer and its backward at cfg-2's video shape (M = 20480, F = 1024, H = 4096): the fused tile-GEMM
route (operand images from the epilogue) against the library route (GEMM + separate split pass), every launch timed on its own."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import _capi, ops
from learnablepoolingmethods_amd.ops import ptr, stream_ptr as st

dev = torch.device("cuda:0")
lib = _capi.load()
M, F, H = 20480, 1024, 4096
g = torch.Generator(device=dev).manual_seed(0)
y = torch.randn(M, F, device=dev, generator=g)
W1 = torch.randn(F, H, device=dev, generator=g) / F ** 0.5
b1 = 0.3 * torch.randn(H, device=dev, generator=g)
W2 = torch.randn(H, F, device=dev, generator=g) / H ** 0.5
do = torch.randn(M, F, device=dev, generator=g)


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


y3 = ops._split_rows(y)
w13n, w13k = ops._split_weight(W1)
w23n, w23k = ops._split_weight(W2)
do3 = ops._split_rows(do, grad=True)
yr = ops._tile_buffer(lib._lpm_row_tiles_bytes(1, M, F), y)
w1t = ops._tile_buffer(lib._lpm_weight_tiles_bytes(F, H), y)
w2tt = ops._tile_buffer(lib._lpm_weight_tiles_bytes(F, H), y)
dor = ops._tile_buffer(lib._lpm_row_tiles_bytes(1, M, F), y)
f3 = torch.empty(M, 3 * H, dtype=torch.bfloat16, device=dev)
dp3 = torch.empty(M, 3 * H, dtype=torch.bfloat16, device=dev)
db1 = torch.empty(H, device=dev)
wsb = lib._lpm_dense_tiles_relu_bwd_workspace_bytes(M, H)
ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
pre = torch.empty(M, H, device=dev)
wsb2 = lib._lpm_split_rows_relu_bwd_workspace_bytes(M, H)
ws2 = torch.empty(wsb2 // 4, dtype=torch.float32, device=dev)
rows = [
    ("fwd  tiles: split_rows_tiles(y)", lambda: lib.check(lib._lpm_split_rows_tiles(ptr(y), F, 1, M, F, ptr(yr), st()), "a")),
    ("fwd  tiles: split_weight_tiles(W1)", lambda: lib.check(lib._lpm_split_weight_tiles(ptr(W1), F, H, 0, ptr(w1t), st()), "b")),
    ("fwd  tiles: GEMM + bias/relu/image epilogue", lambda: lib.check(lib._lpm_dense_tiles_act_image_fwd(ptr(yr), ptr(w1t), ptr(b1), M, F, H, ptr(f3), st()), "c")),
    ("fwd  tiles: plain fp32 store (form 4)", lambda: lib.check(lib._lpm_dense_tiles_fwd(ptr(yr), ptr(w1t), M, F, H, ptr(pre), H, 4, st()), "d")),
    ("fwd  library: GEMM", lambda: ops._mm3(y3, w13n)),
    ("fwd  library: split_rows(pre1, bias, relu)", lambda: ops._split_rows(pre, bias=b1, relu=True)),
    ("bwd  tiles: image_row_tiles(do3)", lambda: lib.check(lib._lpm_image_row_tiles(ptr(do3), M, F, 1, ptr(dor), st()), "e")),
    ("bwd  tiles: split_weight_tiles(W2^T)", lambda: lib.check(lib._lpm_split_weight_tiles(ptr(W2), F, H, 1, ptr(w2tt), st()), "f")),
    ("bwd  tiles: GEMM + mask/dbias/image epilogue", lambda: lib.check(lib._lpm_dense_tiles_relu_bwd_image(ptr(dor), ptr(w2tt), ptr(f3), M, F, H, ptr(dp3), ptr(db1), ptr(ws), wsb, st()), "g")),
    ("bwd  library: GEMM", lambda: ops._mm3(do3, w23k)),
    ("bwd  library: split_rows_relu_bwd", lambda: lib.check(lib._lpm_split_rows_relu_bwd(ptr(pre), M, H, ptr(f3), ptr(dp3), ptr(db1), ptr(ws2), wsb2, st()), "h")),
]
lib.check(lib._lpm_split_rows_tiles(ptr(y), F, 1, M, F, ptr(yr), st()), "a")
lib.check(lib._lpm_split_weight_tiles(ptr(W1), F, H, 0, ptr(w1t), st()), "b")
lib.check(lib._lpm_image_row_tiles(ptr(do3), M, F, 1, ptr(dor), st()), "e")
lib.check(lib._lpm_split_weight_tiles(ptr(W2), F, H, 1, ptr(w2tt), st()), "f")
lib.check(lib._lpm_dense_tiles_act_image_fwd(ptr(yr), ptr(w1t), ptr(b1), M, F, H, ptr(f3), st()), "c")
for name, fn in rows:
    print(f"{name:48s} {timeit(fn):8.1f} us", flush=True)
