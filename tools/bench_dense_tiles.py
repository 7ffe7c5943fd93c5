"""The encoder's dense layers (cfg-2 video encoder: M = 20480 tokens, F = 1024, filter 4096) on the hand-written split-bf16 tile GEMM
(lpm_dense_tiles_fwd: row tiles x weight tiles, 3 MFMAs per product) against the library route the step uses today (hipBLASLt bf16 GEMM
over the [hi|lo|hi] x [Wh;Wh;Wl] images).  Prints time, executed TFLOP/s (3 x useful) and the max error of both against fp64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import _capi, ops

dev = torch.device("cuda:0")
lib = _capi.load()
M, F, H = 20480, 1024, 4096
ptr, st = ops.ptr, ops.stream_ptr


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


forms = [int(a) for a in sys.argv[1:]] or [2, 4]
for name, K, N in (("qkv fwd", F, 3 * F), ("o fwd", F, F), ("ffn1 fwd", F, H), ("ffn2 fwd", H, F), ("qkv dx", 3 * F, F)):
    x = torch.randn(M, K, device=dev)
    W = torch.randn(K, N, device=dev) / K ** 0.5
    xr = torch.empty(lib._lpm_row_tiles_bytes(1, M, K) // 4, dtype=torch.int32, device=dev)
    wt = torch.empty(lib._lpm_weight_tiles_bytes(K, N) // 4, dtype=torch.int32, device=dev)
    lib.check(lib._lpm_split_rows_tiles(ptr(x), K, 1, M, K, ptr(xr), st()), "rows")
    lib.check(lib._lpm_split_weight_tiles(ptr(W), K, N, 0, ptr(wt), st()), "weight")
    y = torch.empty(M, N, device=dev)
    x3 = ops._split_rows(x)
    w3n, _ = ops._split_weight(W, need_t=False)
    ref = (x[:512].double() @ W.double())
    fl = 3 * 2.0 * M * K * N
    t_lib = timeit(lambda: ops._mm3(x3, w3n))
    e_lib = float((ops._mm3(x3, w3n)[:512].double() - ref).abs().max() / ref.abs().max())
    line = f"{name:9s} M={M} K={K} N={N}: library {t_lib:7.1f} us {fl / t_lib / 1e6:6.0f} TF/s (err {e_lib:.1e})"
    for form in forms:
        def run():
            lib.check(lib._lpm_dense_tiles_fwd(ptr(xr), ptr(wt), M, K, N, ptr(y), N, form, st()), "dense")
        t = timeit(run)
        e = float((y[:512].double() - ref).abs().max() / ref.abs().max())
        line += f" | tiles form {form}: {t:7.1f} us {fl / t / 1e6:6.0f} TF/s (err {e:.1e})"
    t_sr = timeit(lambda: ops._split_rows(x))
    t_st = timeit(lambda: lib.check(lib._lpm_split_rows_tiles(ptr(x), K, 1, M, K, ptr(xr), st()), "rows"))
    print(line + f" | operand split: image {t_sr:.1f} us, tiles {t_st:.1f} us", flush=True)
