"""hidden1 weight gradient dW[270336,512] = x^T dy as a tile GEMM (weight-tile operands) vs the fp32 library GEMM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import ops

dev = torch.device("cuda:0")
B, KV, H = 80, 270336, 512
x, dy = torch.randn(B, KV, device=dev), torch.randn(B, H, device=dev)
out = torch.empty(KV, H, device=dev)


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


ref = torch.mm(x.t(), dy)
got = ops.skinny_weight_grad(x, dy, out=out)
print("rel err", float((got - ref).abs().max() / ref.abs().max()))
from learnablepoolingmethods_amd import _capi
lib = _capi.load()
from learnablepoolingmethods_amd._capi import ptr, stream_ptr
xt = torch.empty(lib._lpm_weight_tiles_bytes(B, KV) // 4, dtype=torch.int32, device=dev)
print("split x: %.1f us" % timeit(lambda: lib._lpm_split_weight_tiles(ptr(x), B, KV, 0, ptr(xt), stream_ptr())))
print("tile GEMM dW %.1f us | torch fp32 %.1f us" % (timeit(lambda: ops.skinny_weight_grad(x, dy, out=out)),
                                                       timeit(lambda: torch.mm(x.t(), dy, out=out))))
