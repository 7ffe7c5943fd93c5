"""Every split-bf16 library GEMM of the cfg-2 video encoder (M = 20480 tokens, F = 1024, filter 4096): time and rate of
the form the step uses, plus the transposed formulation (C^T = B^T A^T) for comparison."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")
M, F, H = 20480, 1024, 4096


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


def bf(*s):
    return torch.randn(*s, device=dev).bfloat16()


def mm(a, b):
    return torch.mm(a, b, out_dtype=torch.float32)


cases = []
# forward / dx:  [M, 3K] x [3K, N]
for name, K, N in (("qkv fwd", F, 3 * F), ("o fwd / o dx", F, F), ("ffn1 fwd / ffn2 dx", F, H), ("ffn2 fwd / ffn1 dx", H, F), ("qkv dx", 3 * F, F)):
    a, b = bf(M, 3 * K), bf(3 * K, N)
    bt = b.t().contiguous()
    at = a.t().contiguous()
    fl = 2.0 * M * 3 * K * N
    cases.append((f"{name:22s} [M,{3*K}]x[{3*K},{N}] NN", lambda a=a, b=b: mm(a, b), fl))
    cases.append((f"{name:22s} same, B given as [N,3K]^T", lambda a=a, bt=bt: mm(a, bt.t()), fl))
    cases.append((f"{name:22s} transposed problem", lambda at=at, bt=bt: mm(bt, at), fl))
# dW: [K, 3M] x [3M, N]
for name, K, N in (("qkv dW", F, 3 * F), ("o dW", F, F), ("ffn1 dW", F, H), ("ffn2 dW", H, F)):
    x3, dy3 = bf(3 * M, K), bf(3 * M, N)
    fl = 2.0 * 3 * M * K * N
    cases.append((f"{name:22s} [{K},3M]x[3M,{N}] TN", lambda x3=x3, dy3=dy3: mm(x3.t(), dy3), fl))
    cases.append((f"{name:22s} transposed: [{N},3M]x[3M,{K}]", lambda x3=x3, dy3=dy3: mm(dy3.t(), x3), fl))
for name, fn, fl in cases:
    t = timeit(fn)
    print(f"{name:60s} {t:8.1f} us  {fl / t / 1e6:7.0f} TF/s executed")
