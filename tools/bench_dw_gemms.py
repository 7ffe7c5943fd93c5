"""Weight-gradient GEMM forms for the cfg-2 video encoder: single long-reduction GEMM, its transpose, three separate
plane GEMMs, and split-K as a batched GEMM + sum."""
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


for name, K, N in (("qkv dW", F, 3 * F), ("o dW", F, F), ("ffn1 dW", F, H), ("ffn2 dW", H, F)):
    x3, dy3 = bf(M, 3 * K), bf(M, 3 * N)
    xv, dv = x3.view(3 * M, K), dy3.view(3 * M, N)
    fl = 2.0 * 3 * M * K * N
    forms = [("single TN", lambda: mm(xv.t(), dv)), ("single, transposed", lambda: mm(dv.t(), xv))]

    def sep():
        o = mm(x3[:, :K].t(), dy3[:, :N])
        o += mm(x3[:, K:2 * K].t(), dy3[:, N:2 * N])
        o += mm(x3[:, 2 * K:].t(), dy3[:, 2 * N:])
        return o
    forms.append(("3 separate", sep))
    for S in (2, 3, 4, 6, 8):
        if (3 * M) % S:
            continue
        xb, db = xv.view(S, 3 * M // S, K), dv.view(S, 3 * M // S, N)
        forms.append((f"bmm split-K {S}", lambda xb=xb, db=db: torch.bmm(xb.transpose(1, 2), db, out_dtype=torch.float32).sum(0)))
        forms.append((f"bmm split-K {S} transposed", lambda xb=xb, db=db: torch.bmm(db.transpose(1, 2), xb, out_dtype=torch.float32).sum(0)))
    for fname, fn in forms:
        try:
            t = timeit(fn)
            print(f"{name:8s} {fname:28s} {t:8.1f} us  {fl / t / 1e6:7.0f} TF/s")
        except Exception as e:
            print(f"{name:8s} {fname:28s} failed: {str(e)[:80]}")
