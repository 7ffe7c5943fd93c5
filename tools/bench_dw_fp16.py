"""The one-term fp16 weight-gradient GEMMs of the cfg-2 video encoder (dW = xh^T dyh, ops._dw_x2) as the library runs them:
(a) today's form -- the activation's hi plane read in place from its [M, 3K] image, transposed by the GEMM (TN), S slices of the token
reduction; (b) the same from a contiguous [M, K] copy; (c) from a TRANSPOSED copy xT [K, M] (NN); plus the cost of making that copy.
  python tools/bench_dw_fp16.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")
M, F, H = 20480, 1024, 4096


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


tot = {}
for name, K, N in (("qkv dW", F, 3 * F), ("o dW", F, F), ("ffn1 dW", F, H), ("ffn2 dW", H, F)):
    x2 = torch.randn(M, 3 * K, device=dev).half()
    dy = torch.randn(M, 2 * N, device=dev).half()
    fl = 2.0 * M * K * N
    xc = x2[:, :K].contiguous()
    xT = xc.t().contiguous()
    forms = []
    for S in (1, 2, 4, 8):
        xh = x2.view(S, M // S, 3 * K)[:, :, :K]
        dyv = dy.view(S, M // S, 2 * N)[:, :, :N]
        forms.append((f"image TN S={S}", lambda xh=xh, dyv=dyv: torch.bmm(xh.transpose(1, 2), dyv, out_dtype=torch.float32)))
        xcv = xc.view(S, M // S, K)
        forms.append((f"contig TN S={S}", lambda xcv=xcv, dyv=dyv: torch.bmm(xcv.transpose(1, 2), dyv, out_dtype=torch.float32)))
        xTv = xT.view(K, S, M // S).transpose(0, 1)               # [S, K, M/S], row stride M
        forms.append((f"xT NN S={S}", lambda xTv=xTv, dyv=dyv: torch.bmm(xTv, dyv, out_dtype=torch.float32)))
        dyc = dy[:, :N].contiguous().view(S, M // S, N)
        forms.append((f"xT NN, dy contig S={S}", lambda xTv=xTv, dyc=dyc: torch.bmm(xTv, dyc, out_dtype=torch.float32)))
        forms.append((f"dW^T = dy^T x (TN) S={S}", lambda xcv=xcv, dyc=dyc: torch.bmm(dyc.transpose(1, 2), xcv, out_dtype=torch.float32)))
    forms.append(("transpose copy of the hi plane", lambda: x2[:, :K].t().contiguous()))
    best = {}
    for fname, fn in forms:
        try:
            t = timeit(fn)
        except Exception as e:
            print(f"{name:8s} {fname:30s} failed: {str(e)[:80]}")
            continue
        kind = fname.split(" S=")[0]
        best[kind] = min(best.get(kind, 1e9), t)
        print(f"{name:8s} {fname:30s} {t:8.1f} us  {fl / t / 1e6:7.0f} TF/s")
    for k, v in best.items():
        tot[k] = tot.get(k, 0.0) + v
print("sum over the four GEMMs, best S each:", {k: round(v, 1) for k, v in tot.items()})
