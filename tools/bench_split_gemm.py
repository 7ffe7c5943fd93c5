"""Microbenchmark: fp32 hipBLASLt GEMM vs split-bf16 (3 bf16 GEMMs with fp32 output) on the encoder shapes."""
import time
import torch
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)

def split(a):
    hi = a.to(torch.bfloat16)
    lo = (a - hi.float()).to(torch.bfloat16)
    return hi, lo

def mm3(a, b):
    ah, al = split(a); bh, bl = split(b)
    o = torch.mm(ah, bh, out_dtype=torch.float32)
    o += torch.mm(ah, bl, out_dtype=torch.float32)
    o += torch.mm(al, bh, out_dtype=torch.float32)
    return o

def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

def rel(a, b): return float((a.double() - b).abs().max() / b.abs().max())

for (M, K, N) in [(20480, 1024, 1024), (20480, 1024, 4096), (20480, 4096, 1024), (80, 270336, 512), (1024, 20480, 4096)]:
    a = torch.randn(M, K, device=dev, generator=g); b = torch.randn(K, N, device=dev, generator=g)
    ref = a.double() @ b.double() if M * N * K < 2e11 else None
    t32 = timeit(lambda: a @ b)
    try:
        t3 = timeit(lambda: mm3(a, b))
        ah, al = split(a); bh, bl = split(b)
        t1 = timeit(lambda: torch.mm(ah, bh, out_dtype=torch.float32))
        tb = timeit(lambda: torch.mm(ah, bh))
        e3 = rel(mm3(a, b), ref) if ref is not None else -1
        e1 = rel(torch.mm(ah, bh, out_dtype=torch.float32), ref) if ref is not None else -1
    except Exception as ex:
        print("split failed:", type(ex).__name__, str(ex)[:200]); t3 = t1 = tb = e3 = e1 = float("nan")
    fl = 2.0 * M * K * N
    print(f"{(M,K,N)}: fp32 {t32*1e3:.3f} ms ({fl/t32/1e12:.0f} TF) err {rel(a@b, ref) if ref is not None else -1:.1e} | "
          f"3xbf16 {t3*1e3:.3f} ms err {e3:.1e} | one bf16->f32 {t1*1e3:.3f} ms ({fl/t1/1e12:.0f} TF) err {e1:.1e} | bf16->bf16 {tb*1e3:.3f} ms")
