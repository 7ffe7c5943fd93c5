import torch, time
dev = torch.device("cuda:0")
M, K, N = 20480, 1024, 4096
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
a3 = torch.randn(M, 3 * K, device=dev).bfloat16(); w3 = torch.randn(N, 3 * K, device=dev).bfloat16()
t = bench(lambda: torch.mm(a3, w3.t(), out_dtype=torch.float32))
print(f"bf16 K'=3K  [{M}x{3*K}]x[{3*K}x{N}] -> fp32: {t:.1f} us = {2*M*3*K*N/t/1e9:.2f} PF/s")
a1 = a3[:, :K].contiguous(); w1 = w3[:, :K].contiguous()
t = bench(lambda: torch.mm(a1, w1.t(), out_dtype=torch.float32))
print(f"bf16 K'=K: {t:.1f} us = {2*M*K*N/t/1e9:.2f} PF/s")
for dt in (torch.float8_e4m3fn, torch.float8_e4m3fnuz):
    try:
        a8 = torch.randn(M, 2 * K, device=dev).to(dt); w8 = torch.randn(N, 2 * K, device=dev).to(dt)
        sa = torch.tensor(1.0, device=dev); sb = torch.tensor(1.0, device=dev)
        for od in (torch.float32, torch.bfloat16):
            try:
                f = lambda: torch._scaled_mm(a8, w8.t(), scale_a=sa, scale_b=sb, out_dtype=od)
                t = bench(f)
                print(f"{dt} K'=2K -> {od}: {t:.1f} us = {2*M*2*K*N/t/1e9:.2f} PF/s")
            except Exception as e:
                print(dt, od, "failed:", str(e)[:150])
    except Exception as e:
        print(dt, "failed:", str(e)[:150])
