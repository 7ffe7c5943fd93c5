import torch, os
print("allow_tf32", torch.backends.cuda.matmul.allow_tf32, "prec", torch.get_float32_matmul_precision(),
      "preferred_blas", torch.backends.cuda.preferred_blas_library() if hasattr(torch.backends.cuda, "preferred_blas_library") else None)
for k, v in os.environ.items():
    if "BLAS" in k or "TF32" in k or "ROCBLAS" in k or "HIPBLAS" in k or "TUNABLE" in k:
        print(k, v)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())
for (M, K, N) in [(4, 128, 400), (128, 4, 400), (80, 270336, 512), (24000, 1024, 256), (1024, 24000, 256), (20480, 1024, 4096), (512, 80, 1000)]:
    a = torch.randn(M, K, device=dev, generator=g); b = torch.randn(K, N, device=dev, generator=g)
    ref = a.double() @ b.double()
    at = a.t().contiguous(); bt = b.t().contiguous()
    print((M, K, N), "NN %.2e" % rel(a @ b, ref), "TN %.2e" % rel(at.t() @ b, ref), "NT %.2e" % rel(a @ bt.t(), ref), "TT %.2e" % rel(at.t() @ bt.t(), ref))
x = torch.randn(80, 256, 1024, device=dev, generator=g); w = torch.randn(1024, 1024, device=dev, generator=g)
print("3d x@w %.2e" % rel(x @ w, x.double() @ w.double()))
