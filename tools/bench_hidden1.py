import time, torch
dev="cuda"; g=torch.Generator(device=dev).manual_seed(0)
M,K,N=80,270336,512
x=torch.randn(M,K,device=dev,generator=g); W=torch.randn(K,N,device=dev,generator=g); dy=torch.randn(M,N,device=dev,generator=g)
def timeit(f,n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
ref=(x.double()@W.double())
print("mm fwd", timeit(lambda: x@W))
for S in (8,16,33,66,132):
    if K%S: continue
    f=lambda: torch.bmm(x.view(M,S,K//S).transpose(0,1), W.view(S,K//S,N)).sum(0)
    print("splitK",S, timeit(f), float((f().double()-ref).abs().max()/ref.abs().max()))
print("dW = x^T dy", timeit(lambda: x.t()@dy))
print("dx = dy W^T", timeit(lambda: dy@W.t()))
