"""a9 at a BASELINE shape: forward y = x W and dx = dy W^T, hand-written weight-stream kernels vs the library GEMMs.
  python tools/time_proj.py [M Kd N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import ops
dev = torch.device("cuda:0")
a = [int(v) for v in sys.argv[1:]]
M, Kd, N = a[:3] if len(a) >= 3 else (80, 270336, 512)
PAD = a[3] if len(a) > 3 else 0                 # extra floats per row of x (and dx): moves the row stride off a multiple of 2^15 bytes
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, Kd + PAD, device=dev, generator=g)[:, :Kd].requires_grad_(True)
W = (torch.randn(Kd, N, device=dev, generator=g) / 16).requires_grad_(True)
dy = torch.randn(M, N, device=dev, generator=g)
ref = None
for on in (True, False):
    ops.PROJ_STREAM = on
    ops.KERNEL_TIMELINE = None
    for _ in range(3):
        y = ops.projection(x, W)
        dxv, = torch.autograd.grad(y, x, dy, retain_graph=False)
    torch.cuda.synchronize()
    n = 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(n):
        ev[0].record()
        y = ops.projection(x, W)
        ev[1].record()
        dxv, = torch.autograd.grad(y, x, dy)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    wb = Kd * N * 4
    print(f"M={M} Kd={Kd} N={N} pad={PAD} stream_kernels={on}: fwd {tf / n * 1e3:.1f} us ({wb / (tf / n) / 1e9:.2f} TB/s of weight), "
          f"dx {tb / n * 1e3:.1f} us ({wb / (tb / n) / 1e9:.2f} TB/s)")
    if ref is None:
        ref = (y.detach().clone(), dxv.clone())
    else:
        e1 = float((y - ref[0]).abs().max() / ref[0].abs().max()); e2 = float((dxv - ref[1]).abs().max() / ref[1].abs().max())
        print(f"  max rel difference hand-written vs library: y {e1:.2e}, dx {e2:.2e}")
