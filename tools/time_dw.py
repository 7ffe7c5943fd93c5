"""a9 backward, dW = x^T dy of the hidden projection (lpm_skinny_weight_grad_tiles) at a BASELINE shape.
  python tools/time_dw.py [M Kd N]     (cfg-2: 80 270336 512, cfg-5: 128 540672 1024)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import ops
dev = torch.device("cuda:0")
a = [int(v) for v in sys.argv[1:]]
M, Kd, N = a[:3] if len(a) >= 3 else (80, 270336, 512)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, Kd, device=dev, generator=g)
dy = torch.randn(M, N, device=dev, generator=g)
out = torch.empty(Kd, N, device=dev)
for _ in range(3):
    ops.skinny_weight_grad(x, dy, out=out)
torch.cuda.synchronize()
n = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ops.skinny_weight_grad(x, dy, out=out)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / n
ref = x[:, :4096].double().t() @ dy.double()
err = float((out[:4096].double() - ref).abs().max() / ref.abs().max())
print(f"M={M} Kd={Kd} N={N} LPM_DW_COLS_INNER={os.environ.get('LPM_DW_COLS_INNER', '1')}: {t * 1e3:.1f} us incl. operand splits "
      f"({Kd * N * 4 / t / 1e9:.2f} TB/s of gradient written), max rel err {err:.1e}")
