"""a14 + a15 of hidden1_weights from the factors of its gradient (lpm_factored_clip_adam) at a BASELINE shape.
  python tools/time_factored.py [R N1 N2]     (cfg-2: 80 270336 512, cfg-5: 128 540672 1024; 8 towers of cfg-2: 640 270336 512)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import ops
dev = torch.device("cuda:0")
a = [int(v) for v in sys.argv[1:]]
R, N1, N2 = a[:3] if len(a) >= 3 else (80, 270336, 512)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(min(R, 128), N1, device=dev, generator=g)
dy = torch.randn(min(R, 128), N2, device=dev, generator=g) * 1e-3
fg = ops.FactoredGradient()
fg.put(x, dy)
if R > 128:          # several towers: the same tile buffers concatenated
    assert R % x.shape[0] == 0
    fg.xt, fg.dyt, fg.R = torch.cat([fg.xt] * (R // x.shape[0])), torch.cat([fg.dyt] * (R // x.shape[0])), R
P, M, V = torch.randn(N1 * N2, device=dev, generator=g), torch.zeros(N1 * N2, device=dev), torch.zeros(N1 * N2, device=dev)
sc = None
for i in range(3):
    sc = fg.clip_adam(P, M, V, 1.0, 2e-4, i + 1, scratch=sc)
torch.cuda.synchronize()
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    fg.clip_adam(P, M, V, 1.0, 2e-4, 4 + i, scratch=sc)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / n
print(f"R={R} N1={N1} N2={N2} LPM_FA_DBG={os.environ.get('LPM_FA_DBG', '0')}: {t * 1e3:.1f} us per update "
      f"({N1 * N2 * 24 / t / 1e9:.2f} TB/s of param / m / v traffic)")
