"""The update pass of hidden1_weights with its bf16 compute copy at cfg-5's shape, with and without the projection's input gradient riding along.
  python tools/time_factored_fold.py [copy|dx] [R N1 N2]     (LPM_FA_FOLD=2: the row-block form without dx; default shape 128 540672 1024)
Prints the update's time per call (norm pass included) and, for reference, lpm_proj_dx_w16 alone at the same shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import _capi, ops
from learnablepoolingmethods_amd._capi import ptr, stream_ptr
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "dx"
a = [int(v) for v in sys.argv[2:]]
R, N1, N2 = a[:3] if len(a) >= 3 else (128, 540672, 1024)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(R, N1, device=dev, generator=g)
dy = torch.randn(R, N2, device=dev, generator=g) * 1e-3
fg = ops.FactoredGradient()
fg.put(x, dy)
P, M, V = torch.randn(N1 * N2, device=dev, generator=g) / 30, torch.zeros(N1 * N2, device=dev), torch.zeros(N1 * N2, device=dev)
C = P.view(N1, N2).to(torch.bfloat16).contiguous()
dx = torch.empty(R, N1, device=dev) if mode == "dx" else None
sc = None
for i in range(3):
    sc = fg.clip_adam(P, M, V, 1.0, 2e-4, i + 1, scratch=sc, param_bf16=C, dx=dx)
torch.cuda.synchronize()
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    fg.clip_adam(P, M, V, 1.0, 2e-4, 4 + i, scratch=sc, param_bf16=C, dx=dx)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / n
print(f"mode={mode} LPM_FA_FOLD={os.environ.get('LPM_FA_FOLD', '1')} R={R} N1={N1} N2={N2}: {t * 1e3:.1f} us per update "
      f"({N1 * N2 * 26 / t / 1e9:.2f} TB/s of param / m / v / copy traffic)")
lib = _capi.load()
if N2 % 64 == 0:
    dyt = ops._tile_buffer(lib._lpm_row_tiles_bytes(1, R, N2), dy)
    lib.check(lib._lpm_split_rows_tiles(ptr(dy), N2, 1, R, N2, ptr(dyt), stream_ptr()), "tiles")
    d2 = torch.empty(R, N1, device=dev)
    f = lambda: lib.check(lib._lpm_proj_dx_w16(ptr(dyt), ptr(C), R, N1, N2, ptr(d2), d2.stride(0), stream_ptr()), "dx")
    for _ in range(3):
        f()
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"   lpm_proj_dx_w16 alone: {e0.elapsed_time(e1) / n * 1e3:.1f} us")
    if dx is not None:
        torch.cuda.synchronize()
        # (dx was formed from the weights BEFORE the last update; d2 from the copy after it: not comparable -- the test compares)
