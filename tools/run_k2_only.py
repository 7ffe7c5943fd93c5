"""Runs only the kernels of the a5 function (frame_level_models.py:2798-2822 / video_pooling_modules.py:1641-1658) of one BASELINE
configuration a few times: a short target for rocprofv3 --pmc passes (HBM traffic of the a5 chain, MFMA counters of K1).
  run_k2_only.py <launches> cfg2 [lazy|eager]  NetVladV1 video stream, B=80 T=300 D=1024 K=256: K1, assign_tiles2, K2 raw k-major, row scales
  run_k2_only.py <launches> cfg3 [lazy|eager]  NetVladV2 video stream, same sizes, similarities given: K2 (softmax stage off) + row scales (eager: finalize2)
  run_k2_only.py <launches> cfg5 [lazy|eager]  gated NetVLAD video stream, B=128 T=300 D=1024 K=512, bf16 storage: K1, assign tiles, K2, row scales (eager: finalize2)
(the old form `run_k2_only.py <launches> [lazy|eager]` = cfg2 still works)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cfg = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2].startswith("cfg") else "cfg2"
mode = sys.argv[3] if len(sys.argv) > 3 else (sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("cfg") else "lazy")
g = torch.Generator(device=dev).manual_seed(0)
if cfg == "cfg2":
    B, T, D, K = 80, 300, 1024, 256
    x = torch.randn(B * T, 1152, device=dev, generator=g)
    W = (torch.randn(D, K, device=dev, generator=g) / 32).requires_grad_(True)      # a gradient is wanted: the training-mode chain
    W2 = torch.randn(1, D, K, device=dev, generator=g) / 32
    bn = (torch.ones(K, device=dev), torch.zeros(K, device=dev), torch.zeros(K, device=dev), torch.ones(K, device=dev))
    for _ in range(n):
        out = ops.netvlad(x[:, :D], W, W2, T, bn=bn, kmajor=True, lazy=(mode == "lazy"))
elif cfg == "cfg3":
    B, T, D, K = 80, 300, 1024, 256
    x = torch.randn(B * T, D, device=dev, generator=g)
    sims = torch.randn(B, T, K, device=dev, generator=g).requires_grad_(True)        # BN(relu(.)) output of the frame encoder: any sign
    centres = torch.randn(D, K, device=dev, generator=g) / 32
    for _ in range(n):
        out = ops.materialise(ops.vlad_aggregate(sims, x, centres, T, lazy=(mode == "lazy")))      # (the materialise pass is the tool's own: not part of the chain)
else:
    B, T, D, K = 128, 300, 1024, 512
    raw = torch.randn(B, T, 1152, device=dev, generator=g)
    nf = torch.full((B,), T, dtype=torch.int32, device=dev)
    W = (torch.randn(D, K, device=dev, generator=g) / 32).requires_grad_(True)
    W2 = torch.randn(1, D, K, device=dev, generator=g) / 32
    bn = (torch.ones(K, device=dev), torch.zeros(K, device=dev), torch.zeros(K, device=dev), torch.ones(K, device=dev))
    for _ in range(n):
        y = ops.frame_sample_bn(raw, nf, T, storage="bf16", materialize=False)
        with torch.no_grad():
            xs = y[:, :D]
        out = ops.netvlad(xs, W, W2, T, bn=bn, is_training=True, storage="bf16", lazy=(mode == "lazy"))
    out = ops.materialise(out)
torch.cuda.synchronize()
print("ok", cfg, float(out.detach().float().norm()))
