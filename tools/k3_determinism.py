"""K3's k-major form (lpm_vlad_aggregate_bwd_tiles with LPM_VLAD_RAW_KMAJOR) called over and over on the same inputs: the head of its
workspace (dots [B][3][K], u, v, ctil) and dassign must come out the same bits every time.  Run two copies at once to get the
contention of the two-rank tests.   python tools/k3_determinism.py [B T D K] [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import _capi, ops

a = [int(v) for v in sys.argv[1:]]
B, T, D, K = a[:4] if len(a) >= 4 else (16, 32, 1024, 256)
N = a[4] if len(a) > 4 else 300
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
dout, U = r(B, K, D) * 1e-3, r(B, K, D)
asum = torch.rand(B, K, device=dev, generator=g) * T / K
colsq = (U * U).sum(-1)
csq = torch.ones(B, K, device=dev)
gsq = csq.sum(-1)
logits, scale, shift = r(B * T, K), torch.rand(K, device=dev, generator=g) + 0.5, r(K) * 0.1
x, centres = r(B * T, D), r(D, K) / 32
lib = _capi.load()
flags = _capi.LPM_VLAD_SOFTMAX | _capi.LPM_VLAD_RESIDUAL
first = None
bad = 0
head = B * 16 * 3 * K + 3 * B * K                     # floats: dots (16 split slots, the k-major form fills the first) | u | v | ctil
nwref = torch.einsum("bkd,dk->bk", U * torch.rsqrt(colsq.clamp_min(1e-12))[..., None], centres)
for it in range(N):
    dlt, dcentres, (ws, wsb), g0 = ops._aggregate_bwd_tiles(lib, dout, U, asum, colsq, csq, gsq, logits, scale, shift, x, None, centres,
                                                           B, T, D, K, flags, True, no_dx=True, nrm_raw=True, raw_kmajor=True)
    torch.cuda.synchronize()
    hd = ws.view(torch.float32)[:head].clone()
    dots = hd[:B * 3 * K].view(B, 3, K)
    got = dict(dots0=dots[:, 0].clone(), dots1=dots[:, 1].clone(), dots2=dots[:, 2].clone(), uvc=hd[B * 16 * 3 * K:].clone(), dlt=dlt.clone(),
               dcentres=dcentres.clone(), g0=g0.clone())
    if first is None:
        first = got
        print(f"<N, W2> against torch: max abs {float((got['dots2'] - nwref).abs().max()):.3e} (scale {float(nwref.abs().max()):.3e})")
        continue
    for k, v in got.items():
        if not torch.equal(v, first[k]):
            bad += 1
            d = (v - first[k]).abs()
            extra = ""
            if k == "dots2":
                i = int(d.argmax())
                extra = (f"; entry {i}: first {float(first[k].flatten()[i]):.7f}, now {float(v.flatten()[i]):.7f}, torch {float(nwref.flatten()[i]):.7f}")
            print(f"iteration {it}: {k}: {int((d > 0).sum())} of {d.numel()} entries differ, max abs {float(d.max()):.3e}{extra}")
print(f"B={B} T={T} D={D} K={K}: {N} iterations, {bad} differing tensors" + ("" if bad else " -- bit-identical"))
