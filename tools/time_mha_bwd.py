"""K4 backward alone, two-term fp16 against three-term bf16 (lpm_mha_bwd_set_terms), at the V1 video-encoder shape (B=80, L=256, h=64, d=16)
and the V2 frame-encoder shape with logits_bn (B=80, L=300): HIP events around whole backward calls (dq + dkv kernels [+ statistics]).
  python tools/time_mha_bwd.py [iters]        (under rocprofv3 --kernel-trace --stats for the per-kernel table)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import _capi, ops

lib = _capi.load()
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (B, L, h, d, bn) in ((80, 256, 64, 16, False), (80, 300, 64, 16, True), (80, 64, 16, 8, False)):
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v, do = (torch.randn(B, L, h * d, device=dev, generator=g).requires_grad_(True) for _ in range(4))
    gamma, beta = (1 + 0.1 * torch.randn(L, device=dev, generator=g)).requires_grad_(True), torch.zeros(L, device=dev, requires_grad=True)
    mm, mv = torch.zeros(L, device=dev), torch.ones(L, device=dev)
    res = {}
    grads = {}
    for terms in (3, 2, 3, 2):
        prev = lib._lpm_mha_bwd_set_terms(terms)
        o = ops.mha_core_bn(q, k, v, h, gamma, beta, mm, mv, is_training=True) if bn else ops.mha_core(q, k, v, h, d ** -0.5)
        res.setdefault(terms, []).append(timeit(lambda: o.backward(do, retain_graph=True)))
        q.grad = k.grad = v.grad = None
        o.backward(do, retain_graph=True)
        grads[terms] = [t.grad.clone() for t in (q, k, v)]
        lib._lpm_mha_bwd_set_terms(prev)
    dist = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(grads[2], grads[3])]
    print(f"B={B} L={L} h={h} d={d} logits_bn={bn}: backward three terms {min(res[3]):7.1f} us, two terms {min(res[2]):7.1f} us; "
          f"max |two - three| / max |three| (dq, dk, dv) = " + ", ".join(f"{x:.1e}" for x in dist), flush=True)
