#!/usr/bin/env python3
"""Round-5 diagnosis: which operand producer costs the fp16 encoder path accuracy (W1's gradient in tests/test_gpu_fp16x2.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from learnablepoolingmethods_amd import FLAGS, ops


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


class _calibrated:
    """fn once in split-bf16 under an OperandScales (measures max |x| per site), then on fp16 planes with those scales."""

    def __init__(self, dev):
        self.sc = ops.OperandScales(dev)

    def run(self, fn):
        sc = self.sc
        sc.enabled = False
        sc.begin_step()
        ops._ACTIVE_SCALES = sc
        try:
            fn()
        finally:
            ops._ACTIVE_SCALES = None
        sc.calibrate_from_device()
        sc.enabled = True
        sc.begin_step()
        ops._ACTIVE_SCALES = sc
        try:
            return fn()
        finally:
            ops._ACTIVE_SCALES = None


dev = torch.device("cuda:0")
B, L, F, heads = 4, 256, 1024, 64
g = torch.Generator().manual_seed(11)
def P(*shape, s=1.0):
    return (torch.randn(*shape, generator=g) * s).to(dev).requires_grad_(True)
x = P(B, L, F, s=2e-3)
Wq, Wk, Wv, Wo = (P(F, F, s=F ** -.5) for _ in range(4))
bo = P(F, s=0.1)
g0, be0, g1, be1, g2, be2 = (P(F, s=0.1) for _ in range(6))
with torch.no_grad():
    for t in (g0, g1, g2):
        t += 1.0
W1, b1, W2, b2 = P(F, 4 * F, s=F ** -.5), P(4 * F, s=0.1), P(4 * F, F, s=(4 * F) ** -.5), P(F, s=0.1)
dout = (torch.randn(B, L, F, generator=g) * 1e-4).to(dev)
params = (x, Wq, Wk, Wv, Wo, bo, g0, be0, W1, b1, W2, b2, g1, be1, g2, be2)
names = "x Wq Wk Wv Wo bo g0 be0 W1 b1 W2 b2 g1 be1 g2 be2".split()
def fn():
    for t in params:
        t.grad = None
    a = ops.attention_block_x3(x, Wq, Wk, Wv, Wo, bo, g0, be0, heads, (F // heads) ** -0.5, next_kernel=W1)
    out = ops.ffn_block_x3(a, W1, b1, W2, b2, g1, be1, g2, be2)
    out.backward(dout)
    return out.detach().clone(), [t.grad.detach().clone() for t in params]
ref_out, ref_g = fn()
def run(tag):
    cal = _calibrated(dev)
    out, grads = cal.run(fn)
    errs = {nm: rel_l2(a, b) for nm, a, b in zip(names, grads, ref_g)}
    print(tag, "out %.1e" % rel_l2(out, ref_out), " ".join(f"{k}={v:.1e}" for k, v in errs.items()), flush=True)
    return cal
cal = run("default          ")
for k, (a, s) in cal.sc.report().items():
    print("   site", k[0], "amax %.3e scale 2^%d" % (a, round(__import__('math').log2(s))))
ops.LN_IMAGE = False; run("LN_IMAGE off     "); ops.LN_IMAGE = True
FLAGS.ln_gradient_image = False; run("ln_grad_image off"); FLAGS.ln_gradient_image = True
FLAGS.mha_gradient_image = False; run("mha_image off    "); FLAGS.mha_gradient_image = True
ops.FFN_TILES = False; run("FFN_TILES off    "); ops.FFN_TILES = True
