"""-m gpu: the fp16 operand formats of NetVladV1's encoder GEMMs (round 5; csrc/operand_format.h, ops.OperandScales).

Forward products keep three terms on fp16 (hi, lo) planes of both operands (as exact as the split-bf16 form: no extra ReLU flips);
backward products are TWO-term -- the gradient exact to ~2^-22, the other operand (the weight for dx, the activation for dW) rounded
once to fp16: 1.4e-4 per GEMM on random data (documented; the split-bf16 form is 5e-6).  Each producer of an operand image and each
consumer is held to the fp64 product here; the model-level tests keep the north-star's 1e-3 (tests/test_gpu_models.py runs NetVladV1
through these formats once the scales are calibrated).
Reference: transformer_utils.py:559-561,583,701-711 (the dense layers) and TF autodiff of them.
"""
import math

import pytest
import torch

from tests._util import assert_close, cuda, rel_l2

pytestmark = pytest.mark.gpu

TOL_GEMM = 4e-4        # a two-term product, one operand rounded once to fp16: 2^-12 / sqrt(3) = 1.4e-4 rms, x ~3 for the max norm
TOL_FWD = 2e-5         # a three-term product on fp16 planes


def _scales(dev):
    from learnablepoolingmethods_amd import ops
    return ops.OperandScales(dev)


class _calibrated:
    """Run ``fn`` twice under an OperandScales: once in split-bf16 (measures max |x| per site), then in fp16 with those scales."""

    def __init__(self, dev):
        self.sc = _scales(dev)

    def run(self, fn):
        from learnablepoolingmethods_amd import ops
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
        assert sc.fp16_now, "every site was measured: the second pass must run in fp16"
        ops._ACTIVE_SCALES = sc
        try:
            return fn()
        finally:
            ops._ACTIVE_SCALES = None


@pytest.mark.parametrize("scale_log2,mag", [(0, 1.0), (14, 1e-3), (-3, 40.0), (30, 1e-8)])
def test_split_rows_fp16_image_round_trip(scale_log2, mag):
    """x -> [hi | lo] fp16 planes of x * 2^s: (hi + lo) / 2^s gives x back to 2^-21 of the tensor's maximum wherever the scale puts the
    tensor inside the format, max |x| is recorded un-scaled, and values beyond the range saturate (finite)."""
    from learnablepoolingmethods_amd import _capi, ops
    dev = cuda()
    g = torch.Generator().manual_seed(5)
    M, K = 384, 256
    x = (torch.randn(M, K, generator=g) * mag).to(dev)
    x[0, :8] = 0.0
    amax = torch.zeros(_capi.LPM_OPERAND_AMAX_SUB * _capi.LPM_OPERAND_AMAX_STRIDE, device=dev)      # a site's sub-slots
    site = ops.OperandSite(True, 2.0 ** scale_log2, amax.data_ptr(), role="g")
    img = ops._split_rows(x, site=site)
    assert img.dtype == torch.float16 and tuple(img.shape) == (M, 2 * K)
    amax.zero_()
    img3 = ops._split_rows(x, site=ops.OperandSite(True, 2.0 ** scale_log2, amax.data_ptr(), role="a"))     # activations: [hi | lo | hi]
    assert tuple(img3.shape) == (M, 3 * K) and torch.equal(img3[:, :2 * K], img) and torch.equal(img3[:, 2 * K:], img[:, :K])
    back = (img[:, :K].double() + img[:, K:].double()) / 2.0 ** scale_log2
    top = float(x.abs().max())
    assert float(amax.max()) == top, "max |x| is recorded before scaling"
    if top * 2.0 ** scale_log2 <= 65504:
        err = float((back - x.double()).abs().max()) / top
        assert err <= 2.0 ** -20, f"round trip error {err:.2e} of the maximum"
    assert torch.isfinite(img.float()).all()
    # saturation: a value 100 x beyond the range stays finite
    big = torch.full((8, 64), 3.0e6, device=dev)
    imgb = ops._split_rows(big, site=ops.OperandSite(True, 1.0, amax.data_ptr(), role="g"))
    assert torch.isfinite(imgb.float()).all() and float(imgb[:, :64].float().max()) == 65504.0
    # the bf16x3 format through the same entry point is bit-identical to the round 1-4 entry point
    lib = _capi.load()
    old = torch.empty((M, 3 * K), dtype=torch.bfloat16, device=dev)
    lib.check(lib._lpm_split_rows(_capi.ptr(x), K, M, K, None, 0, 0, _capi.ptr(old), _capi.stream_ptr()), "lpm_split_rows")
    new = ops._split_rows(x, site=ops.OperandSite(False, 1.0, amax.data_ptr()))
    assert torch.equal(old.view(torch.int16), new.view(torch.int16))


@pytest.mark.parametrize("M,K,N", [(2048, 1024, 1024), (4096, 128, 512), (1536, 4096, 1024)])
def test_dense_fp16x2(M, K, N):
    """y = x W (three terms), dx, dW (two terms) of one dense layer on fp16 planes against fp64 -- gradients 1e-4 of the activations'
    size, so that the scales have work to do."""
    from learnablepoolingmethods_amd import ops
    dev = cuda()
    g = torch.Generator().manual_seed(M)
    x, W, dy = torch.randn(M, K, generator=g) * 3e-3, torch.randn(K, N, generator=g) / K ** 0.5, torch.randn(M, N, generator=g) * 1e-6
    xd, Wd = x.double().requires_grad_(True), W.double().requires_grad_(True)
    (xd @ Wd).backward(dy.double())
    xg, Wg = x.to(dev).requires_grad_(True), W.to(dev).requires_grad_(True)
    cal = _calibrated(dev)

    def fn():
        xg.grad = Wg.grad = None
        y = ops.dense_x3(xg, Wg)
        y.backward(dy.to(dev))
        return y
    y = cal.run(fn)
    rep = cal.sc.report()
    assert len(rep) == 2 and all(s != 1.0 for _, s in rep.values()), rep
    assert_close(y, (xd @ Wd).detach(), tol=TOL_FWD, what="dense fp16 fwd")
    assert_close(xg.grad, xd.grad, tol=TOL_GEMM, what="dense fp16x2 dx")
    assert_close(Wg.grad, Wd.grad, tol=TOL_GEMM, what="dense fp16x2 dW")
    # (dW: both operands rounded once to fp16 with ops.DW_TERMS = 1, the default: 2.9e-4 measured; 2.1e-4 with the gradient kept exact)
    assert rel_l2(xg.grad, xd.grad) <= 2.5e-4 and rel_l2(Wg.grad, Wd.grad) <= 3.5e-4


@pytest.mark.parametrize("M,F,H,tiles", [(2048, 128, 512, False), (2048, 256, 1024, True), (20480, 1024, 4096, True), (2048, 256, 1024, False)])
def test_ffn_fp16x2(M, F, H, tiles):
    """FeedForwardNetwork's core relu(y W1 + b1) W2 (transformer_utils.py:701-711) in the two-product format: the library path with the
    fused split passes and (tiles) the hand-written 256-row tile GEMM -- fp16 MFMAs, hi-plane weight tiles, image epilogues with scale,
    saturation and max |x| -- forward and backward against fp64."""
    from learnablepoolingmethods_amd import ops
    dev = cuda()
    g = torch.Generator().manual_seed(3)
    old = ops.FFN_TILES
    ops.FFN_TILES = tiles
    try:
        y, W1, b1, W2, dout = (torch.randn(M, F, generator=g), torch.randn(F, H, generator=g) / F ** .5, 0.3 * torch.randn(H, generator=g),
                               torch.randn(H, F, generator=g) / H ** .5, torch.randn(M, F, generator=g) * 1e-5)
        yd, W1d, b1d, W2d = (t.double().requires_grad_(True) for t in (y, W1, b1, W2))
        ref = torch.relu(yd @ W1d + b1d) @ W2d
        ref.backward(dout.double())
        yg, W1g, b1g, W2g = (t.to(dev).requires_grad_(True) for t in (y, W1, b1, W2))
        cal = _calibrated(dev)

        def fn():
            for t in (yg, W1g, b1g, W2g):
                t.grad = None
            out = ops.ffn_x3(yg, W1g, b1g, W2g)
            out.backward(dout.to(dev))
            return out
        out = cal.run(fn)
        assert len(cal.sc.report()) == 4
        assert_close(out, ref, tol=5e-5, what="ffn fp16 fwd")
        for got, want, nm in ((yg, yd, "dy"), (W1g, W1d, "dW1"), (b1g, b1d, "db1"), (W2g, W2d, "dW2")):
            e = rel_l2(got.grad, want.grad)
            print(f"[ffn fp16 M={M} F={F} H={H} tiles={tiles}] gradient of {nm}: {e:.2e}")
            assert e <= 5e-3, f"ffn fp16x2 {nm}: relative L2 error {e:.3e}"      # (ReLU flips near zero: see test_ffn_split_bf16_fused_bias_relu)
        err = (yg.grad.double().cpu() - yd.grad).abs().amax(dim=1) / yd.grad.abs().max()
        assert int((err > 1e-3).sum()) <= max(20, M // 100), f"{int((err > 1e-3).sum())} of {M} dy rows differ: not ReLU-flip noise"
    finally:
        ops.FFN_TILES = old


@pytest.mark.parametrize("B,L,F,heads", [(4, 256, 1024, 64), (6, 64, 128, 16)])
def test_encoder_blocks_fp16x2_against_split_bf16(B, L, F, heads):
    """The two block Functions of the V1 cluster encoder (transformer_utils.py:374-413) on fp16 planes against the same blocks in
    split-bf16: the output to 1e-4 (three-term forward), every gradient to 2e-3 in the Frobenius norm (measured 5e-4 at cfg-2's
    width) -- five two-term products deep plus the attention backward in between, and NO ReLU margins here (random weights: a few
    hidden units flip between the two forwards); the model-level tests hold the whole step to 1e-3 on prepared weights
    (tests/test_gpu_models.py)."""
    from learnablepoolingmethods_amd import ops
    dev = cuda()
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
    ref_out, ref_g = fn()                                   # no scales active: split-bf16 x3
    cal = _calibrated(dev)
    out, grads = cal.run(fn)
    assert cal.sc.steps_fp16 == 1 and len(cal.sc.report()) == 8, cal.sc.report()
    assert_close(out, ref_out, tol=1e-4, what="encoder fp16 output")
    for nm, a, b in zip(names, grads, ref_g):
        e = rel_l2(a, b)
        print(f"[encoder fp16 B={B} L={L} F={F}] gradient of {nm}: {e:.2e}")
        # W1 / b1 sit right behind the first ReLU: the two FORWARDS differ by ~5e-6 (split-bf16's own error), which flips a few of the
        # ~1e6 hidden units' masks -- each flip moves these two gradients by ~1e-3 (tools/fp16_diag.py: b1, the mask's column sums, carries
        # the same 2e-3 as W1 while W2 / b2 behind it are at 3e-4); not an error of either arithmetic
        tol = 8e-3 if nm in ("W1", "b1") else 2e-3
        assert e <= tol, f"encoder fp16 gradient of {nm}: relative L2 error {e:.3e}"


def test_operand_scales_delay_and_warm_in():
    """The scale of step t comes from maxima measured at EARLIER steps: the first steps of a run stay on split-bf16 until a read-back has
    arrived, then every step is fp16; a site's scale puts the larger of its last two measured maxima into [2^10, 2^11)."""
    from learnablepoolingmethods_amd import ops
    dev = cuda()
    sc = _scales(dev)
    x = torch.randn(512, 256, device=dev) * 1e-4
    W = (torch.randn(256, 256, device=dev) / 16).requires_grad_(True)
    modes = []
    for step in range(6):
        sc.begin_step()
        ops._ACTIVE_SCALES = sc
        try:
            modes.append(sc.fp16_now)
            xs = (x * (3.0 if step == 4 else 1.0)).requires_grad_(True)
            ops.dense_x3(xs, W).sum().backward()
        finally:
            ops._ACTIVE_SCALES = None
        torch.cuda.synchronize()                            # (so that the read-back of this step is there at the next begin_step)
    assert modes[0] is False and modes[1] is False and all(modes[2:]), modes
    sc.begin_step()                                         # queues the read-back of the last step ...
    torch.cuda.synchronize()
    sc._harvest()                                           # ... and here it is: the host holds steps 4 (three times larger) and 5
    amax, scale = sc.report()[("a", W.data_ptr())]
    assert 2 ** 10 <= max(sc.hist[0][0], sc.hist[1][0]) * scale < 2 ** 11
    assert math.isclose(max(sc.hist[0][0], sc.hist[1][0]), 3.0 * float(x.abs().max()), rel_tol=1e-6), "the larger of the last two measurements"


def test_a_nan_operand_reaches_the_host_through_the_recorded_maximum():
    """the fp16 split clamps with v_med3_f32, which turns a NaN into -65504 -- a NaN activation or gradient entering an encoder
    GEMM would go on as a finite value.  The producers' recorded maxima keep it (bit-pattern maximum: NaN above Inf above everything
    finite) and ops.OperandScales raises when the maximum arrives on the host."""
    from learnablepoolingmethods_amd import ops
    dev = cuda()
    sc = ops.OperandScales(dev)
    sc.begin_step()
    W = torch.zeros(64, 64, device=dev)
    ops._ACTIVE_SCALES = sc
    try:
        site = ops._site("a", W)
        x = torch.randn(256, 64, device=dev)
        ops._split_rows(x, site=site)
        sc.calibrate_from_device()                     # finite: fine
        assert sc.report()[("a", W.data_ptr())][0] == pytest.approx(float(x.abs().max()), rel=1e-6)
        sc.begin_step()
        site = ops._site("a", W)
        x[17, 5] = float("nan")
        ops._split_rows(x, site=site)
        with pytest.raises(ops.LpmError, match="NaN / Inf"):
            sc.calibrate_from_device()
        x[17, 5] = float("inf")
        sc.amax.zero_()
        ops._split_rows(x, site=site)
        with pytest.raises(ops.LpmError, match="NaN / Inf"):
            sc.calibrate_from_device()
    finally:
        ops._ACTIVE_SCALES = None
