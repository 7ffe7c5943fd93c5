"""-m gpu: NetVladV1 / NetVladV2 through the registry API vs the oracle (forward, one and two
optimiser steps).  BASELINE configs: cfg-1 exactly; cfg-2 / cfg-3 at their real layer sizes with the
batch cut to what the CPU oracle finishes in seconds."""
import numpy as np
import pytest
import torch

from oracle import lpm_oracle as O
from tests._util import assert_close, cuda, rel_err, rel_l2

pytestmark = pytest.mark.gpu


def _product_forward(name, params, x, nf, cfg, is_training, dev, **kw):
    from learnablepoolingmethods_amd import registry
    from learnablepoolingmethods_amd import variables as vs
    store = vs.VariableStore(device=dev, seed=1)
    model = registry.get_model(name)
    args = dict(vocab_size=cfg.vocab_size, num_frames=nf.to(dev), iterations=cfg.iterations, cluster_size=cfg.cluster_size,
                hidden_size=cfg.hidden_size, is_training=is_training, **kw)
    with vs.use_store(store):
        with torch.no_grad():
            model.create_model(x.to(dev), **args)          # creates the variables
        missing = [n for n in params if n not in store.vars and not n.endswith("cluster_biases")]
        assert not missing, f"variables the oracle has but the model did not create: {missing}"
        extra = [n for n in store.vars if n not in params]
        assert not extra, f"variables the model created that the oracle does not know: {extra}"
        store.load(params)
        out = model.create_model(x.to(dev), **args)
        store.pop_regularization_losses()
    return out["predictions"], store


@pytest.mark.parametrize("training", [True, False])
def test_cfg1_netvladv1_forward(training):
    """BASELINE configs[0]: NetVladV1 K=16 hidden=128, 30-frame rgb-only (1024-d), bs=8."""
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV1", iterations=30, cluster_size=16, hidden_size=128)
    x, nf, _ = O.make_synthetic_batch(8, 30, 1024, cfg.vocab_size, seed=0, min_frames=10)
    p = O.init_params(cfg, 1024, seed=1000)
    ref = O.model_forward({k: v.double() for k, v in p.items()}, x.double(), nf, cfg, training)
    pred, _ = _product_forward("NetVladV1", p, x, nf, cfg, training, dev)
    assert pred.shape == (8, 3862)
    assert_close(pred, ref, what="cfg-1 predictions")


def test_netvladv1_rgb_audio_small():
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV1", iterations=20, cluster_size=32, hidden_size=64, vocab_size=100)
    x, nf, _ = O.make_synthetic_batch(4, 25, 1152, 100, seed=1, min_frames=8)
    p = O.init_params(cfg, 1152, seed=1001)
    ref = O.model_forward({k: v.double() for k, v in p.items()}, x.double(), nf, cfg, True)
    pred, _ = _product_forward("NetVladV1", p, x, nf, cfg, True, dev)
    assert_close(pred, ref, what="V1 rgb+audio predictions")


def test_gated_netvlad_no_encoder():
    """BASELINE cfg-5 family: NetVLAD + gating + MoE-4 without the cluster encoders."""
    from learnablepoolingmethods_amd import FLAGS
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV1", iterations=16, cluster_size=64, hidden_size=64, vocab_size=50, encoder=False,
                         moe_num_mixtures=4)
    x, nf, _ = O.make_synthetic_batch(3, 20, 1152, 50, seed=2, min_frames=8)
    p = O.init_params(cfg, 1152, seed=1002)
    ref = O.model_forward({k: v.double() for k, v in p.items()}, x.double(), nf, cfg, True)
    FLAGS.moe_num_mixtures = 4
    try:
        pred, _ = _product_forward("NetVladV1", p, x, nf, cfg, True, dev, encoder=False)
    finally:
        FLAGS.reset()
    assert_close(pred, ref, what="gated NetVLAD predictions")


def test_netvladv2_forward_with_dropout_mask():
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV2", iterations=24, cluster_size=32, hidden_size=64, vocab_size=80)
    B = 3
    x, nf, _ = O.make_synthetic_batch(B, 30, 1152, 80, seed=3, min_frames=10)
    p = O.init_params(cfg, 1152, seed=1003)
    g = torch.Generator().manual_seed(4)
    masks = {"video": (torch.rand(B, 24, 1024, generator=g) >= 0.9).float(),
             "audio": (torch.rand(B, 24, 128, generator=g) >= 0.9).float()}
    ref = O.model_forward({k: v.double() for k, v in p.items()}, x.double(), nf, cfg, True,
                          dropout_masks={k: v.double() for k, v in masks.items()})
    pred, _ = _product_forward("NetVladV2", p, x, nf, cfg, True, dev,
                               dropout_masks={k: v.to(dev) for k, v in masks.items()})
    assert_close(pred, ref, what="V2 predictions")


def _relu_units_at_rounding_distance(p, x, nf, cfg, dm64, reach=1e-5):
    """How many ReLU units of the fp64 forward have a pre-activation closer to zero than fp32 arithmetic can resolve (|z| < reach x
    the site's rms): the units whose mask any fp32 implementation may decide differently from the oracle.  -> {site: (units, total)}"""
    O.RELU_TAPS = {}
    try:
        with torch.no_grad():
            O.model_forward(p, x, nf, cfg, True, None, dm64)
        taps = O.RELU_TAPS
    finally:
        O.RELU_TAPS = None
    out = {}
    for site, z in taps.items():
        z = z.reshape(-1, z.shape[-1]).double()
        out[site] = ((z.abs() < reach * float(z.pow(2).mean().sqrt())).any(0).nonzero().flatten(), int(z.shape[-1]))
    return out


def _full_size_compare(name, cfg, B, seed, dev, scale_hidden1=True, dropout_masks=None, grad_tol=1e-3, prepare_relu=True, **kw):
    """One training step at a BASELINE configuration's real layer sizes through the Trainer against the fp64 oracle: the named
    intermediates of the forward (the product reports them as summaries), loss, predictions and the gradient of EVERY variable.
    The seeded weights keep all ReLU pre-activations away from zero (oracle/test_weights.separate_relu_units), so the whole-model
    gradients are held to ``grad_tol`` = the north-star's 1e-3 in the Frobenius norm of each variable."""
    from learnablepoolingmethods_amd import registry
    from learnablepoolingmethods_amd.train import Trainer
    from tests._util import separate_relu_units
    x, nf, lab = O.make_synthetic_batch(B, 300, 1152, cfg.vocab_size, seed=seed)
    p = {k: v.double() for k, v in O.init_params(cfg, 1152, seed=1000 + seed).items()}
    if scale_hidden1:
        p = _well_conditioned(p)
    dm64 = None if dropout_masks is None else {k: v.double() for k, v in dropout_masks.items()}
    if prepare_relu:
        p, report = separate_relu_units(p, [(x.double(), nf, dm64)], cfg)
    at_risk = {}
    if not prepare_relu:
        # the weights as initialised: find the ReLU units within fp32 rounding of zero instead of moving them.  Such a unit takes its mask
        # from the last bit of whichever arithmetic computed it; its own column of the layer's kernel / bias gradient is then a coin
        # toss between two correct answers and is left out of that variable's comparison -- everything else is held to the tolerance.
        at_risk = _relu_units_at_rounding_distance(p, x.double(), nf, cfg, dm64)
        report = {k: (f"{len(v[0])} of {v[1]} at rounding distance",) for k, v in at_risk.items()}
    with torch.no_grad():
        _, inter = O.model_forward(p, x.double(), nf, cfg, True, None, dm64, return_intermediates=True)
    pred, loss, grads, _ = O.loss_and_grads(p, x.double(), nf, lab, cfg, dm64)
    tr = Trainer(registry.get_model(name), vocab_size=cfg.vocab_size, batch_size=B, base_learning_rate=cfg.base_learning_rate, device=dev,
                 model_kwargs=dict(iterations=cfg.iterations, cluster_size=cfg.cluster_size, hidden_size=cfg.hidden_size, **kw))
    tr.build(x, nf, lab)
    tr.store.load({"tower/" + k: v for k, v in p.items()})
    step_kw = {} if dropout_masks is None else {"dropout_masks": {k: v.to(dev) for k, v in dropout_masks.items()}}
    # NetVladV1: the encoder GEMMs of the compared step run in the fp16 two-product operand format, scales measured on this batch
    # (without this call the first steps of a run stay on split-bf16: ops.OperandScales)
    if tr.calibrate_operand_scales(x, nf, lab, **step_kw):
        print(f"[{name}] operand scales calibrated: {len(tr.operand_scales.slots)} sites")
    tr.store.summaries = {}
    out = tr.step(x, nf, lab, **step_kw)
    assert tr.operand_scales is None or not tr.operand_scales.slots or tr.operand_scales.steps_fp16 == 1
    got, tr.store.summaries = tr.store.summaries, None
    K, Ka = cfg.cluster_size, cfg.cluster_size // 4
    errs = {}
    for key in ("input_bn", "vlad_video", "vlad_audio", "vlad", "activation"):
        ref, g = inter[key], got[key].double().cpu()
        if g.dim() == 3:                                   # the App. C5 token view [B, K, D] of the d-major descriptor
            Kk = K if key == "vlad_video" else Ka
            ref = ref.reshape(B, -1, Kk).transpose(1, 2)
        errs[key] = assert_close(g.reshape(ref.shape), ref, what=f"{name} intermediate {key}")
    errs["loss"] = assert_close(out["loss"], loss, tol=1e-4, what="loss")
    errs["predictions"] = assert_close(out["predictions"], pred, what="predictions")
    gscale = max(float(g.abs().max()) for g in grads.values())
    worst = (0.0, "")
    raw_worst, over = (0.0, ""), []
    yard = {}
    if not prepare_relu:
        # the yardstick at an untouched initialisation: the SAME graph evaluated by the oracle in fp32 (what a TF1 fp32 run is) against
        # its fp64 evaluation -- how far two correct fp32-grade evaluations of this ill-conditioned start may be apart
        p32 = {k: v.float() for k, v in p.items()}
        dm32 = None if dm64 is None else {k: v.float() for k, v in dm64.items()}
        _, _, g32, _ = O.loss_and_grads(p32, x.float(), nf, lab, cfg, dm32)
        yard = {n: rel_l2(g32[n].double(), grads[n], floor=1e-4 * gscale * grads[n].numel() ** 0.5) for n in g32}
    for n in O.trainable_names(p, cfg):
        g = tr.gradient("tower/" + n).double().cpu()
        ref = grads[n]
        raw_worst = max(raw_worst, (rel_l2(g, ref, floor=1e-4 * gscale * ref.numel() ** 0.5), n))
        site = n if n in at_risk else (n[:-len("kernel")] + "bias" if n.endswith("/kernel") else None)
        if site in at_risk and len(at_risk[site][0]):
            keep = torch.ones(ref.shape[-1], dtype=torch.bool)
            keep[at_risk[site][0]] = False
            g, ref = g[..., keep], ref[..., keep]
        e = rel_l2(g, ref, floor=1e-4 * gscale * ref.numel() ** 0.5)
        worst = max(worst, (e, n))
        if prepare_relu:
            assert e <= grad_tol, f"{name} gradient {n}: relative L2 error {e:.3e} > {grad_tol:.1e}"
        else:
            over.append((e, n))
    if not prepare_relu:
        # Untouched initialisation: what the prepared weights of the other tests take out of the comparison is MEASURED here and held to
        # a bound.  Two effects: (i) ReLU units within rounding of zero (their own kernel / bias columns are left out above; through
        # the layers below them they still reach every earlier variable), (ii) the saturated head of a freshly initialised model
        # (|activation| ~ 30-70: an absolute logit error of 1e-5 is a relative error of 1e-5 in every 1 - p, amplified through
        # the batch norms of NetVladV2).  Bound: every gradient within the north-star's 1e-3 OR within 3 x the WORST distance of the fp32 oracle (the
        # same graph evaluated in fp32 on the CPU: what a TF1 fp32 run is) from the fp64 oracle on this model; 1e-2 with the at-risk units' own
        # columns included; the variables above 1e-3 are listed.
        above = sorted((e, n) for e, n in over if e > grad_tol)
        print(f"[{name} B={B}] untouched initialisation: {len(over) - len(above)} of {len(over)} gradients within {grad_tol:.0e}; above: "
              f"{[(n, float(f'{e:.2e}')) for e, n in above]}; worst INCLUDING the columns of the at-risk ReLU units {raw_worst[0]:.2e} "
              f"({raw_worst[1]}); at-risk units {({k.split('/')[-2]: len(v[0]) for k, v in at_risk.items()})}")
        yworst = max((v, k) for k, v in yard.items())
        ratio = max((e / max(yard[n], grad_tol / 4), n) for e, n in over)
        print(f"[{name} B={B}] the fp32 ORACLE against the fp64 oracle on the same weights: worst gradient {yworst[0]:.2e} ({yworst[1]}), "
              f"{sum(v > grad_tol for v in yard.values())} of {len(yard)} above {grad_tol:.0e}; the product's error is at most "
              f"{ratio[0]:.2f} x the fp32 oracle's ({ratio[1]})")
        assert raw_worst[0] <= 1e-2, f"gradient {raw_worst[1]}: {raw_worst[0]:.3e} > 1e-2 with the at-risk units' own columns included"
        # (per model, not per variable: an error made in one stream's ill-conditioned batch norms reaches the other stream's gradients
        # through hidden1_bn's batch statistics)
        # NetVladV1 (round 6): the 3 x allowance is for the variables DIRECTLY BEHIND an at-risk ReLU only -- a unit within
        # rounding of zero takes its mask from the last bit of whichever arithmetic computed it, its own columns are left out above, and
        # the layer that feeds it receives the flipped tokens' gradient through one GEMM: FeedForwardNetwork's first layer behind the
        # second one's ReLU (transformer_utils.py:701-711), the attention block's LayerNorm behind the first one's.  Every other variable is
        # held to max(1e-3, 1.5 x the fp32 oracle's worst) -- the allowance cannot absorb an arithmetic choice elsewhere.  Measured: the two
        # variables above 1e-3 are video_attention/filter_outputencode1/{kernel, bias} = 1.15e-3 / 1.12e-3, behind ff_outputencode1's 21 at-risk
        # units; with that kernel's weight gradient on two terms (LPM_DW_TERMS_FFN1=2) 1.14e-3 -- it is not the one-term product --, with
        # the dense GEMMs on split-bf16 (LPM_DENSE_ARITHMETIC=bf16x3) every gradient of the model within 1e-3 (worst 5.7e-4).
        behind = set()
        for site, (units, _) in at_risk.items():
            if not len(units):
                continue
            scope, layer = site.rsplit("/", 2)[0], site.rsplit("/", 2)[1]
            if layer.startswith("ff_output"):
                behind |= {f"{scope}/{layer.replace('ff_output', 'filter_output')}/kernel", f"{scope}/{layer.replace('ff_output', 'filter_output')}/bias"}
            elif layer.startswith("filter_output"):
                behind |= {f"{scope}/LayerNorm/gamma", f"{scope}/LayerNorm/beta"}
        for e, n in over:
            # (NetVladV2: 2 x since round 5 -- 3 x before: test_cfg3_untouched_initialisation_family_by_family shows it at 1.28 x with its
            # dense GEMMs on fp16 planes and at 1.00 x with the logits_bn attention in exact fp32)
            k = 2.0 if name == "NetVladV2" else (3.0 if n in behind else 1.5)
            assert e <= max(grad_tol, k * yworst[0]), (f"gradient {n}: {e:.3e} > 1e-3 and > {k:.1f} x the fp32 oracle's own worst distance from "
                                                        f"fp64 on this model ({yworst[0]:.3e}, {yworst[1]})"
                                                        + ("" if name == "NetVladV2" or n in behind else "; not directly behind an at-risk ReLU"))
        if behind:
            print(f"[{name} B={B}] directly behind an at-risk ReLU (3 x allowance): {sorted(behind)}")
    print(f"[{name} B={B}] intermediates {({k: f'{v:.1e}' for k, v in errs.items()})}; worst gradient {worst[0]:.2e} ({worst[1]}); "
          f"ReLU units moved {({k.split('/')[-2]: v[0] for k, v in report.items()})}")
    return errs, worst


def test_cfg2_layer_sizes_reduced_batch():
    """BASELINE configs[1] layer sizes (NetVladV1 K=256, hidden=512, 300 x 1152) with 16 clips -- enough tokens that both encoders
    run as block Functions into the shared descriptor buffer, i.e. the code path of the benchmark: input_bn, both NetVLAD
    descriptors, the encoded descriptor, the gated activation, loss, predictions and all gradients against the fp64 oracle."""
    cfg = O.OracleConfig(model="NetVladV1", iterations=300, cluster_size=256, hidden_size=512, base_learning_rate=2e-4)
    _full_size_compare("NetVladV1", cfg, 16, 0, cuda())


def test_cfg2_reference_initialisation_unscaled():
    """The same with hidden1_weights exactly as the reference initialises it (stddev 1/sqrt(K) on a layer-normed descriptor): the
    model starts saturated, predictions within 1e-9 of 0 or 1.  What the maths allows there: the fused MoE + cross-entropy
    kernel forms 1 - p from its small terms, so loss and d loss / d p carry no cancellation; the activations (|a| ~ 30-70) reach
    the gates as fp32 GEMM results with ~1e-6 relative = ~3e-5 absolute error in the logits, and exp() turns that into a
    3e-5 relative error of every gate / expert probability -- tolerance 1e-3 on the whole-model gradients stays."""
    cfg = O.OracleConfig(model="NetVladV1", iterations=300, cluster_size=256, hidden_size=512, base_learning_rate=2e-4)
    _full_size_compare("NetVladV1", cfg, 8, 2, cuda(), scale_hidden1=False)


@pytest.mark.parametrize("which", ["cfg2", "cfg3", "cfg5"])
def test_untouched_reference_initialisation(which):
    """one case per single-GPU model at the reference's initialisation AS IT IS -- no ReLU-margin preparation of
    the biases (oracle/test_weights.separate_relu_units), no 0.02 scale on hidden1_weights -- at the BASELINE layer sizes: forward
    intermediates, loss and predictions at 1e-3 in max-norm; the gradient of every variable within 1e-3 in Frobenius norm OR within
    twice the worst distance between the oracle's own fp32 and fp64 evaluations of the model's gradients (a freshly initialised model is
    saturated and ill-conditioned: two correct fp32 evaluations differ by more than 1e-3 there, NetVladV2's batch norms most of all),
    with the variables above 1e-3 and the number of ReLU units whose fp64 pre-activation lies within fp32 rounding of zero (the
    units the engineered margins of the other tests move away) printed: the margins are a measured, bounded effect, not a precondition
    of parity."""
    dev = cuda()
    if which == "cfg2":
        cfg = O.OracleConfig(model="NetVladV1", iterations=300, cluster_size=256, hidden_size=512, base_learning_rate=2e-4)
        _full_size_compare("NetVladV1", cfg, 8, 12, dev, scale_hidden1=False, prepare_relu=False)
    elif which == "cfg3":
        cfg = O.OracleConfig(model="NetVladV2", iterations=300, cluster_size=256, hidden_size=512, base_learning_rate=2e-4)
        B = 8
        g = torch.Generator().manual_seed(15)
        masks = {"video": (torch.rand(B, 300, 1024, generator=g) >= 0.9).float(), "audio": (torch.rand(B, 300, 128, generator=g) >= 0.9).float()}
        _full_size_compare("NetVladV2", cfg, B, 13, dev, scale_hidden1=False, prepare_relu=False, dropout_masks=masks)
    else:
        from learnablepoolingmethods_amd import FLAGS
        cfg = O.OracleConfig(model="NetVladV1", vocab_size=3862, base_learning_rate=2e-4, **CFG5)
        FLAGS.moe_num_mixtures = 4
        try:
            _full_size_compare("NetVladV1", cfg, 8, 14, dev, scale_hidden1=False, prepare_relu=False, encoder=False)
        finally:
            FLAGS.reset()


def test_cfg3_untouched_initialisation_family_by_family(monkeypatch):
    """what costs NetVladV2 its factor over the fp32 oracle at the untouched initialisation?  The same case as
    test_untouched_reference_initialisation[cfg3] with each kernel family switched to its most exact form in turn -- the dense GEMMs as
    plain fp32 library GEMMs, the logits_bn attention in exact fp32 MFMA, the pooling (K2 / K3) in exact fp32 MFMA, the dense GEMMs on
    split-bf16 instead of fp16 planes -- and the worst gradient (Frobenius, at-risk ReLU columns left out) printed per setting.  The
    table goes into DESIGN.md section 2; the assertion is the tightened bound: every setting within 2 x the fp32 oracle's own worst
    distance from fp64 (3 x until round 4)."""
    from learnablepoolingmethods_amd import FLAGS, ops
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV2", iterations=300, cluster_size=256, hidden_size=512, base_learning_rate=2e-4)
    B = 8
    g = torch.Generator().manual_seed(15)
    masks = {"video": (torch.rand(B, 300, 1024, generator=g) >= 0.9).float(), "audio": (torch.rand(B, 300, 128, generator=g) >= 0.9).float()}
    settings = [("default (dense on fp16 planes, attention mixed, pooling split-bf16)", {}),
                ("dense GEMMs on split-bf16", {"env": {"LPM_V2_FP16": "0"}}),
                ("dense GEMMs in fp32 (library)", {"flags": {"dense_precision": "f32"}}),
                ("logits_bn attention in exact fp32", {"ops": {"MHA_BN_PRECISION": "f32"}}),
                ("pooling K2 / K3 in exact fp32", {"ops": {"VLAD_PRECISION": "f32"}})]
    rows = []
    for name, sw in settings:
        with monkeypatch.context() as m:
            for k, v in sw.get("env", {}).items():
                m.setenv(k, v)
            for k, v in sw.get("ops", {}).items():
                m.setattr(ops, k, v)
            try:
                for k, v in sw.get("flags", {}).items():
                    setattr(FLAGS, k, v)
                _, worst = _full_size_compare("NetVladV2", cfg, B, 13, dev, scale_hidden1=False, prepare_relu=False, dropout_masks=masks)
            finally:
                FLAGS.reset()
        rows.append((name, worst))
    print("[cfg3 untouched initialisation, family by family] worst gradient (relative L2 to the fp64 oracle; the fp32 oracle's own worst is printed above):")
    for name, (e, n) in rows:
        print(f"    {name}: {e:.2e} ({n})")


def test_cfg3_layer_sizes_reduced_batch():
    """BASELINE configs[2]: NetVladV2 K=256, hidden 512, 300 x 1152, with the reference's dropout ON (rate 0.9,
    transformer_utils.py:450): the masks are drawn here and handed to both sides.  4 clips = 1200 frame tokens per stream: the
    dense layers take the split-bf16 path of the benchmark."""
    cfg = O.OracleConfig(model="NetVladV2", iterations=300, cluster_size=256, hidden_size=512, base_learning_rate=2e-4)
    B = 4
    g = torch.Generator().manual_seed(5)
    masks = {"video": (torch.rand(B, 300, 1024, generator=g) >= 0.9).float(), "audio": (torch.rand(B, 300, 128, generator=g) >= 0.9).float()}
    _full_size_compare("NetVladV2", cfg, B, 1, cuda(), dropout_masks=masks)


def _well_conditioned(p):
    """The reference initialises hidden1_weights with stddev 1/sqrt(K) on a layer-normed (unit-variance)
    descriptor, so fresh models start with |activation| ~ 30-70 and predictions saturated at 1 - 1e-9:
    d loss / d p = 1/(1 - p + 1e-5) is then not representable in fp32 and even the fp32-vs-fp64 CPU oracle
    disagree by 4e-3 on gradients.  Scaling this one tensor keeps the test in the well-conditioned regime
    where a 1e-3 gradient comparison means something."""
    p = dict(p)
    p["hidden1_weights"] = p["hidden1_weights"] * 0.02
    return p


def _train_compare(name, cfg, feat, B, MF, steps, dev, tol=1e-3, stat_tol=1e-3, **kw):
    """``tol`` applies to whole-model gradients in the Frobenius norm of each variable (the north-star's 1e-3).  The seeded
    weights keep every ReLU pre-activation away from zero (oracle/test_weights.separate_relu_units): a unit within fp32 rounding of
    zero takes its mask from the last bit of whichever arithmetic computed it and alone moves the filter_output kernel
    gradient by 1/sqrt(tokens*units) ~ 2e-3 -- round 1 carried a 5e-3 tolerance for that."""
    from learnablepoolingmethods_amd import registry
    from learnablepoolingmethods_amd.train import Trainer
    from tests._util import separate_relu_units
    x, nf, lab = O.make_synthetic_batch(B, MF, feat, cfg.vocab_size, seed=7, min_frames=max(2, MF // 3))
    p = _well_conditioned({k: v.double() for k, v in O.init_params(cfg, feat, seed=1007).items()})
    p, _ = separate_relu_units(p, [(x.double(), nf, None)], cfg)
    tr = Trainer(registry.get_model(name), vocab_size=cfg.vocab_size, batch_size=B, base_learning_rate=cfg.base_learning_rate,
                 learning_rate_decay=cfg.learning_rate_decay, learning_rate_decay_examples=cfg.learning_rate_decay_examples,
                 device=dev, model_kwargs=dict(iterations=cfg.iterations, cluster_size=cfg.cluster_size,
                                               hidden_size=cfg.hidden_size, **kw))
    tr.build(x, nf, lab)
    tr.store.load({"tower/" + k: v for k, v in p.items()})
    st = {"step": 0, "m": {}, "v": {}}
    names = O.trainable_names(p, cfg)
    for s in range(steps):
        p_before = p
        _, _, raw_grads, _ = O.loss_and_grads(p, x.double(), nf, lab, cfg)
        gscale = max(float(g.abs().max()) for g in raw_grads.values())   # scale of "a gradient" in this model
        out = tr.step(x, nf, lab)
        p, st, info = O.train_step(p, st, x.double(), nf, lab, cfg, 1)
        assert_close(out["loss"], info["loss"], tol=1e-4, what=f"step {s} loss")
        assert_close(out["predictions"], info["predictions"], what=f"step {s} predictions")
        for n in names:
            g = tr.gradient("tower/" + n)      # raw (pre-clip) gradient of this step
            if s == 0:   # later steps start from weights that already differ by Adam's sign noise (see below)
                e = rel_l2(g, raw_grads[n], floor=1e-4 * gscale * raw_grads[n].numel() ** 0.5)
                assert e <= tol, f"step {s} gradient {n}: relative L2 error {e:.3e} > {tol:.1e}"
            # Adam's first steps move every element by ~lr * sign(g): only elements whose gradient is well above
            # fp32 noise have a reproducible sign, compare the update on those.
            gr = raw_grads[n]
            mask = gr.abs() > max(1e-3 * float(gr.abs().max()), 1e-4 * gscale)
            got = tr.store.vars["tower/" + n].detach().double().cpu()
            dw_got, dw_ref = (got - p_before[n])[mask], (p[n] - p_before[n])[mask]
            if mask.any() and s == 0:
                e = rel_l2(dw_got, dw_ref)
                assert e <= 1e-2, f"step {s} update {n}: relative L2 error {e:.3e}"
    for n in p:
        if n.endswith("moving_mean") or n.endswith("moving_variance"):
            # (the batch mean of the logits of batch-normalised frames is zero up to rounding: an absolute floor for it)
            assert_close(tr.store.vars["tower/" + n], p[n], tol=stat_tol, what=f"moving stat {n}", floor=1e-6)


def test_train_steps_cfg1_v1():
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV1", iterations=30, cluster_size=16, hidden_size=128, base_learning_rate=1e-3)
    _train_compare("NetVladV1", cfg, 1024, 8, 30, 2, dev)


def test_train_steps_v1_audio():
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV1", iterations=12, cluster_size=16, hidden_size=32, vocab_size=40,
                         base_learning_rate=1e-3)
    _train_compare("NetVladV1", cfg, 1152, 4, 16, 2, dev)


def test_train_steps_v1_relu6_and_remove_diag():
    """The non-default branches of the shared tail: hidden1_bn + relu6 instead of the bias (netvlad_relu,
    frame_level_models.py:2321-2337) and context gating with the diagonal removed (gating_remove_diag, :2349-2352)."""
    from learnablepoolingmethods_amd import FLAGS
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV1", iterations=12, cluster_size=16, hidden_size=32, vocab_size=40,
                         base_learning_rate=1e-3, relu=True, remove_diag=True, encoder=False)
    FLAGS.netvlad_relu, FLAGS.gating_remove_diag = True, True
    try:     # (without the cluster encoders: this case is about the tail)
        _train_compare("NetVladV1", cfg, 1152, 4, 16, 2, dev, encoder=False)
    finally:
        FLAGS.reset()


@pytest.mark.parametrize("low_rank,prob", [(8, False), (-1, True), (8, True)])
def test_train_steps_moe_low_rank_and_probability_gating(low_rank, prob):
    """MoeModel's optional branches through the trainer (video_level_models.py:94-108,128-156): two-layer low-rank gates (both with
    their L2 regularisers, which enter as gradients) and probability gating behind the fused mixture kernel -- two optimiser steps
    against the oracle."""
    from learnablepoolingmethods_amd import FLAGS
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV1", iterations=12, cluster_size=16, hidden_size=32, vocab_size=40, base_learning_rate=1e-3,
                         encoder=False, moe_low_rank_gating=low_rank, moe_prob_gating=prob, moe_l2=1e-3)
    FLAGS.moe_low_rank_gating, FLAGS.moe_prob_gating, FLAGS.moe_l2 = low_rank, prob, 1e-3
    try:
        # (moving statistics after the SECOND step at 5e-3: gating_prob_bn normalises p W, a sum over the 40 class probabilities, and the
        # second step starts from weights that differ between any two implementations by Adam's sign noise of +-lr on elements whose
        # gradient is at rounding level -- 3e-3 of a low-rank gates weight; measured 1.6e-3 on its moving mean, 1e-3-level elsewhere)
        _train_compare("NetVladV1", cfg, 1152, 4, 16, 2, dev, stat_tol=5e-3, encoder=False)
    finally:
        FLAGS.reset()


def test_train_steps_v2():
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV2", iterations=12, cluster_size=16, hidden_size=32, vocab_size=40,
                         base_learning_rate=1e-3, v2_dropout_rate=0.0)
    _train_compare("NetVladV2", cfg, 1152, 4, 16, 2, dev, dropout_rate=0.0)


def test_train_step_accepts_quantised_reader_output():
    """The trainer fed with the reader's uint8 frames (dequantise + pad + L2-normalise fused on the device) takes the same
    step as when fed the reference's float matrix."""
    from learnablepoolingmethods_amd import registry, utils
    from learnablepoolingmethods_amd.train import Trainer
    dev = cuda()
    B, MF = 4, 30
    g = torch.Generator().manual_seed(5)
    q = torch.randint(0, 256, (B, MF, 1152), generator=g, dtype=torch.uint8)
    nf = torch.tensor([30, 12, 25, 7], dtype=torch.int32)
    lab = torch.zeros(B, 40)
    lab[torch.arange(B), torch.tensor([1, 5, 9, 30])] = 1.0
    t = torch.arange(MF).view(1, -1, 1)
    raw = torch.where(t < nf.view(-1, 1, 1), utils.Dequantize(q.float()), torch.zeros(()))
    outs = []
    for x in (q, raw):
        tr = Trainer(registry.get_model("NetVladV1"), vocab_size=40, batch_size=B, base_learning_rate=1e-3, device=dev, seed=3,
                     model_kwargs=dict(iterations=30, cluster_size=32, hidden_size=64))
        outs.append(tr.step(x, nf, lab))
    # the two paths round the dequantised value differently (one fma on the device vs multiply + add on the host): 1e-7 on the
    # inputs, amplified by the freshly initialised (saturating) network
    assert_close(outs[0]["loss"], outs[1]["loss"].double().cpu(), 1e-4, "loss")
    assert_close(outs[0]["predictions"], outs[1]["predictions"].double().cpu(), 1e-3, "predictions")


def test_willow_model_reg_forward_backward():
    """WillowModelReg (frame_level_models.py:2516-2635): random frame sampling with given uniform draws,
    NetVladOrthoReg on both streams (2-D cluster_weights2, orthogonality penalty in the loss), shared tail -- predictions,
    loss and whole-model gradients against the fp64 oracle."""
    from learnablepoolingmethods_amd import registry
    from learnablepoolingmethods_amd.train import Trainer
    dev = cuda()
    cfg = O.OracleConfig(model="WillowModelReg", iterations=12, cluster_size=32, hidden_size=32, vocab_size=40, base_learning_rate=1e-3)
    B, MF = 4, 16
    x, nf, lab = O.make_synthetic_batch(B, MF, 1152, cfg.vocab_size, seed=9, min_frames=5)
    p = _well_conditioned({k: v.double() for k, v in O.init_params(cfg, 1152, seed=1011).items()})
    u = torch.rand(B, cfg.iterations, generator=torch.Generator().manual_seed(4))
    tr = Trainer(registry.get_model("WillowModelReg"), vocab_size=cfg.vocab_size, batch_size=B, base_learning_rate=1e-3, device=dev,
                 model_kwargs=dict(iterations=cfg.iterations, cluster_size=cfg.cluster_size, hidden_size=cfg.hidden_size,
                                   frame_uniform=u))
    tr.build(x, nf, lab)
    expected = sorted(n for n in p if "/cluster_biases" not in n)          # the bias variables exist only without batch norm
    assert sorted(n[len("tower/"):] for n in tr.store.vars) == expected, "variable names follow the reference"
    tr.store.load({"tower/" + k: v for k, v in p.items()})
    pred, loss, grads, _ = O.loss_and_grads(p, x.double(), nf, lab, cfg, dropout_masks={"frame_uniform": u})
    gscale = max(float(g.abs().max()) for g in grads.values())
    out = tr.step(x, nf, lab)
    assert_close(out["loss"], loss, tol=1e-4, what="loss")
    assert_close(out["predictions"], pred, what="predictions")
    for n in O.trainable_names(p, cfg):
        g = tr.gradient("tower/" + n)
        e = rel_l2(g, grads[n], floor=1e-4 * gscale * grads[n].numel() ** 0.5)
        assert e <= 1e-3, f"gradient {n}: relative L2 error {e:.3e}"


def test_training_is_bitwise_reproducible_across_runs():
    """Two trainers from the same seed on the same batch take bit-identical steps: the two-stream schedule (audio branch on a
    second HIP stream), the shared gradient buffers and the arena gather leave no ordering-dependent result behind."""
    from learnablepoolingmethods_amd import registry
    from learnablepoolingmethods_amd.train import Trainer
    dev = cuda()
    B, MF = 6, 40
    x, nf, lab = O.make_synthetic_batch(B, MF, 1152, 50, seed=21, min_frames=10)
    finals = []
    for _ in range(2):
        tr = Trainer(registry.get_model("NetVladV1"), vocab_size=50, batch_size=B, base_learning_rate=1e-3, device=dev, seed=11,
                     model_kwargs=dict(iterations=32, cluster_size=64, hidden_size=64))
        losses = [tr.step(x, nf, lab)["loss"].item() for _ in range(4)]
        torch.cuda.synchronize()
        finals.append((losses, tr.arena.param.clone(), tr.arena.m.clone()))
    assert finals[0][0] == finals[1][0], f"losses differ: {finals[0][0]} vs {finals[1][0]}"
    assert torch.equal(finals[0][1], finals[1][1]) and torch.equal(finals[0][2], finals[1][2])


def test_block_functions_and_descriptor_slots_do_not_change_the_step():
    """The encoder as block Functions writing into the shared descriptor buffer (the default at production sizes) against
    the layer-by-layer graph with a real concat: same loss, same gradient arena to summation-order noise, and the default
    path is itself bitwise repeatable with the audio branch on its own stream."""
    from learnablepoolingmethods_amd import FLAGS, ops, registry
    from learnablepoolingmethods_amd.train import Trainer
    dev = cuda()
    B, MF = 16, 40                      # 16 x 256 and 16 x 64 tokens: both encoders take the split-GEMM / block path
    x, nf, lab = O.make_synthetic_batch(B, MF, 1152, 50, seed=23, min_frames=10)
    res = []
    for fused, slots in ((True, True), (True, True), (True, False), (False, False)):
        FLAGS.fused_encoder_blocks, FLAGS.descriptor_slots = fused, slots
        # every variant on the 128 x 128 form of K2: the layer-by-layer graph has no lazy descriptor and therefore no clip-wide K2,
        # whose assignment sums are added in another order -- a last-bit difference of the descriptor (2e-7) that this toy head
        # (saturated predictions) turns into 5e-4 of every gradient (test_lazily_normalised_descriptor_... measures exactly that)
        clip0, ops.VLAD_CLIP = ops.VLAD_CLIP, False
        try:
            tr = Trainer(registry.get_model("NetVladV1"), vocab_size=50, batch_size=B, base_learning_rate=1e-3, device=dev, seed=13,
                         model_kwargs=dict(iterations=32, cluster_size=256, hidden_size=64))
            loss = tr.step(x, nf, lab)["loss"].item()
            torch.cuda.synchronize()
            res.append((loss, tr.arena.grad.clone(), tr.arena.param.clone(), tr.predict(x, nf).clone()))
        finally:
            FLAGS.reset()
            ops.VLAD_CLIP = clip0
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    for other in res[2:]:
        assert abs(other[0] - res[0][0]) <= 1e-6 * abs(res[0][0])
        assert rel_l2(other[1], res[0][1]) < 2e-5
        assert rel_l2(other[3], res[0][3]) < 2e-5           # the inference-mode forward (no autograd) takes the same paths


def test_logits_bn_attention_gradient_image_does_not_change_the_step():
    """MultiHeadAttentionBN's q/k/v projections + logits_bn attention as one node whose backward writes [dq | dk | dv] as the projections'
    operand image (ops._QKVAttnBNX3, round 6) against the two-node graph with the fp32 gradient and the operand-split pass: the image is
    the split of the same fp32 numbers, so four steps -- two on split-bf16 operands, two on the fp16 planes with their recorded scales --
    end in the same parameters bit for bit."""
    from learnablepoolingmethods_amd import ops, registry
    from learnablepoolingmethods_amd.train import Trainer
    dev = cuda()
    B, MF = 8, 150                      # 1 200 frame tokens: the encoders' dense layers take the split-GEMM path
    x, nf, lab = O.make_synthetic_batch(B, MF, 1152, 50, seed=29, min_frames=100)
    res = []
    for image in (True, False, True):
        flag0, ops.MHA_BN_GRAD_IMAGE = ops.MHA_BN_GRAD_IMAGE, image
        blk0, ops.ATTN_BLOCK_BN = ops.ATTN_BLOCK_BN, False      # (the whole-attention-half node adds dz in a GEMM's beta = 1: another rounding)
        torch.manual_seed(77)               # the encoders' dropout masks (rate 0.9) come from the global generator
        try:
            tr = Trainer(registry.get_model("NetVladV2"), vocab_size=50, batch_size=B, base_learning_rate=1e-3, device=dev, seed=13,
                         model_kwargs=dict(iterations=MF, cluster_size=32, hidden_size=64))
            losses = [tr.step(x, nf, lab)["loss"].item() for _ in range(4)]
            torch.cuda.synchronize()
            res.append((losses, tr.arena.grad.clone(), tr.arena.param.clone()))
        finally:
            ops.MHA_BN_GRAD_IMAGE, ops.ATTN_BLOCK_BN = flag0, blk0
    assert all(l == l for l in res[0][0])
    for other in res[1:]:
        assert other[0] == res[0][0]
        assert torch.equal(other[1], res[0][1]) and torch.equal(other[2], res[0][2])


def test_v2_attention_half_as_one_node_does_not_change_the_step():
    """TransformerEncoderMod's attention half as ONE autograd node (ops._AttnBlockBNX3, round 6: q/k/v GEMM, logits_bn attention,
    attention_bn + output transform, bias + dropout + residual layer norm) against the node-by-node graph: the same kernels; the residual's
    gradient enters the q/k/v input-gradient GEMM as its beta = 1 operand instead of through an add pass -- one rounding placed differently.
    First step: identical loss (the forward is the same code), gradients to fp32 rounding; after four steps the parameters agree to the few
    elements whose Adam step changes sign within that noise."""
    from learnablepoolingmethods_amd import ops, registry
    from learnablepoolingmethods_amd.train import Trainer
    dev = cuda()
    B, MF = 8, 150
    x, nf, lab = O.make_synthetic_batch(B, MF, 1152, 50, seed=31, min_frames=100)
    res = []
    for block in (True, False):
        blk0, ops.ATTN_BLOCK_BN = ops.ATTN_BLOCK_BN, block
        puts0 = ops.GradJoin.puts
        torch.manual_seed(79)
        try:
            tr = Trainer(registry.get_model("NetVladV2"), vocab_size=50, batch_size=B, base_learning_rate=1e-3, device=dev, seed=13,
                         model_kwargs=dict(iterations=MF, cluster_size=32, hidden_size=64))
            first = tr.step(x, nf, lab)["loss"].item()
            g1 = tr.arena.grad.clone()
            losses = [first] + [tr.step(x, nf, lab)["loss"].item() for _ in range(3)]
            torch.cuda.synchronize()
            res.append((losses, g1, tr.arena.param.clone()))
            # both streams' frames are read by their encoder and by their aggregation: two joins per step with the node, none without
            assert ops.GradJoin.puts - puts0 == (8 if block else 0)
        finally:
            ops.ATTN_BLOCK_BN = blk0
    (la, ga, pa), (lb, gb, pb) = res
    assert la[0] == lb[0], "the forward is the same kernels in the same order"
    assert rel_l2(ga, gb) < 2e-6
    # ... and without ops.GradJoin (the aggregation's gradient of the frames back through autograd's add) the node alone
    join0, ops.GRAD_JOIN = ops.GRAD_JOIN, False
    torch.manual_seed(79)
    try:
        tr = Trainer(registry.get_model("NetVladV2"), vocab_size=50, batch_size=B, base_learning_rate=1e-3, device=dev, seed=13,
                     model_kwargs=dict(iterations=MF, cluster_size=32, hidden_size=64))
        assert tr.step(x, nf, lab)["loss"].item() == la[0]
        assert rel_l2(tr.arena.grad, ga) < 2e-6 and rel_l2(tr.arena.grad, gb) < 2e-6
    finally:
        ops.GRAD_JOIN = join0
    for a, b in zip(la, lb):
        assert abs(a - b) <= 1e-4 * abs(b)
    assert rel_l2(pa, pb) < 5e-3


def test_input_bn_gradient_shortcut_matches_the_input_gradient_path():
    """NetVladV1 training never forms the [B*S, 1152] input gradient: input_bn's gamma / beta gradients come in closed form
    from K3's by-products (ops._NetVLAD.backward).  Against the explicit path (dx GEMMs + frame pass) on the same step, and
    against the oracle through the usual train-step comparison at a shape where the shortcut is active (K, K/4 multiples
    of 32)."""
    from learnablepoolingmethods_amd import FLAGS, registry
    from learnablepoolingmethods_amd.train import Trainer
    dev = cuda()
    B, MF = 6, 40
    x, nf, lab = O.make_synthetic_batch(B, MF, 1152, 50, seed=29, min_frames=10)
    res = []
    for on in (True, False):
        FLAGS.input_bn_grad_shortcut = on
        try:
            tr = Trainer(registry.get_model("NetVladV1"), vocab_size=50, batch_size=B, base_learning_rate=1e-3, device=dev, seed=17,
                         model_kwargs=dict(iterations=32, cluster_size=128, hidden_size=64))
            tr.build(x, nf, lab)
            with torch.no_grad():          # non-trivial affine parameters (the reference initialises gamma = 1, beta = 0)
                g = torch.Generator(device=dev).manual_seed(3)
                tr.store.vars["tower/input_bn/gamma"].copy_(1 + 0.6 * (torch.rand(1152, device=dev, generator=g) - 0.5))
                tr.store.vars["tower/input_bn/beta"].copy_(0.2 * torch.randn(1152, device=dev, generator=g))
            loss = tr.step(x, nf, lab)["loss"].item()
            grads = {n: tr.gradient(n).clone() for n in tr.arena.names}
            res.append((loss, grads))
        finally:
            FLAGS.reset()
    assert res[0][0] == res[1][0]
    for n in res[0][1]:
        tol = 2e-4 if "input_bn" in n else 1e-6
        assert rel_l2(res[0][1][n], res[1][1][n]) < tol, f"{n}: {rel_l2(res[0][1][n], res[1][1][n]):.3e}"
    cfg = O.OracleConfig(model="NetVladV1", iterations=24, cluster_size=128, hidden_size=32, vocab_size=40, base_learning_rate=1e-3)
    _train_compare("NetVladV1", cfg, 1152, 4, 30, 1, dev)
    # a gamma element within rounding of zero: the watch on min |gamma| switches the model to the explicit path (with a warning)
    tr = Trainer(registry.get_model("NetVladV1"), vocab_size=50, batch_size=B, base_learning_rate=1e-3, device=dev, seed=17,
                 model_kwargs=dict(iterations=32, cluster_size=128, hidden_size=64))
    tr.build(x, nf, lab)
    with torch.no_grad():
        tr.store.vars["tower/input_bn/gamma"][5] = 1e-6
    with pytest.warns(UserWarning, match="closed-form gamma / beta gradients are switched off"):
        out = tr.step(x, nf, lab)
    a0, _ = tr.arena.segment("tower/input_bn/gamma")
    assert torch.isfinite(out["loss"]) and torch.isfinite(tr.arena.grad[a0:a0 + 1152]).all()
    assert tr.store.vars["tower/input_bn/gamma"]._lpm_gamma_watch.disabled


@pytest.mark.parametrize("B", [1, 13])
def test_ragged_batch_sizes_through_one_trainer(B):
    """Batch sizes that are not multiples of any tile (a single clip; 13 clips, then the 10-clip tail of an epoch through the
    same trainer): every shape-dependent path (skinny weight gradient, block Functions, descriptor slots, closed-form input_bn
    gradients) either applies or falls back, losses and inference outputs stay finite and the step matches the oracle's."""
    from learnablepoolingmethods_amd import registry
    from learnablepoolingmethods_amd.train import Trainer
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV1", iterations=20, cluster_size=64, hidden_size=48, vocab_size=60, base_learning_rate=1e-3)
    x, nf, lab = O.make_synthetic_batch(B, 30, 1152, 60, seed=40 + B, min_frames=5)
    tr = Trainer(registry.get_model("NetVladV1"), vocab_size=60, batch_size=B, base_learning_rate=1e-3, device=dev, seed=3,
                 model_kwargs=dict(iterations=20, cluster_size=64, hidden_size=48))
    tr.build(x, nf, lab)
    p = _well_conditioned({k: v.double() for k, v in O.init_params(cfg, 1152, seed=77).items()})
    tr.store.load({"tower/" + k: v for k, v in p.items()})
    out = tr.step(x, nf, lab)
    if B > 1:     # (one clip: every batch norm sees a single row, the reference itself is degenerate there)
        _, _, info = O.train_step(p, {"step": 0, "m": {}, "v": {}}, x.double(), nf, lab, cfg, 1)
        assert_close(out["loss"], info["loss"], tol=1e-4, what="loss")
        assert_close(out["predictions"], info["predictions"], what="predictions")
        tail = tr.step(x[:B - 3], nf[:B - 3], lab[:B - 3])
        assert torch.isfinite(tail["loss"]) and tail["predictions"].shape == (B - 3, 60)
    assert torch.isfinite(out["loss"]) and torch.isfinite(tr.predict(x, nf)).all()


def test_checkpoint_resume_and_inference_csv(tmp_path):
    """Save after two steps, restore into a fresh trainer: the third step is bit-identical to the uninterrupted run
    (variables, Adam slots, global_step under the reference's TF names).  Then reader -> predict -> CSV end to end."""
    import io
    from learnablepoolingmethods_amd import inference, readers, registry
    from learnablepoolingmethods_amd.train import Trainer
    dev = cuda()
    B, MF = 4, 20
    rng = np.random.default_rng(3)
    recs = []
    for i in range(B):
        n = int(rng.integers(8, MF + 1))
        recs.append(readers.make_sequence_example(f"vid{i}", [int(rng.integers(0, 30))],
                                                  {"rgb": rng.integers(0, 256, (n, 1024), dtype=np.uint8),
                                                   "audio": rng.integers(0, 256, (n, 128), dtype=np.uint8)}))
    path = str(tmp_path / "t.tfrecord")
    readers.write_tfrecord(path, recs)
    reader = readers.YT8MFrameFeatureReader(num_classes=30, max_frames=MF)
    (ids, q, y, nf), = list(reader.batches([path], batch_size=B))
    kw = dict(vocab_size=30, batch_size=B, base_learning_rate=1e-3, device=dev, seed=5,
              model_kwargs=dict(iterations=16, cluster_size=32, hidden_size=32))
    a = Trainer(registry.get_model("NetVladV1"), **kw)
    for _ in range(2):
        a.step(q, nf, y.float())
    ck = str(tmp_path / "ckpt.pt")
    a.save(ck)
    ref = a.step(q, nf, y.float())
    b = Trainer(registry.get_model("NetVladV1"), **{**kw, "seed": 99})      # different init: everything must come from the file
    b.build(q, nf, y.float())
    b.restore(ck)
    got = b.step(q, nf, y.float())
    assert got["global_step"] == ref["global_step"] == 3
    assert torch.equal(got["loss"], ref["loss"]) and torch.equal(got["predictions"], ref["predictions"])
    assert torch.equal(a.arena.param, b.arena.param)
    sd = a.state_dict()
    assert "tower/video_VLAD/cluster_weights" in sd and "tower/hidden1_weights/Adam_1" in sd and "tower/input_bn/moving_mean" in sd
    out = io.StringIO()
    assert inference.write_predictions(out, a, reader.batches([path], batch_size=3), top_k=5) == B
    lines = out.getvalue().splitlines()
    assert lines[0] == "VideoId,LabelConfidencePairs" and len(lines) == B + 1
    assert lines[1].startswith("vid0,") and len(lines[1].split(",")[1].split()) == 10


# ---- BASELINE configs[4]: "Gated NetVLAD K=512 + MoE-4 classifier, 300x1152 bf16, bs=1024 on 8xMI355X" = 128 clips per GPU ---------
CFG5 = dict(iterations=300, cluster_size=512, hidden_size=1024, moe_num_mixtures=4, encoder=False)
CFG5_FWD_TOL = 2e-2      # bf16 storage of frames / logits / assignment / descriptor against the exact fp64 oracle (max-norm relative);
CFG5_GRAD_TOL = 3e-2     # whole-model gradients, Frobenius norm per variable.  Measured: descriptors 7e-3, gradients <= 7.5e-3 (printed)


def _cfg5_trainer(B, dev, storage):
    from learnablepoolingmethods_amd import FLAGS, registry
    from learnablepoolingmethods_amd.train import Trainer
    FLAGS.moe_num_mixtures, FLAGS.netvlad_storage = 4, storage
    return Trainer(registry.get_model("NetVladV1"), vocab_size=3862, batch_size=B, base_learning_rate=2e-4, device=dev,
                   model_kwargs=dict(iterations=300, cluster_size=512, hidden_size=1024, encoder=False))


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_cfg5_layer_sizes_reduced_batch(storage):
    """cfg-5 at its real layer sizes (K = 512 / 128, hidden 1024, MoE-4, 300 x 1152, no cluster encoders) with 8 clips: intermediates,
    loss, predictions and the gradient of every variable against the fp64 oracle -- with fp32 storage to the usual 1e-3, with the
    bf16 storage the configuration names to the documented bf16 tolerance."""
    from learnablepoolingmethods_amd import FLAGS
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV1", vocab_size=3862, base_learning_rate=2e-4, **CFG5)
    B = 8
    x, nf, lab = O.make_synthetic_batch(B, 300, 1152, cfg.vocab_size, seed=5)
    p = _well_conditioned({k: v.double() for k, v in O.init_params(cfg, 1152, seed=1005).items()})
    with torch.no_grad():
        _, inter = O.model_forward(p, x.double(), nf, cfg, True, None, None, return_intermediates=True)
    pred, loss, grads, _ = O.loss_and_grads(p, x.double(), nf, lab, cfg)
    try:
        tr = _cfg5_trainer(B, dev, storage)
        tr.build(x, nf, lab)
        tr.store.load({"tower/" + k: v for k, v in p.items()})
        tr.store.summaries = {}
        out = tr.step(x, nf, lab)
        got, tr.store.summaries = tr.store.summaries, None
    finally:
        FLAGS.reset()
    ftol, gtol = (1e-3, 1e-3) if storage == "f32" else (CFG5_FWD_TOL, CFG5_GRAD_TOL)
    errs = {k: assert_close(got[k].float().reshape(inter[k].shape), inter[k], tol=ftol, what=f"cfg-5 {storage} {k}")
            for k in ("vlad_video", "vlad_audio", "vlad", "activation")}
    errs["loss"] = assert_close(out["loss"], loss, tol=ftol, what="loss")
    errs["predictions"] = assert_close(out["predictions"], pred, tol=ftol, what="predictions")
    gscale = max(float(g.abs().max()) for g in grads.values())
    worst = (0.0, "")
    for n in O.trainable_names(p, cfg):
        e = rel_l2(tr.gradient("tower/" + n), grads[n], floor=1e-4 * gscale * grads[n].numel() ** 0.5)
        worst = max(worst, (e, n))
        assert e <= gtol, f"cfg-5 {storage} gradient {n}: relative L2 error {e:.3e} > {gtol:.1e}"
    print(f"[cfg-5 {storage} B={B}] " + ", ".join(f"{k}: {v:.1e}" for k, v in errs.items()) + f"; worst gradient {worst[0]:.2e} ({worst[1]})")


def test_cfg5_keeps_a_bf16_compute_copy_of_hidden1_weights():
    """SURVEY section 7 hard part 2 ("keep master fp32 + bf16 compute copy"), BASELINE configs[4]: with bf16 storage and the factored
    update the trainer attaches ops.ComputeCopy to hidden1_weights; the Adam epilogue rewrites it every step (ONE full rebuild in a run:
    the first use), it is always exactly bf16(master), and three steps with it follow three steps that stream the fp32 weight
    (FLAGS.hidden1_compute_copy = False) within the bf16 tolerance of this configuration."""
    from learnablepoolingmethods_amd import FLAGS
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV1", vocab_size=3862, base_learning_rate=2e-4, **CFG5)
    B = 16
    x, nf, lab = O.make_synthetic_batch(B, 300, 1152, cfg.vocab_size, seed=6)
    p = _well_conditioned({k: v.double() for k, v in O.init_params(cfg, 1152, seed=1006).items()})
    res = {}
    for copy in (True, False):
        try:
            FLAGS.hidden1_compute_copy = copy
            tr = _cfg5_trainer(B, dev, "bf16")
            tr.build(x, nf, lab)
            tr.store.load({"tower/" + k: v for k, v in p.items()})
            losses = [float(tr.step(x, nf, lab)["loss"]) for _ in range(3)]
            W = tr.arena.views["tower/hidden1_weights"]
            if copy:
                assert tr.w16 is not None and tr.w16.refreshes == 1, "the copy is rebuilt once (first use), then kept by the update pass"
                torch.cuda.synchronize()
                assert torch.equal(tr.w16.buf, W.detach().to(torch.bfloat16)), "the copy is bf16(master) after every step"
            else:
                assert tr.w16 is None
            res[copy] = (losses, W.detach().clone(), tr.predict(x, nf).detach().clone() if hasattr(tr, "predict") else None)
        finally:
            FLAGS.reset()
    la, lb = res[True][0], res[False][0]
    print(f"[cfg-5 compute copy] losses with the copy {la}, streaming fp32 {lb}")
    for a, b in zip(la, lb):
        assert abs(a - b) <= CFG5_FWD_TOL * abs(b) + 1e-3
    assert rel_l2(res[True][1], res[False][1]) <= 1e-3, "three Adam steps of 2e-4: the two masters stay together"


def test_cfg5_input_gradient_folded_into_the_update_pass_does_not_change_the_step():
    """FLAGS.hidden1_fold_input_gradient (round 6): with the compute copy and the early update, the projection's input
    gradient comes out of hidden1_weights' update pass (lpm_factored_clip_adam_copy_dx) instead of lpm_proj_dx_w16.  Same operands (dy and
    the old weight rounded once to bf16), another summation order: three steps end within fp32 rounding of the two-pass schedule, the copy is
    still exactly bf16(master), and the update really did run inside the projection's backward."""
    from learnablepoolingmethods_amd import FLAGS
    dev = cuda()
    cfg = O.OracleConfig(model="NetVladV1", vocab_size=3862, base_learning_rate=2e-4, **CFG5)
    B = 16
    x, nf, lab = O.make_synthetic_batch(B, 300, 1152, cfg.vocab_size, seed=8)
    p = _well_conditioned({k: v.double() for k, v in O.init_params(cfg, 1152, seed=1008).items()})
    res = {}
    for fold in (True, False):
        try:
            FLAGS.hidden1_fold_input_gradient = fold
            tr = _cfg5_trainer(B, dev, "bf16")
            tr.build(x, nf, lab)
            tr.store.load({"tower/" + k: v for k, v in p.items()})
            losses, folded = [], []
            for _ in range(3):
                losses.append(float(tr.step(x, nf, lab)["loss"]))
                folded.append(bool(tr.factored.dx_done))
            torch.cuda.synchronize()
            W = tr.arena.views["tower/hidden1_weights"]
            assert folded == [fold] * 3, f"fold={fold}: the projection's backward took the {'other' if fold else 'folded'} route: {folded}"
            assert tr.w16.refreshes == 1 and torch.equal(tr.w16.buf, W.detach().to(torch.bfloat16))
            res[fold] = (losses, tr.arena.param.detach().clone())
        finally:
            FLAGS.reset()
    print(f"[cfg-5 dx fold] losses folded {res[True][0]}, two passes {res[False][0]}")
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) <= 2e-5 * abs(b)
    # (Adam's first steps move every element by ~lr whatever its gradient's size: the few elements whose gradient changes sign within the
    # summation-order noise differ by 2 lr per step -- 1e-6 of the elements at 4e-2 of a typical weight)
    assert rel_l2(res[True][1], res[False][1]) <= 2e-4


def test_cfg5_bf16_storage_survives_the_gamma_watch_switching_off():
    """once min |gamma| of input_bn falls below the watch's floor the closed-form gamma / beta gradients are switched
    off; under bf16 storage (frames written as operand tiles only, no input-gradient path) the step must then fall back to fp32
    storage -- explicit gradient path, same variables -- instead of reaching the pooling op with unmaterialised frames.  The
    fallback step is held to the fp64 oracle at the fp32 tolerance."""
    import warnings
    from learnablepoolingmethods_amd import FLAGS
    from learnablepoolingmethods_amd import frame_level_models as flm
    dev = cuda()
    from learnablepoolingmethods_amd import registry
    from learnablepoolingmethods_amd.train import Trainer
    sizes = dict(iterations=300, cluster_size=512, hidden_size=128, encoder=False)     # bf16 storage: D, K multiples of 128 (audio K/4 = 128)
    cfg = O.OracleConfig(model="NetVladV1", vocab_size=200, base_learning_rate=2e-4, moe_num_mixtures=4, **sizes)
    B = 4
    x, nf, lab = O.make_synthetic_batch(B, 300, 1152, cfg.vocab_size, seed=7)
    p = _well_conditioned({k: v.double() for k, v in O.init_params(cfg, 1152, seed=1007).items()})
    p["input_bn/gamma"] = p["input_bn/gamma"].clone()
    p["input_bn/gamma"][5] = 0.05                    # below _GammaWatch.FLOOR
    pred, loss, grads, _ = O.loss_and_grads(p, x.double(), nf, lab, cfg)
    try:
        FLAGS.moe_num_mixtures = 4
        FLAGS.netvlad_storage = "bf16"
        flm.NetVladV1._warned_bf16_fallback = False
        tr = Trainer(registry.get_model("NetVladV1"), vocab_size=cfg.vocab_size, batch_size=B, base_learning_rate=2e-4, device=dev,
                     model_kwargs=sizes)
        tr.build(x, nf, lab)
        tr.store.load({"tower/" + k: v for k, v in p.items()})
        for v in tr.store.vars.values():
            if hasattr(v, "_lpm_gamma_watch"):
                del v._lpm_gamma_watch
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            out = tr.step(x, nf, lab)
        assert any("fall" in str(m.message) and "fp32 storage" in str(m.message) for m in w), [str(m.message) for m in w]
    finally:
        FLAGS.reset()
    assert_close(out["predictions"], pred, tol=1e-3, what="predictions")
    gscale = max(float(g.abs().max()) for g in grads.values())
    for n in O.trainable_names(p, cfg):
        e = rel_l2(tr.gradient("tower/" + n), grads[n], floor=1e-4 * gscale * grads[n].numel() ** 0.5)
        assert e <= 1e-3, f"gradient {n}: relative L2 error {e:.3e}"


def test_cfg5_full_batch_properties():
    """cfg-5 at its full per-GPU batch (128 clips, bf16 storage), through size-independent properties: every pooled descriptor has
    unit norm and every cluster column norm 1/sqrt(K) (frame_level_models.py:2819-2822) at bf16 resolution, the loss is finite and
    goes down over a few steps on one batch, and the step equals the fp32-storage step of the same trainer state to bf16 accuracy."""
    from learnablepoolingmethods_amd import FLAGS
    dev = cuda()
    B = 128
    x, nf, lab = O.make_synthetic_batch(B, 300, 1152, 3862, seed=6)
    res = {}
    for storage in ("bf16", "f32"):
        try:
            tr = _cfg5_trainer(B, dev, storage)
            tr.store._gen_seed = 77
            tr.build(x, nf, lab)
            with torch.no_grad():
                tr.store.vars["tower/hidden1_weights"].mul_(0.02)
            tr.store.summaries = {}
            losses = [float(tr.step(x, nf, lab)["loss"])]
            got, tr.store.summaries = tr.store.summaries, None
            a0, _ = tr.arena.segment("tower/video_VLAD/cluster_weights")
            res[storage] = (losses[0], got, tr.arena.grad[a0:a0 + 1024 * 512].clone())
            if storage == "bf16":
                for _ in range(3):
                    losses.append(float(tr.step(x, nf, lab)["loss"]))
                assert all(np.isfinite(l) for l in losses) and losses[-1] < losses[0], f"losses {losses}"
                v = got["vlad_video"].float().reshape(B, 1024, 512)
                assert torch.allclose(v.norm(dim=(1, 2)), torch.ones(B, device=dev), atol=4e-3)
                assert torch.allclose(v.norm(dim=1), torch.full((B, 512), 512 ** -0.5, device=dev), atol=2e-3 * 512 ** -0.5 * 8)
            del tr
            torch.cuda.empty_cache()
        finally:
            FLAGS.reset()
    assert abs(res["bf16"][0] - res["f32"][0]) <= CFG5_FWD_TOL * abs(res["f32"][0])
    assert rel_err(res["bf16"][1]["vlad"].float(), res["f32"][1]["vlad"]) <= CFG5_FWD_TOL
    assert rel_l2(res["bf16"][2], res["f32"][2]) <= CFG5_GRAD_TOL


def test_lazily_normalised_descriptor_does_not_change_the_step():
    """NetVladV1's video pooling hands its cluster encoder the un-normalised sums [B, K, D] + one scale per (clip, cluster)
    (FLAGS.netvlad_lazy_descriptor, ops.netvlad(lazy=True)): no finalize pass, no transposes in the pooling backward.  Against the
    materialised path on the same step: same loss, same gradient arena to summation-order noise, same inference output; the
    summaries still show the normalised descriptor.  (test_cfg2_layer_sizes_reduced_batch holds the lazy path to the oracle.)"""
    from learnablepoolingmethods_amd import FLAGS, ops, registry
    from learnablepoolingmethods_amd.train import Trainer
    dev = cuda()
    B, MF = 16, 40
    x, nf, lab = O.make_synthetic_batch(B, MF, 1152, 50, seed=31, min_frames=10)
    res = []
    clip0 = ops.VLAD_CLIP
    for lazy in (True, False, True, "softmax inside K2", "clip-wide K2"):
        FLAGS.netvlad_lazy_descriptor = bool(lazy)
        ops.VLAD_SOFTMAX_FUSED = lazy == "softmax inside K2"
        ops.VLAD_CLIP = lazy == "clip-wide K2"          # (the other four on the 128 x 128 form, whose sums the materialised path shares)
        try:
            tr = Trainer(registry.get_model("NetVladV1"), vocab_size=50, batch_size=B, base_learning_rate=1e-3, device=dev, seed=19,
                         model_kwargs=dict(iterations=48, cluster_size=256, hidden_size=64))      # (48 frames: >= 3 steps for the in-kernel softmax)
            tr.build(x, nf, lab)
            tr.store.summaries = {}
            loss = tr.step(x, nf, lab)["loss"].item()
            summ, tr.store.summaries = tr.store.summaries, None
            torch.cuda.synchronize()
            res.append((loss, tr.arena.grad.clone(), tr.predict(x, nf).clone(), summ["vlad_video"].clone()))
        finally:
            FLAGS.reset()
            ops.VLAD_SOFTMAX_FUSED = False
            ops.VLAD_CLIP = clip0
    # The clip-wide form of K2 (the default) adds the assignment sums in another fixed order: the descriptor agrees to the last bits,
    # the forward to 1e-5 -- and the gradients of THIS toy problem to 2e-3 only: its predictions are saturated, and a 2e-7 perturbation
    # of the descriptor moves every gradient by ~5e-4 (measured the same with round 3's one-launch form, tests/diagnostics/debug_clip.py).  Parity of
    # the default path is held against the oracle (test_cfg2_*), not here.
    # (res[.][2], the predictions AFTER the optimiser step, are not compared: Adam's first step is lr * sign(g))
    assert abs(res[4][0] - res[0][0]) <= 1e-5 * abs(res[0][0])
    assert rel_err(res[4][3], res[0][3]) < 2e-6
    assert rel_l2(res[4][1], res[0][1]) < 2e-3
    assert res[0][0] == res[2][0] and torch.equal(res[0][1], res[2][1]), "the lazy path is bitwise repeatable"
    # ... and with the softmax inside the aggregation kernel (ops.VLAD_SOFTMAX_FUSED) the whole step is the same bits
    assert res[0][0] == res[3][0] and torch.equal(res[0][1], res[3][1]) and torch.equal(res[0][2], res[3][2]), "softmax inside K2"
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[1][0])
    assert rel_l2(res[0][1], res[1][1]) < 2e-5
    assert rel_l2(res[0][2], res[1][2]) < 2e-5
    assert rel_err(res[0][3], res[1][3]) < 1e-6
    v = res[0][3].reshape(B, 256, 1024)                               # normalised: every cluster row has norm 1 / sqrt(K)
    assert torch.allclose(v.norm(dim=2), torch.full((B, 256), 256 ** -0.5, device=dev), atol=1e-6)


def test_trainer_activates_the_recorded_library_gemm_solutions():
    """A CUDA trainer hands PyTorch the recorded fp32 GEMM solutions (FLAGS.library_gemm_selection; TunableOp on, tuning OFF -- nothing is
    searched at run time); a process that configured TunableOp through the environment is left alone."""
    import os
    import torch.cuda.tunable as tunable
    from learnablepoolingmethods_amd import ops, registry, train
    if "PYTORCH_TUNABLEOP_ENABLED" in os.environ or os.environ.get("LPM_LIBRARY_GEMM_SELECTION") == "0":
        pytest.skip("TunableOp configured by the environment")
    train.Trainer(registry.get_model("NetVladV1"), vocab_size=16, batch_size=2, device=cuda(), seed=0,
                  model_kwargs=dict(iterations=4, cluster_size=8, hidden_size=16))
    if not ops._LIBRARY_SELECTION:
        pytest.skip("the recorded solutions' validators do not match this box's libraries: PyTorch ignores the file")
    assert tunable.is_enabled() and not tunable.tuning_is_enabled()
    shapes = {r[1] for r in tunable.get_results()}
    assert any(s.startswith("nn_19310_128_1024") for s in shapes), sorted(shapes)[:5]


def test_a_step_that_fails_behind_the_early_update_poisons_the_trainer(tmp_path):
    """with FLAGS.hidden1_early_update the update of hidden1_weights (~85 % of the parameters at cfg-2) runs INSIDE backward; if
    the rest of the step then raises, hidden1_weights and its Adam moments are at step t + 1 while global_step and every other variable are
    at step t.  The trainer must refuse to go on from there -- step() and state_dict() raise -- until a consistent state is restored; and
    a second use of the weight in one backward, once the early update has consumed the first product, raises instead of being dropped."""
    from learnablepoolingmethods_amd import ops, registry
    from learnablepoolingmethods_amd.train import Trainer
    dev = cuda()
    B = 16
    x, nf, lab = O.make_synthetic_batch(B, 30, 1152, 40, seed=4, min_frames=10)
    tr = Trainer(registry.get_model("NetVladV1"), vocab_size=40, batch_size=B, base_learning_rate=1e-3, device=dev, seed=5,
                 model_kwargs=dict(iterations=30, cluster_size=32, hidden_size=64))
    tr.step(x, nf, lab)
    assert tr.factored is not None, "this test needs the factored route of hidden1_weights (one tower, 16 clips)"
    path = str(tmp_path / "good.pt")
    tr.save(path)
    good_step = tr.global_step
    h1 = tr.arena.views["tower/hidden1_weights"]
    before = h1.detach().clone()
    real_collect = tr.arena.collect

    def failing_collect(*a, **k):
        raise ops.LpmError("injected: a kernel failed behind the projection's backward")
    tr.arena.collect = failing_collect
    with pytest.raises(ops.LpmError, match="injected"):
        tr.step(x, nf, lab)
    tr.arena.collect = real_collect
    torch.cuda.synchronize()
    assert tr.factored.early_done and not torch.equal(h1.detach(), before), "the early update ran: hidden1_weights is one step ahead"
    assert tr.global_step == good_step
    with pytest.raises(RuntimeError, match="inconsistent state"):
        tr.step(x, nf, lab)
    with pytest.raises(RuntimeError, match="inconsistent state"):
        tr.state_dict()
    tr.restore(path)                                   # a consistent state again: training goes on
    out = tr.step(x, nf, lab)
    assert tr.global_step == good_step + 1 and bool(torch.isfinite(out["loss"]))
    # the second use of the weight inside one backward, after the early update took the first product
    fg = tr.factored
    fg.armed, fg.puts, fg.early_done = True, 1, True
    try:
        W = tr.arena.views["tower/hidden1_weights"]
        xx = torch.randn(B, W.shape[0], device=dev, requires_grad=True)
        with pytest.raises(ops.LpmError, match="used twice"):
            ops.projection(xx, W).sum().backward()
    finally:
        fg.armed = False
        fg.clear()
