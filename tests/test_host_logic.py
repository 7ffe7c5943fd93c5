"""-m "not gpu": host-side logic that mirrors the reference's registry / flags / variable scoping /
training-loop arithmetic (no kernels)."""
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from learnablepoolingmethods_amd import FLAGS, registry, train, utils
from learnablepoolingmethods_amd import models as lp_models
from learnablepoolingmethods_amd import variables as vs
from oracle import lpm_oracle as O


def test_registry_finds_reference_class_names():
    for name in ("NetVladV1", "NetVladV2", "MoeModel"):
        cls = registry.find_class_by_name(name)
        assert cls.__name__ == name and issubclass(cls, lp_models.BaseModel)
        cls()                                   # no-arg constructible, as train.py:680-681 needs
        assert registry.validate_class_name(name)
    with pytest.raises(ValueError):
        registry.find_class_by_name("NoSuchModel")
    with pytest.raises(NotImplementedError):
        lp_models.BaseModel().create_model(None)


def test_flag_defaults_match_reference():
    assert FLAGS.netvlad_cluster_size == 256 and FLAGS.netvlad_hidden_size == 1024 and FLAGS.iterations == 30
    assert FLAGS.netvlad_add_batch_norm and FLAGS.gating and not FLAGS.netvlad_relu and not FLAGS.gating_remove_diag
    assert FLAGS.moe_num_mixtures == 2 and FLAGS.moe_l2 == 1e-8
    assert FLAGS.clip_gradient_norm == 1.0 and FLAGS.regularization_penalty == 1.0
    FLAGS.moe_num_mixtures = 4
    FLAGS.reset()
    assert FLAGS.moe_num_mixtures == 2
    with pytest.raises(AttributeError):
        FLAGS.no_such_flag = 1


def test_variable_store_scoping_and_reuse():
    store = vs.VariableStore(device="cpu", seed=3)
    with vs.use_store(store):
        with vs.variable_scope("tower"):
            with vs.variable_scope("video_VLAD"):
                a = vs.get_variable("cluster_weights", [4, 2], vs.random_normal_initializer(0.5))
                b = vs.get_variable("cluster_weights", [4, 2], vs.random_normal_initializer(0.5))   # reuse
            mm = vs.get_variable("moving_mean", [2], vs.zeros_initializer(), trainable=False)
        assert a is b and "tower/video_VLAD/cluster_weights" in store.vars
        assert a.requires_grad and not mm.requires_grad
        assert list(store.trainable_variables()) == ["tower/video_VLAD/cluster_weights"]
        with pytest.raises(ValueError):
            with vs.variable_scope("tower"), vs.variable_scope("video_VLAD"):
                vs.get_variable("cluster_weights", [5, 2], vs.zeros_initializer())
    store.load({"tower/video_VLAD/cluster_weights": torch.ones(4, 2)})
    assert torch.equal(a.detach(), torch.ones(4, 2))


def test_learning_rate_matches_oracle():
    cfg = O.OracleConfig(base_learning_rate=2e-4, learning_rate_decay=0.85, learning_rate_decay_examples=4000000)
    for step in (0, 1, 6249, 6250, 12500, 100000):
        assert train.learning_rate(2e-4, step, 80, 8, 4000000, 0.85) == O.learning_rate(cfg, step, 80, 8)


def test_utils_combine_and_clip_match_oracle():
    g = torch.Generator().manual_seed(0)
    towers = [{"a": torch.randn(5, 3, generator=g) * s, "b": torch.randn(7, generator=g) * 0.01} for s in (1.0, 3.0)]
    got = utils.clip_gradient_norms(utils.combine_gradients(towers), 1.0)
    ref = O.clip_gradient_norms(O.combine_gradients(towers), 1.0)
    for n in ref:
        assert torch.equal(got[n], ref[n])
    assert abs(float(got["a"].norm()) - 1.0) < 1e-6            # clipped
    assert float(got["b"].norm()) < 1.0                         # left alone


def test_dequantize_matches_reference_formula():
    q = torch.arange(256, dtype=torch.float32)
    d = utils.Dequantize(q)
    assert abs(float(d[0]) - (4 / 512 - 2)) < 1e-7 and abs(float(d[255]) - (4 + 4 / 512 - 2)) < 1e-6


def test_parameter_arena_layout_and_views():
    store = vs.VariableStore(device="cpu", seed=0)
    with vs.use_store(store):
        w = vs.get_variable("hidden1_weights", [300, 7], vs.random_normal_initializer(1.0))
        b = vs.get_variable("b", [5], vs.ones_initializer())
        vs.get_variable("moving_mean", [5], vs.zeros_initializer(), trainable=False)
    w0 = w.detach().clone()
    arena = train.ParameterArena(store, first=["b"])
    assert arena.names == ["b", "hidden1_weights"]
    assert all(o % train.ARENA_ALIGN == 0 for o in arena.offsets_host)
    assert arena.total == 2 * train.ARENA_ALIGN
    assert torch.equal(w.detach(), w0) and w.data_ptr() == arena.param[train.ARENA_ALIGN:].data_ptr()
    (w.sum() * 2 + b.sum()).backward()
    a0, _ = arena.segment("hidden1_weights")
    assert torch.equal(arena.grad[a0:a0 + 2100], torch.full((2100,), 2.0))
    assert float(arena.grad[a0 + 2100:].abs().sum()) == 0.0      # padding stays zero
    arena.zero_grad()
    assert float(arena.grad.abs().sum()) == 0.0
    with pytest.raises(RuntimeError):
        with vs.use_store(store):
            vs.get_variable("late", [1], vs.zeros_initializer())


def test_orthogonal_regularizer_known_answers():
    """module_utils.py:55-90: scale * sum |Wn^T Wn - I|, Wn = rows of W normalised over the clusters."""
    from learnablepoolingmethods_amd import module_utils
    from oracle import lpm_oracle as O
    # rows are one-hot -> Wn = W; W^T W = diag(count of rows on each cluster): [[2,0],[0,1]] -> |.-I| sums to 1
    w = torch.tensor([[1.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    assert abs(float(module_utils.orthogonal_regularizer(0.5)(w)) - 0.5) < 1e-7
    assert module_utils.orthogonal_regularizer(0.0)(w) is None
    with pytest.raises(ValueError):
        module_utils.orthogonal_regularizer(1)
    with pytest.raises(ValueError):
        module_utils.orthogonal_regularizer(-0.1)
    g = torch.Generator().manual_seed(0)
    w = torch.randn(24, 6, generator=g, dtype=torch.float64)
    assert abs(float(module_utils.orthogonal_regularizer(1e-4)(w) - O.orthogonal_regularizer(w, 1e-4))) < 1e-15


def test_random_frame_sampling_index_rules():
    """model_utils.py:26-78 with the uniform draws given: frames = int32(u * nf); sequence = clamped contiguous run."""
    from learnablepoolingmethods_amd import model_utils
    x = torch.arange(3 * 10, dtype=torch.float32).reshape(3, 10, 1)
    nf = torch.tensor([10, 4, 7])
    u = torch.tensor([[0.0, 0.5, 0.999], [0.0, 0.5, 0.999], [0.26, 0.5, 0.75]])
    got = model_utils.SampleRandomFrames(x, nf, 3, uniform=u).squeeze(-1)
    assert got.tolist() == [[0.0, 5.0, 9.0], [10.0, 12.0, 13.0], [21.0, 23.0, 25.0]]
    seq = model_utils.SampleRandomSequence(x, nf, 6, uniform=torch.tensor([[0.99], [0.5], [0.5]])).squeeze(-1)
    # clip 0: max start 4 -> int(0.99 * 5) = 4 -> frames 4..9; clip 1: nf 4 < 6 -> start 0, clamped to frame 3; clip 2: start int(0.5*2)=1
    assert seq.tolist() == [[4, 5, 6, 7, 8, 9], [10, 11, 12, 13, 13, 13], [21, 22, 23, 24, 25, 26]]


def test_inference_csv_format():
    """inference.py:88-96,182: header, top-k in descending score order, "%i %g" pairs, byte ids decoded."""
    from learnablepoolingmethods_amd import inference
    p = torch.tensor([[0.1, 0.9, 0.5, 0.0], [0.25, 1e-7, 0.75, 0.123456789]])
    lines = list(inference.format_lines([b"ab12", "cd34"], p, 3))
    assert lines == ["ab12,1 0.9 2 0.5 0 0.1\n", "cd34,2 0.75 0 0.25 3 0.123457\n"]
    assert inference.CSV_HEADER == "VideoId,LabelConfidencePairs\n"
    assert list(inference.format_lines(["x"], p[:1], 10)) == ["x,1 0.9 2 0.5 0 0.1 3 0\n"]       # top_k larger than the vocabulary


def test_attention_modules_host_side_and_variable_names():
    """attention_modules on the CPU: the library-GEMM branches (head width not 8/16, do_shift=False) equal the oracle and the
    variables carry TF1's default layer names; the pooling branch that needs the HIP kernels fails loudly without a GPU."""
    from learnablepoolingmethods_amd import attention_modules
    from learnablepoolingmethods_amd._capi import LpmError
    F, L, heads, B = 12, 6, 2, 3
    x = torch.randn(B * L, F, generator=torch.Generator().manual_seed(0))
    store = vs.VariableStore(device="cpu", seed=2)
    with vs.use_store(store), vs.variable_scope("tower"):
        out = attention_modules.TransformerEncoderBlock(True, F, L, F, heads, 1).forward(x)
    names = set(store.vars)
    assert {"tower/Block1Layer0/dense/kernel", "tower/Block1Layer1/dense_2/bias", "tower/dense/kernel", "tower/conv1d/kernel",
            "tower/conv1d_1/bias", "tower/LayerNorm/gamma", "tower/LayerNorm_1/beta"} <= names
    assert tuple(store.vars["tower/conv1d/kernel"].shape) == (1, F, 4 * F)
    ref = O.transformer_encoder_block(x.double(), {k: v.detach().double() for k, v in store.vars.items()}, "tower", F, L, F,
                                      heads, 1)
    assert out.shape == (B * L, F) and float((out.detach().double() - ref).abs().max()) < 1e-5
    store = vs.VariableStore(device="cpu", seed=2)
    with vs.use_store(store):
        raw = attention_modules.OneFcAttention(F, L, 5, do_shift=False).forward(x)
        ref = O.one_fc_attention_forward(x.double(), {k: v.detach().double() for k, v in store.vars.items()}, "", L, False)
        assert float((raw.detach().double() - ref).abs().max()) < 1e-5
        with pytest.raises(LpmError):
            attention_modules.OneFcAttention(F, L, 5, do_shift=True).forward(x)


@pytest.mark.parametrize("low_rank,prob,inp,remove_diag", [(-1, False, "prob", False), (6, False, "prob", False), (-1, True, "prob", False),
                                                           (6, True, "prob", True), (-1, True, "input", False)])
def test_moe_low_rank_and_probability_gating_branches(low_rank, prob, inp, remove_diag):
    """MoeModel's optional branches (video_level_models.py:37-45,94-108,128-156; off by default): the two-layer low-rank gates and
    the gating of the class probabilities by sigmoid(BN(p W)) / sigmoid(BN(x W)), with the reference's variable names, L2
    regularisers on every gates layer, and gradients -- host-side torch against the oracle's restatement (fp64, CPU)."""
    from learnablepoolingmethods_amd import video_level_models
    torch.manual_seed(3)
    B, H, V, m = 5, 12, 7, 3
    cfg = O.OracleConfig(vocab_size=V, hidden_size=H, moe_num_mixtures=m, moe_low_rank_gating=low_rank, moe_prob_gating=prob,
                         moe_prob_gating_input=inp, remove_diag=remove_diag, moe_l2=1e-2)
    act = torch.randn(B, H, dtype=torch.float64)
    try:
        FLAGS.moe_num_mixtures, FLAGS.moe_low_rank_gating, FLAGS.moe_prob_gating = m, low_rank, prob
        FLAGS.moe_prob_gating_input, FLAGS.gating_remove_diag, FLAGS.moe_l2 = inp, remove_diag, 1e-2
        store = vs.VariableStore(device="cpu", seed=1)
        with vs.use_store(store):
            video_level_models.MoeModel().create_model(act.float(), V, is_training=True)          # creates the variables
            want_names = {"experts/weights", "experts/biases"} | ({"gates/weights"} if low_rank == -1 else {"gates1/weights", "gates2/weights"})
            if prob:
                want_names |= {"gating_prob_weights", "gating_prob_bn/beta", "gating_prob_bn/gamma", "gating_prob_bn/moving_mean",
                               "gating_prob_bn/moving_variance"}
            assert set(store.vars) == want_names
            assert store.vars["gating_prob_weights"].shape == ((V, V) if inp == "prob" else (H, V)) if prob else True
            p = {n: (torch.randn(v.shape, dtype=torch.float64) * 0.3 if v.dim() > 1 or "biases" in n else v.detach().double().clone())
                 for n, v in store.vars.items()}
            store.pop_regularization_losses(), store.pop_l2_regularizers()
            store.vars = {n: p[n].clone().requires_grad_(store.trainable[n]) for n in p}
            out = video_level_models.MoeModel().create_model(act, V, is_training=True)["predictions"]
            regs = store.pop_regularization_losses()       # (a CPU store collects slim.l2_regularizer penalties as loss tensors)
        pt = {n: v.clone().requires_grad_(True) for n, v in p.items()}
        ref = O.moe_forward(act, pt, V, m, cfg, True, {})
        assert torch.allclose(out, ref, rtol=1e-10, atol=1e-12)
        assert len(regs) == (2 if low_rank == -1 else 3)          # one penalty per gates layer + the experts' (:91,99,106,113)
        assert torch.allclose(torch.stack(regs).sum(), O.regularization_loss(pt, cfg), rtol=1e-12)
        w = torch.randn_like(out)
        (out * w).sum().backward()
        (ref * w).sum().backward()
        for n in O.trainable_names(p, cfg):
            assert torch.allclose(store.vars[n].grad, pt[n].grad, rtol=1e-9, atol=1e-12), n
    finally:
        FLAGS.reset()


def test_direct_gradient_slots_in_the_gather_mode_arena():
    """ParameterArena.mark_direct + collect (round 3: the encoders' dense kernels and the MoE matrices): a producer writes the gradient
    into the variable's arena slice itself; collect() zeroes a slot nobody wrote, folds in what autograd accumulated on the side (a second
    use of the weight), adds the analytic L2 penalty coefficient * w, and leaves the gathered variables to gather_names."""
    from learnablepoolingmethods_amd import ops
    store = vs.VariableStore(device="cpu", seed=0)
    with vs.use_store(store):
        w = vs.get_variable("experts/weights", [6, 4], vs.random_normal_initializer(1.0))
        u = vs.get_variable("unused/kernel", [3, 2], vs.random_normal_initializer(1.0))
        b = vs.get_variable("b", [5], vs.ones_initializer())
    arena = train.ParameterArena(store, gather=True)
    arena.mark_direct("experts/weights")
    arena.mark_direct("unused/kernel")
    arena.zero_grad()
    # the producer: ops._grad_slot hands out the free slot once per step
    slot = ops._grad_slot(w)
    assert slot is not None and slot.data_ptr() == arena.grad[arena.segment("experts/weights")[0]:].data_ptr()
    slot.copy_(torch.full((6, 4), 3.0))
    ops._grad_done(w)
    assert ops._grad_slot(w) is None                       # second use in the same step: back through autograd
    w.grad = torch.full((6, 4), 0.5)                       # ... which accumulated this
    (b.sum() * 2).backward()
    arena.l2 = {"experts/weights": 0.1}
    arena.collect()
    a0, _ = arena.segment("experts/weights")
    assert torch.allclose(arena.grad[a0:a0 + 24].view(6, 4), 3.0 + 0.5 + 0.1 * w.detach())
    a1, _ = arena.segment("unused/kernel")
    assert float(arena.grad[a1:a1 + 6].abs().sum()) == 0.0 and u.grad is None
    a2, _ = arena.segment("b")
    assert torch.equal(arena.grad[a2:a2 + 5], torch.full((5,), 2.0)) and b.grad is None
    arena.zero_grad()
    assert ops._grad_slot(w) is not None                   # re-armed for the next step
    old = ops.DIRECT_WGRAD
    try:
        ops.DIRECT_WGRAD = False
        assert ops._grad_slot(w) is None                   # the run-time switch sends everything through autograd
    finally:
        ops.DIRECT_WGRAD = old


def test_split_vector_hands_back_one_concatenated_gradient():
    """ops.split_vector (NetVladV1's input_bn gamma / beta halves): the two views carry the right values and their gradients come back
    as ONE concatenation -- also when only one half was used (the other half's gradient is zero, not missing)."""
    from learnablepoolingmethods_amd import ops
    v = torch.arange(10, dtype=torch.float32, requires_grad=True)
    a, b = ops.split_vector(v, 4)
    assert torch.equal(a, v[:4]) and torch.equal(b, v[4:])
    (a * 2).sum().backward(retain_graph=True)
    assert torch.equal(v.grad, torch.tensor([2.0] * 4 + [0.0] * 6))
    v.grad = None
    ((a * 2).sum() + (b * torch.arange(6.0)).sum()).backward()
    assert torch.equal(v.grad, torch.cat([torch.full((4,), 2.0), torch.arange(6.0)]))


def test_recorded_library_gemm_solutions_file_is_well_formed():
    """FLAGS.library_gemm_selection: the file PyTorch's TunableOp reads (tuning off) -- validators first, then one recorded solution per
    fp32 GEMM shape; the MoE-head shapes of the three benchmark configurations are in it."""
    import os
    from learnablepoolingmethods_amd import ops
    assert FLAGS.library_gemm_selection is True
    lines = [l.strip().split(",") for l in open(ops.LIBRARY_GEMM_FILE) if l.strip()]
    validators = {l[1]: l[2] for l in lines if l[0] == "Validator"}
    assert {"PT_VERSION", "HIP_VERSION", "HIPBLASLT_VERSION", "ROCBLAS_VERSION", "GCN_ARCH_NAME"} <= set(validators)
    assert validators["GCN_ARCH_NAME"].startswith("gfx950")
    entries = [l for l in lines if l[0] != "Validator"]
    assert all(len(l) == 4 and l[0].startswith("Gemm") and float(l[3]) > 0 for l in entries)
    keys = {l[1] for l in entries}
    assert len(keys) == len(entries)
    for shape in ("nn_11586_80_512", "nn_19310_128_1024", "tn_1024_128_19310", "nt_15448_1024_128"):     # MoE gates / experts, cfg-2 and cfg-5
        assert any(k.startswith(shape + "_") for k in keys), shape


def test_operand_sites_carry_the_header_s_format_kinds():
    """ops.OperandSite (host side of csrc/operand_format.h): activations ("a") are three fp16 planes, gradients ("g") two; without fp16 both
    are split-bf16 x3 with scale 1; the enum values are the header's."""
    import re
    from learnablepoolingmethods_amd import _capi, ops
    txt = open(os.path.join(ROOT, "include", "lpm_hip.h")).read()
    m = re.search(r"enum \{ LPM_OPERAND_BF16X3 = (\d+), LPM_OPERAND_FP16X2 = (\d+), LPM_OPERAND_FP16X3 = (\d+) \}", txt)
    assert m and (int(m.group(1)), int(m.group(2)), int(m.group(3))) == (_capi.LPM_OPERAND_BF16X3, _capi.LPM_OPERAND_FP16X2, _capi.LPM_OPERAND_FP16X3)
    a, g, b = ops.OperandSite(True, 1024.0, 0, role="a"), ops.OperandSite(True, 2.0 ** -3, 0, role="g"), ops.OperandSite(False, 8.0, 0, role="g")
    assert (a.kind, a.planes, a.inv) == (_capi.LPM_OPERAND_FP16X3, 3, 1.0 / 1024) and a.struct.scale == 1024.0
    assert (g.kind, g.planes, g.inv) == (_capi.LPM_OPERAND_FP16X2, 2, 8.0)
    assert (b.kind, b.planes, b.scale, b.inv) == (_capi.LPM_OPERAND_BF16X3, 3, 1.0, 1.0)


def test_compute_copy_tracks_every_writer_of_its_master():
    """ops.ComputeCopy (the bf16 compute copy of hidden1_weights): rebuilt at first use, kept while nothing writes the
    master, stale after a write through torch to the variable OR to the flat arena it is a view of (two version counters: the variable's
    `.data` was re-pointed into the arena), stale after invalidate(); `current()` hands the buffer to the update pass only while it is
    current.  Pure host logic: runs on the CPU."""
    import torch
    from learnablepoolingmethods_amd import ops
    arena = torch.randn(64 * 32 + 100)
    W = torch.empty(64, 32, requires_grad=True)
    with torch.no_grad():
        W.data = arena[:64 * 32].view(64, 32)                   # as ParameterArena binds a variable
    cc = ops.ComputeCopy(W, also=(arena,))
    assert cc.current(W) is None and cc.refreshes == 0          # never built: nothing for the update pass to write
    b = cc.tensor(W)
    assert cc.refreshes == 1 and torch.equal(b, W.detach().to(torch.bfloat16)) and cc.current(W) is b
    assert cc.tensor(W) is b and cc.refreshes == 1              # no writer in between: no rebuild
    with torch.no_grad():
        W.mul_(2.0)                                             # a write to the variable
    assert cc.current(W) is None
    assert torch.equal(cc.tensor(W), W.detach().to(torch.bfloat16)) and cc.refreshes == 2
    with torch.no_grad():
        arena.add_(1.0)                                         # a write to the arena (tools/determinism_check.py resets it this way)
    assert cc.current(W) is None
    assert torch.equal(cc.tensor(W), W.detach().to(torch.bfloat16)) and cc.refreshes == 3
    cc.invalidate()                                             # a write through raw pointers (the generic update): told explicitly
    assert cc.current(W) is None and torch.equal(cc.tensor(W), W.detach().to(torch.bfloat16)) and cc.refreshes == 4


def test_grad_join_hands_one_gradient_over_once_and_refuses_the_wrong_order():
    """ops.GradJoin (round 6): the aggregation's gradient of the frames travels to the encoder's node outside autograd.  It is taken exactly
    once; a gradient that arrives after the taker ran (the two readers ordered the other way round) or on top of an untaken one raises
    instead of being dropped; a join nobody accepted is inert."""
    from learnablepoolingmethods_amd import ops
    j = ops.GradJoin()
    assert not j.accepted and j.take() is None
    j = ops.GradJoin()
    j.accept()
    n0 = ops.GradJoin.puts
    g = object()
    j.put(g)
    assert ops.GradJoin.puts == n0 + 1 and j.take() is g and j.take() is None
    with pytest.raises(ops.LpmError, match="BEFORE"):
        j.put(object())
    j.accept()                      # a new forward pass re-arms it
    j.put(g)
    with pytest.raises(ops.LpmError, match="second gradient"):
        j.put(g)
