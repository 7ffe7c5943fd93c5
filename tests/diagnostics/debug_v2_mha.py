"""Capture the attention-core operands inside the small NetVladV2 model and compare both arithmetics against fp64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import lpm_oracle as O
from tests.test_gpu_models import _well_conditioned
from learnablepoolingmethods_amd import ops, registry
from learnablepoolingmethods_amd.train import Trainer

dev = torch.device("cuda:0")
cfg = O.OracleConfig(model="NetVladV2", iterations=12, cluster_size=16, hidden_size=32, vocab_size=40,
                     base_learning_rate=1e-3, v2_dropout_rate=0.0)
B, MF, feat = 4, 16, 1152
x, nf, lab = O.make_synthetic_batch(B, MF, feat, cfg.vocab_size, seed=7, min_frames=max(2, MF // 3))
p = _well_conditioned({k: v.double() for k, v in O.init_params(cfg, feat, seed=1007).items()})
cap = []
orig = ops.mha_core_bn


def spy(q, k, v, h, gamma, beta, mm, mv, is_training=True):
    cap.append((q.detach().clone(), k.detach().clone(), v.detach().clone(), h, gamma.detach().clone(), beta.detach().clone(),
                mm.detach().clone(), mv.detach().clone()))
    return orig(q, k, v, h, gamma, beta, mm, mv, is_training)


ops.mha_core_bn = spy
tr = Trainer(registry.get_model("NetVladV2"), vocab_size=cfg.vocab_size, batch_size=B, base_learning_rate=cfg.base_learning_rate,
             learning_rate_decay=cfg.learning_rate_decay, learning_rate_decay_examples=cfg.learning_rate_decay_examples,
             device=dev, model_kwargs=dict(iterations=cfg.iterations, cluster_size=cfg.cluster_size, hidden_size=cfg.hidden_size,
                                           dropout_rate=0.0))
tr.build(x, nf, lab)
tr.store.load({"tower/" + k: v for k, v in p.items()})
cap.clear()
tr.step(x, nf, lab)
ops.mha_core_bn = orig
for (q, k, v, h, gamma, beta, mm, mv) in cap:
    Bq, L, F = q.shape
    print("operands", tuple(q.shape), "heads", h, "|q|max %.2f |k|max %.2f |v|max %.2f" % (q.abs().max(), k.abs().max(), v.abs().max()))
    do = torch.randn_like(q)
    pd = {"bn/gamma": gamma.double().cpu().requires_grad_(True), "bn/beta": beta.double().cpu().requires_grad_(True),
          "bn/moving_mean": mm.double().cpu(), "bn/moving_variance": mv.double().cpu()}
    qd, kd, vd = (t.double().cpu().requires_grad_(True) for t in (q, k, v))
    upd = {}
    ref = O._combine_heads(O.attention_core(O._split_heads(qd, h), O._split_heads(kd, h), O._split_heads(vd, h), 1.0,
                                            lambda lg: O.batch_norm(lg, pd, "bn", True, upd)))
    ref.backward(do.double().cpu())
    s = torch.einsum("bqhd,bkhd->bhqk", qd.view(Bq, L, h, -1), kd.view(Bq, L, h, -1))
    print("  logits |s|max %.1f std %.2f" % (s.abs().max(), s.std()))
    for prec in ("f32", "bf16x3"):
        ops.MHA_PRECISION = prec
        qg, kg, vg, gg, bg = (t.clone().requires_grad_(True) for t in (q, k, v, gamma, beta))
        out = ops.mha_core_bn(qg, kg, vg, h, gg, bg, mm.clone(), mv.clone(), is_training=True)
        out.backward(do)

        def err(a, b):
            b = b.to(a.device).float()
            return float((a - b).abs().max() / b.abs().max())
        print("  %-7s fwd %.1e dq %.1e dk %.1e dv %.1e dgamma %.1e dbeta %.1e" % (prec, err(out, ref), err(qg.grad, qd.grad), err(kg.grad, kd.grad),
                                                                          err(vg.grad, vd.grad), err(gg.grad, pd["bn/gamma"].grad), err(bg.grad, pd["bn/beta"].grad)))
