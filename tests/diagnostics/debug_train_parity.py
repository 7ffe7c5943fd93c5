"""Per-variable gradient / update parity report (GPU box): product Trainer vs oracle train_step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import lpm_oracle as O
from learnablepoolingmethods_amd import registry, ops
from learnablepoolingmethods_amd.train import Trainer
import learnablepoolingmethods_amd.train as T

dev = torch.device("cuda:0")
cfg = O.OracleConfig(model="NetVladV1", iterations=30, cluster_size=16, hidden_size=128, vocab_size=200, base_learning_rate=1e-3)
B = 4
x, nf, lab = O.make_synthetic_batch(B, 30, 1152, cfg.vocab_size, seed=0, min_frames=10)
p = {k: v.double() for k, v in O.init_params(cfg, 1152, seed=1000).items()}
tr = Trainer(registry.get_model("NetVladV1"), vocab_size=cfg.vocab_size, batch_size=B, base_learning_rate=1e-3, device=dev,
             model_kwargs=dict(iterations=30, cluster_size=16, hidden_size=128))
tr.build(x, nf, lab)
tr.store.load({"tower/" + k: v for k, v in p.items()})
# capture raw grads by monkeypatching the optimizer call
captured = {}
orig = ops.clip_adam_step
def spy(param, grad, m, v, offsets, nt, clip, lr, step, **kw):
    captured["grad"] = grad.clone()
    return orig(param, grad, m, v, offsets, nt, clip, lr, step, **kw)
ops.clip_adam_step = spy
out = tr.step(x, nf, lab)
pred, loss, gd, upd = O.loss_and_grads(p, x.double(), nf, lab, cfg)
p2, _, info = O.train_step(p, {"step": 0, "m": {}, "v": {}}, x.double(), nf, lab, cfg, 1)
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))
rows = []
for n in O.trainable_names(p, cfg):
    a0, a1 = tr.arena.segment("tower/" + n)
    g = captured["grad"][a0:a0 + p[n].numel()].reshape(p[n].shape)
    rows.append((rel(g, gd[n]), rel(tr.store.vars["tower/" + n], p2[n]), float(gd[n].abs().max()), n))
for r in sorted(rows, reverse=True):
    print("grad_rel %.2e  w_rel %.2e  |g|max %.2e  %s" % r)
