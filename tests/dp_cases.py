"""Data-parallel parity cases: a real NetVladV1 trainer on N towers against ``oracle.train_step(num_towers=N)``.

Shared by tests/golden/make_dp_golden.py (writes the 2-tower / 2-step fixture of SURVEY 8(c)), tests/test_dp_golden.py
(``-m "not gpu"``: the live oracle against the committed fixture) and tests/test_gpu_dp_trainer.py (``-m gpu``: two ranks of the
product sharing GPU 0 over gloo against the live oracle and against the fixture).  Reference semantics: train.py:266-336
(split, per-tower loss incl. the regularisers, SUM, per-variable clip, Adam), utils.py:170-213.
"""
import torch

from oracle import lpm_oracle as O
from tests._util import separate_relu_units

# moe_l2 is raised from the reference's 1e-8 so that the penalty's gradient (moe_l2 * w, a13) is a visible part of the MoE
# weight gradients: dropping it anywhere on the data-parallel path fails these cases.
CASES = {
    # the fixture case: small enough that its weights need not be stored (seeded) and the CPU suite can re-run it
    "toy": dict(cfg=dict(iterations=12, cluster_size=16, hidden_size=32, vocab_size=40, base_learning_rate=1e-3, moe_l2=1e-2),
                per_tower=4, max_frames=16, towers=2, steps=2, data_seed=31, weight_seed=1031),
    # the production code path: both encoders as block Functions into the shared descriptor buffer (>= 1024 tokens per
    # stream), closed-form input_bn gradients, tile forms of K1 / K2 / K3, head + encoder buckets all-reduced from hooks
    "blocks": dict(cfg=dict(iterations=32, cluster_size=256, hidden_size=64, vocab_size=50, base_learning_rate=1e-3, moe_l2=1e-2),
                   per_tower=16, max_frames=40, towers=2, steps=2, data_seed=33, weight_seed=1033),
}


def make_case(name):
    """-> dict(cfg, x, nf, lab, params): the global batch (towers x per_tower clips) and fp64 weights with hidden1_weights in the
    well-conditioned regime (tests/test_gpu_models._well_conditioned) and every ReLU pre-activation of every tower kept away
    from zero (oracle/test_weights.separate_relu_units)."""
    c = CASES[name]
    cfg = O.OracleConfig(model="NetVladV1", **c["cfg"])
    B = c["per_tower"] * c["towers"]
    x, nf, lab = O.make_synthetic_batch(B, c["max_frames"], 1152, cfg.vocab_size, seed=c["data_seed"],
                                        min_frames=max(2, c["max_frames"] // 3))
    p = {k: v.double() for k, v in O.init_params(cfg, 1152, seed=c["weight_seed"]).items()}
    p["hidden1_weights"] = p["hidden1_weights"] * 0.02
    per = c["per_tower"]
    batches = [(x[i * per:(i + 1) * per].double(), nf[i * per:(i + 1) * per], None) for i in range(c["towers"])]
    p, report = separate_relu_units(p, batches, cfg)
    return dict(name=name, cfg=cfg, x=x, nf=nf, lab=lab, params=p, relu_report=report, **{k: c[k] for k in ("per_tower", "towers", "steps")})


def run_oracle(case):
    """The reference's multi-tower step (train.py:266-336 as oracle.train_step restates it), ``steps`` times, with every
    tower's gradients computed once.  -> dict with, per step, loss / predictions / the SUMMED raw gradients / the clipped
    gradients / weights and Adam slots after the step, and the per-tower batch-norm statistics at the end."""
    cfg, x, nf, lab = case["cfg"], case["x"].double(), case["nf"], case["lab"]
    T, per = case["towers"], case["per_tower"]
    p, st = case["params"], {"step": 0, "m": {}, "v": {}}
    out = {"steps": []}
    tower_stats = [dict() for _ in range(T)]           # what each tower alone would hold as moving statistics
    for n in p:
        if n.endswith("moving_mean") or n.endswith("moving_variance"):
            for ts in tower_stats:
                ts[n] = p[n].clone()
    for s in range(case["steps"]):
        raw, losses, preds = [], [], []
        new_p = dict(p)
        for i in range(T):                                                      # train.py:273-284, one tower per shard
            sl = slice(i * per, (i + 1) * per)
            pred, loss, gd, upd = O.loss_and_grads(p, x[sl], nf[sl], lab[sl], cfg)
            raw.append(gd); losses.append(loss); preds.append(pred)
            for n, val in upd.items():
                tower_stats[i][n] = tower_stats[i][n] * O.BN_DECAY + val * (1 - O.BN_DECAY)
                new_p[n] = new_p[n] * O.BN_DECAY + val * (1 - O.BN_DECAY)       # shared variables, tower order (:309-316)
        summed = O.combine_gradients(raw)                                       # :330
        clipped = O.clip_gradient_norms(summed, cfg.clip_gradient_norm)         # :332-334
        lr = O.learning_rate(cfg, st["step"], per, T)                           # :244-249
        t = st["step"] + 1
        m, v = {}, {}
        for n, g in clipped.items():
            new_p[n], m[n], v[n] = O.adam_tf_update(p[n], g, st["m"].get(n, torch.zeros_like(p[n])), st["v"].get(n, torch.zeros_like(p[n])), lr, t)
        p, st = new_p, {"step": t, "m": m, "v": v}
        out["steps"].append(dict(loss=torch.stack(losses).mean(), predictions=torch.cat(preds, 0), summed=summed, clipped=clipped, lr=lr,
                                 params=p, m=m, v=v))
    out["params"], out["m"], out["v"] = p, st["m"], st["v"]
    # what the product reports at a checkpoint: the mean over towers of their own moving averages
    out["moving_mean_of_towers"] = {n: torch.stack([ts[n] for ts in tower_stats]).mean(0) for n in tower_stats[0]}
    out["tower_stats"] = tower_stats
    return out


def digest(t, k=16):
    """A small, order-sensitive summary of a tensor for the fixture: its first k entries in flat order, its sum and its L2 norm."""
    t = t.detach().double().reshape(-1)
    return torch.cat([t[:k], torch.zeros(max(0, k - t.numel()), dtype=torch.float64), t.sum().reshape(1), t.norm().reshape(1)])
