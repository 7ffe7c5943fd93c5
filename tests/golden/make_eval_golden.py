"""Generates tests/golden/eval_golden.npz by running the REFERENCE's own eval_util (eval_util.py,
average_precision_calculator.py, mean_average_precision_calculator.py import here: numpy only) on seeded inputs.
Run in the build container (needs /root/reference); the .npz is data -- inputs and expected outputs -- and is the only
thing that travels.  Usage: python tests/golden/make_eval_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
import eval_util as E  # noqa: E402


def make_batch(rng, B, V, zero_label_row=False, zero_preds=False):
    """Distinct prediction values (the reference ranks tied predictions in the order of a seeded shuffle of its heap array;
    the GAP / AP goldens stay clear of ties, see eval_util.py in the build) except for the ``zero_preds`` case, which only
    pins Hit@1 and PERR (PERR counts predictions > 0 only)."""
    logits = rng.normal(size=(B, V)) * 2.0 - 3.0
    y = np.zeros((B, V), dtype=bool)
    for b in range(B):
        y[b, rng.choice(V, size=int(rng.integers(1, 6)), replace=False)] = True
        logits[b, y[b]] += rng.uniform(0.0, 5.0)                      # positives tend to score higher
    p = 1.0 / (1.0 + np.exp(-logits))                                 # float64: distinct values
    assert len(np.unique(p)) == p.size, "tied predictions"
    if zero_label_row:
        y[B // 2] = False
    if zero_preds:
        p[0, rng.choice(V, size=V - 2, replace=False)] = 0.0
    return p, y


def main():
    rng = np.random.default_rng(2018)
    out = {}
    pz, yz = make_batch(rng, 8, 40, zero_preds=True)
    out["zeros/predictions"], out["zeros/labels"] = pz, yz
    out["zeros/functions"] = np.array([E.calculate_hit_at_one(pz, yz), E.calculate_precision_at_equal_recall_rate(pz, yz)], dtype=np.float64)
    cases = [("small", 16, 100, 20, [16]), ("multi", 60, 400, 20, [24, 20, 16]), ("k5", 12, 64, 5, [12])]
    for name, B, V, k, splits in cases:
        p, y = make_batch(rng, B, V, zero_label_row=(name == "multi"))
        loss = rng.uniform(1.0, 9.0, size=B)
        m = E.EvaluationMetrics(V, k)
        per_batch, o = [], 0
        for s in splits:
            r = m.accumulate(p[o:o + s], y[o:o + s], loss[o:o + s])
            per_batch.append([r["hit_at_one"], r["perr"], r["loss"]])
            o += s
        g = m.get()
        out[f"{name}/predictions"], out[f"{name}/labels"], out[f"{name}/loss"] = p, y, loss
        out[f"{name}/splits"], out[f"{name}/top_k"] = np.array(splits), np.array(k)
        out[f"{name}/per_batch"] = np.array(per_batch, dtype=np.float64)
        out[f"{name}/epoch"] = np.array([g["avg_hit_at_one"], g["avg_perr"], g["avg_loss"], g["gap"]], dtype=np.float64)
        out[f"{name}/aps"] = np.array([float(a) for a in g["aps"]], dtype=np.float64)
        out[f"{name}/functions"] = np.array([E.calculate_hit_at_one(p, y), E.calculate_precision_at_equal_recall_rate(p, y),
                                             E.calculate_gap(p, y, top_k=k)], dtype=np.float64)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eval_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.endswith("epoch") or k.endswith("aps")})


if __name__ == "__main__":
    main()
