"""-m "not gpu": the oracle's 2-tower / 2-step run against the committed fixture (tests/golden/dp_2tower_golden.npz, provenance
in make_dp_golden.py), and the known-answer structure of that run: SUM (not mean) over towers, per-variable clip to norm 1,
the L2 regulariser inside every tower's gradient, the staircase learning rate on step * batch * towers examples."""
import os

import numpy as np
import torch

from oracle import lpm_oracle as O
from tests import dp_cases

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dp_2tower_golden.npz"))


def test_oracle_reproduces_the_two_tower_fixture():
    from tests.golden import make_dp_golden
    now = make_dp_golden.build()
    assert sorted(now) == sorted(G.files)
    for k in G.files:
        np.testing.assert_allclose(now[k], G[k], rtol=1e-9, atol=1e-13, err_msg=k)


def test_two_tower_step_structure():
    case = dp_cases.make_case("toy")
    ref = dp_cases.run_oracle(case)
    cfg, per = case["cfg"], case["per_tower"]
    x, nf, lab, p = case["x"].double(), case["nf"], case["lab"], case["params"]
    st0 = ref["steps"][0]
    # SUM of the per-tower gradients, each including moe_l2 * w (train.py:296-303,321; utils.py:207-211)
    g = [O.loss_and_grads(p, x[i * per:(i + 1) * per], nf[i * per:(i + 1) * per], lab[i * per:(i + 1) * per], cfg)[2] for i in range(2)]
    no_reg = O.OracleConfig(**{**cfg.__dict__, "moe_l2": 0.0})
    g0 = [O.loss_and_grads(p, x[i * per:(i + 1) * per], nf[i * per:(i + 1) * per], lab[i * per:(i + 1) * per], no_reg)[2] for i in range(2)]
    for n in st0["summed"]:
        torch.testing.assert_close(st0["summed"][n], g[0][n] + g[1][n], rtol=1e-12, atol=1e-15)
    for n in ("gates/weights", "experts/weights"):
        torch.testing.assert_close(st0["summed"][n], g0[0][n] + g0[1][n] + 2 * cfg.moe_l2 * p[n], rtol=1e-9, atol=1e-14)
        assert float((2 * cfg.moe_l2 * p[n]).norm() / st0["summed"][n].norm()) > 1e-2, "the penalty must be visible in this case"
    # per-variable clip (utils.py:181-188): norms never exceed 1, directions unchanged
    for n, c in st0["clipped"].items():
        s = st0["summed"][n]
        assert float(c.norm()) <= 1.0 + 1e-12
        torch.testing.assert_close(c, s * (1.0 / max(float(s.norm()), 1.0)), rtol=1e-12, atol=1e-15)
    # one Adam step from zero slots: m = 0.1 g, v = 0.001 g^2; and the inline step equals oracle.train_step on the whole batch
    assert st0["lr"] == cfg.base_learning_rate
    for n, c in st0["clipped"].items():
        torch.testing.assert_close(st0["m"][n], 0.1 * c, rtol=1e-12, atol=1e-18)
    p1, s1, info = O.train_step(p, {"step": 0, "m": {}, "v": {}}, x, nf, lab, cfg, 2)
    for n in p1:
        torch.testing.assert_close(st0["params"][n], p1[n], rtol=1e-12, atol=1e-15)
    torch.testing.assert_close(st0["loss"], info["loss"])
