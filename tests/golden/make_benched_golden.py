#!/usr/bin/env python3
"""Generates tests/golden/benched_<cfg>.npz: digests of ONE training step at the shapes bench.py times -- cfg-2 (NetVladV1, B = 80),
cfg-3 (NetVladV2, B = 80, the reference's rate-0.9 dropout through seeded keep masks), cfg-5 (gated NetVLAD K = 512 + MoE-4,
B = 128) -- from the reference-style initialisation (init_params, UN-scaled hidden1_weights).

PROVENANCE: produced by THIS REPO'S ORACLE (oracle/lpm_oracle.py, fp64 torch-CPU), not by the reference -- TensorFlow 1.x cannot run
here and the reference holds no vectors for this path: parity stays "unpinned" by the reference.  What the fixture
buys: every BASELINE single-GPU configuration is compared with the oracle AT THE SIZE THE BENCH TIMES, although
the fp64 oracle needs 1-3 minutes and 25-45 GB per configuration, which the GPU tier cannot spend: it was run once, in the build
container, and the GPU test (tests/test_gpu_benched_shapes.py) regenerates inputs and weights from the same seeds and holds the HIP
path to these digests.

Per tensor: its first 16 entries in flat order, 48 entries at a fixed stride, its sum and its L2 norm (dp_cases.digest extended).
Inputs and initial weights are digested too so that a drifting generator is noticed; the ReLU-margin bias vectors
(oracle/test_weights.separate_relu_units, computed on this very batch) are stored in full -- the GPU box does not run the oracle.

Run from the repo root, one configuration per process (memory):  python tests/golden/make_benched_golden.py cfg2|cfg3|cfg5
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lpm_oracle as O  # noqa: E402
from oracle.test_weights import separate_relu_units  # noqa: E402

CASES = {
    "cfg2": dict(model="NetVladV1", B=80, seed=11, sizes=dict(iterations=300, cluster_size=256, hidden_size=512), dropout=False),
    "cfg3": dict(model="NetVladV2", B=80, seed=12, sizes=dict(iterations=300, cluster_size=256, hidden_size=512), dropout=True),
    "cfg5": dict(model="NetVladV1", B=128, seed=13, sizes=dict(iterations=300, cluster_size=512, hidden_size=1024, encoder=False,
                                                                moe_num_mixtures=4), dropout=False),
}
INTERMEDIATES = ("input_bn", "vlad_video", "vlad_audio", "vlad", "activation")


def digest(t, k=16, m=48):
    """first k entries (flat order), m entries at stride numel // m, sum, L2 norm -- fp64."""
    t = t.detach().double().cpu().reshape(-1)
    n = t.numel()
    head = torch.cat([t[:k], torch.zeros(max(0, k - n), dtype=torch.float64)])
    idx = (torch.arange(m, dtype=torch.int64) * max(1, n // m)).clamp_max(n - 1)
    return torch.cat([head, t[idx], t.sum().reshape(1), t.norm().reshape(1)])


def make_inputs(name):
    """Seeded inputs, reference-style initial weights and dropout keep masks of a case (shared with the GPU test)."""
    c = CASES[name]
    cfg = O.OracleConfig(model=c["model"], vocab_size=3862, base_learning_rate=2e-4, **c["sizes"])
    x, nf, lab = O.make_synthetic_batch(c["B"], 300, 1152, cfg.vocab_size, seed=c["seed"])
    p = O.init_params(cfg, 1152, seed=1000 + c["seed"])
    masks = None
    if c["dropout"]:
        g = torch.Generator().manual_seed(500 + c["seed"])
        masks = {"video": (torch.rand(c["B"], 300, 1024, generator=g) >= 0.9).float(),
                 "audio": (torch.rand(c["B"], 300, 128, generator=g) >= 0.9).float()}
    return cfg, x, nf, lab, p, masks


def build(name):
    t0 = time.time()
    cfg, x, nf, lab, p32, masks = make_inputs(name)
    out = {"input_digest": digest(x).numpy(), "num_frames": nf.numpy(), "labels_digest": digest(lab.double()).numpy()}
    for n, v in p32.items():
        out["w0/" + n] = digest(v).numpy()
    p = {k: v.double() for k, v in p32.items()}
    dm = None if masks is None else {k: v.double() for k, v in masks.items()}
    x64 = x.double()
    p, report = separate_relu_units(p, [(x64, nf, dm)], cfg)
    for site, (moved, zmin) in report.items():
        out["relu_bias/" + site] = p[site].numpy()               # in full: the GPU test loads these instead of running the oracle
        out["relu_moved/" + site] = np.array([moved, zmin])
    print(f"[{name}] ReLU margins: {({k: v[0] for k, v in report.items()})} ({time.time() - t0:.0f} s)", flush=True)
    with torch.no_grad():
        _, inter = O.model_forward(p, x64, nf, cfg, True, None, dm, return_intermediates=True)
    for k in INTERMEDIATES:
        out["inter/" + k] = digest(inter[k]).numpy()
    del inter
    pred, loss, grads, _ = O.loss_and_grads(p, x64, nf, lab, cfg, dm)
    out["loss"] = np.array(float(loss))
    out["predictions"] = pred.detach().numpy().astype(np.float32)
    out["predictions_digest"] = digest(pred).numpy()
    for n in O.trainable_names(p, cfg):
        out["grad/" + n] = digest(grads[n]).numpy()
        out["grad_numel/" + n] = np.array(grads[n].numel())
    print(f"[{name}] loss {float(loss):.6f}, {len(grads)} gradients ({time.time() - t0:.0f} s)", flush=True)
    return out


if __name__ == "__main__":
    name = sys.argv[1]
    torch.set_num_threads(os.cpu_count())
    path = os.path.join(ROOT, "tests", "golden", f"benched_{name}.npz")
    data = build(name)
    np.savez_compressed(path, **data)
    print(f"wrote {path}: {len(data)} arrays, {os.path.getsize(path)} bytes")
