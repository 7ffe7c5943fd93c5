#!/usr/bin/env python3
"""Generates tests/golden/dp_2tower_golden.npz: the 2-tower / 2-step toy run of SURVEY 8(c).

PROVENANCE: produced by THIS REPO'S ORACLE (oracle/lpm_oracle.train_step with num_towers=2, fp64), not by the reference --
TensorFlow 1.x cannot run here and the reference holds no vectors for this path: parity stays "unpinned" by
the reference.  The fixture freezes what the restatement of train.py:266-336 / utils.py:170-213 (split over towers, per-tower
loss incl. the L2 regularisers, SUM, per-variable clip, TF-Adam, staircase learning rate) answers on a seeded toy problem:
(inputs, weights) -> (losses, predictions, per-variable summed + clipped gradients, post-step weights and Adam slots).
Inputs and weights are seeded (tests/dp_cases.py, case "toy"); the file holds digests of them (first 16 entries, sum, norm) so a
drifting generator is noticed, and digests or full values of every output.
Run from the repo root:  python tests/golden/make_dp_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import dp_cases  # noqa: E402


def build():
    case = dp_cases.make_case("toy")
    ref = dp_cases.run_oracle(case)
    out = {"input_digest": dp_cases.digest(case["x"]).numpy(), "num_frames": case["nf"].numpy(),
           "labels": case["lab"].numpy().astype(np.uint8)}
    for n, v in case["params"].items():
        out["w0/" + n] = dp_cases.digest(v).numpy()
    for s, st in enumerate(ref["steps"]):
        out[f"step{s}/loss"] = np.array(float(st["loss"]))
        out[f"step{s}/lr"] = np.array(float(st["lr"]))
        out[f"step{s}/predictions"] = st["predictions"].numpy()
        for n in st["summed"]:
            out[f"step{s}/summed/{n}"] = dp_cases.digest(st["summed"][n]).numpy()
            out[f"step{s}/clipped/{n}"] = dp_cases.digest(st["clipped"][n]).numpy()
    for n, v in ref["params"].items():
        out["w/" + n] = dp_cases.digest(v).numpy()
    for n in ref["m"]:
        out["adam_m/" + n] = dp_cases.digest(ref["m"][n]).numpy()
        out["adam_v/" + n] = dp_cases.digest(ref["v"][n]).numpy()
    for n, v in ref["moving_mean_of_towers"].items():
        out["tower_mean_stats/" + n] = v.numpy()
    return out


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "dp_2tower_golden.npz")
    data = build()
    np.savez_compressed(path, **data)
    print(f"wrote {path}: {len(data)} arrays, {os.path.getsize(path)} bytes")
