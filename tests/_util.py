"""Shared helpers for the parity tests."""
import numpy as np
import torch

from oracle import lpm_oracle as O

REL_TOL = 1e-3   # BASELINE.json north_star: "within 1e-3 relative fp32"


def rel_err(a, b, floor=1e-30):
    """max |a-b| / max(max |b|, floor) -- error relative to the tensor's scale (both converted to float64).
    ``floor`` is an absolute scale for tensors that are mathematically zero (e.g. the gradient of a BN gamma
    that a later L2-normalisation makes scale-invariant): there fp32 noise / 0 is not a meaningful ratio."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(floor))


def assert_close(a, b, tol=REL_TOL, what="", floor=1e-30):
    e = rel_err(a, b, floor)
    assert np.isfinite(e) and e <= tol, f"{what}: relative error {e:.3e} > {tol:.1e}"
    return e


def rel_l2(a, b, floor=1e-30):
    """||a-b||_F / max(||b||_F, floor).  Used for whole-model gradients, where a ReLU pre-activation within fp32
    rounding of 0 flips its mask between any two implementations (one column of one weight gradient changes
    by O(1): invisible in the Frobenius norm, fatal for a max-norm check, and not an error of either side)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(floor))


def cuda():
    assert torch.cuda.is_available(), "GPU test needs an MI355X"
    return torch.device("cuda:0")


def to64(p):
    return {k: v.detach().double().cpu() for k, v in p.items()}


def oracle_cfg(model="NetVladV1", **kw):
    return O.OracleConfig(model=model, **kw)


from oracle.test_weights import separate_relu_units  # noqa: E402,F401  (lives beside the oracle: smoke() uses it too)
