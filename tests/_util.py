"""Shared helpers for the parity tests."""
import numpy as np
import torch

from oracle import lpm_oracle as O

REL_TOL = 1e-3   # BASELINE.json north_star: "within 1e-3 relative fp32"


def rel_err(a, b):
    """max |a-b| / max |b| -- error relative to the tensor's scale (both converted to float64)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_close(a, b, tol=REL_TOL, what=""):
    e = rel_err(a, b)
    assert np.isfinite(e) and e <= tol, f"{what}: relative error {e:.3e} > {tol:.1e}"
    return e


def cuda():
    assert torch.cuda.is_available(), "GPU test needs an MI355X"
    return torch.device("cuda:0")


def to64(p):
    return {k: v.detach().double().cpu() for k, v in p.items()}


def oracle_cfg(model="NetVladV1", **kw):
    return O.OracleConfig(model=model, **kw)
