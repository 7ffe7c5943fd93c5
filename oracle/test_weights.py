"""Test-weight preparation for the parity checks (TEST INFRASTRUCTURE, like everything under oracle/: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product never does).

``separate_relu_units`` walks the oracle's ReLU sites and nudges unit biases so that no pre-activation sits within fp32 rounding of
zero -- the one place where two correct implementations of the reference graph (transformer_utils.py:701-711,741-760) may
legitimately disagree."""
import torch

from oracle import lpm_oracle as O


def separate_relu_units(params, batches, cfg, margin=2e-4, is_training=True):
    """Seeded weights whose ReLU pre-activations all keep a distance from zero.

    d relu / dz jumps at z = 0: a unit whose pre-activation is within rounding of zero gets its mask from the last bit of
    whichever arithmetic computed it, and one such unit moves a weight gradient of the encoder by ~1/sqrt(tokens * units) of its
    norm (observed 1-2e-3) -- not an error of either side, but enough to blur a 1e-3 comparison.  Instead of loosening the
    tolerance, the test weights are prepared: walking the ReLU sites in forward order (the oracle reports them through
    ``RELU_TAPS``), the bias of every unit that has a pre-activation closer to zero than ``margin`` x (rms of the site) is moved
    by the smallest shift that puts zero into the middle of a gap of at least twice the margin between that unit's sorted
    pre-activations.  Shifts are ~1e-3 of the activation scale for a handful of units per site; the result is an ordinary
    weight set on which oracle and kernels must agree to the full tolerance.

    ``batches``: list of (model_input, num_frames, dropout_masks | None) -- every tower's batch for a data-parallel test
    (batch-norm statistics are per tower).  Returns (new params, report: site -> (units moved, smallest |z| / rms after))."""
    p = {k: v.clone() for k, v in params.items()}
    report, done = {}, []
    while True:
        taps_all = []
        for x, nf, dm in batches:
            O.RELU_TAPS = {}
            try:
                with torch.no_grad():
                    O.model_forward(p, x.to(next(iter(p.values())).dtype), nf, cfg, is_training, None, dm)
                taps_all.append(O.RELU_TAPS)
            finally:
                O.RELU_TAPS = None
        sites = [s for s in taps_all[0] if s not in done]
        if not sites:
            return p, report
        site = sites[0]                                   # dict order = forward order; later sites see the earlier fixes
        z = torch.cat([t[site].reshape(-1, t[site].shape[-1]) for t in taps_all], 0).double()      # [tokens, units]
        m = margin * float(z.pow(2).mean().sqrt())
        bad = (z.abs() < m).any(0).nonzero().flatten().tolist()
        for u in bad:
            v = torch.sort(z[:, u]).values
            ext = torch.cat([v[:1] - 4 * m, v, v[-1:] + 4 * m])
            gaps = ext[1:] - ext[:-1]
            mids = 0.5 * (ext[1:] + ext[:-1])
            ok = gaps >= 2.5 * m
            cand = mids[ok]
            mid = cand[cand.abs().argmin()]
            p[site][u] -= mid.to(p[site].dtype)           # z + delta with delta = -mid: zero now sits in the middle of that gap
            z[:, u] -= mid
        report[site] = (len(bad), float(z.abs().min() / z.pow(2).mean().sqrt()))
        done.append(site)
