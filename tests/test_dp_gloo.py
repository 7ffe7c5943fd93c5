"""-m "not gpu": the data-parallel path with world_size 2 on the gloo backend (CPU).
Each rank is a tower (train.py:266-284): it computes gradients on its shard; the arena all-reduce must equal
utils.combine_gradients (SUM, not mean) of the oracle's per-tower gradients, including the early-launched
hidden1_weights bucket, and ranks must start from rank 0's weights."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import lpm_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_loss(vars_, x, y):
    """A small stand-in model built only from host-side torch ops (the HIP ops need a GPU): two dense layers
    named like the real variables so the bucket logic is exercised."""
    h = torch.tanh(x @ vars_["tower/hidden1_weights"] + vars_["tower/hidden1_biases"])
    p = torch.sigmoid(h @ vars_["tower/gates/weights"])
    return ((p - y) ** 2).sum(dim=1).mean()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from learnablepoolingmethods_amd import train
    from learnablepoolingmethods_amd import variables as vs
    torch.manual_seed(100 + rank)                      # ranks start DIFFERENT: the arena broadcast must fix that
    store = vs.VariableStore(device="cpu", seed=rank)
    with vs.use_store(store), vs.variable_scope("tower"):
        vs.get_variable("hidden1_biases", [6], vs.random_normal_initializer(0.1))
        vs.get_variable("hidden1_weights", [10, 6], vs.random_normal_initializer(0.3))
        with vs.variable_scope("gates"):
            vs.get_variable("weights", [6, 4], vs.random_normal_initializer(0.3))
    arena = train.ParameterArena(store, first=["tower/hidden1_weights"])
    a0, a1 = arena.segment("tower/hidden1_weights")
    sync = train.GradientSynchronizer(arena.grad, [(a0, a1), (a1, arena.total)])
    assert sync.active
    launched = []

    def early(_param):
        launched.append(1)
        sync.launch(0)
    arena.views["tower/hidden1_weights"].register_post_accumulate_grad_hook(early)
    dist.broadcast(arena.param, src=0)
    g = torch.Generator().manual_seed(7)               # same global batch on every rank, each takes its shard
    X, Y = torch.randn(8, 10, generator=g), torch.rand(8, 4, generator=g)
    per = 8 // world
    sl = slice(rank * per, (rank + 1) * per)
    arena.zero_grad()
    _toy_loss(arena.views, X[sl], Y[sl]).backward()
    sync.finish()
    # numpy, not tensors: torch tensors cross a mp.Queue as shared-memory handles that die with the worker
    out = {n: arena.views[n].grad.detach().numpy().copy() for n in arena.names}
    params = {n: arena.views[n].detach().numpy().copy() for n in arena.names}
    q.put((rank, out, params, len(launched)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gradient_sum_allreduce_matches_tower_combine():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, grads, params, launched = q.get(timeout=120)
        res[r] = ({n: torch.from_numpy(v) for n, v in grads.items()}, {n: torch.from_numpy(v) for n, v in params.items()},
                  launched)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # ranks share rank 0's weights
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], res[1][1][n])
    # oracle: per-tower gradients on the same shards, SUMMED (utils.py:207-211)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 10, generator=g), torch.rand(8, 4, generator=g)
    tower = []
    for r in range(world):
        leaf = {n: v.clone().requires_grad_(True) for n, v in res[0][1].items()}
        _toy_loss(leaf, X[r * 4:(r + 1) * 4], Y[r * 4:(r + 1) * 4]).backward()
        tower.append({n: v.grad for n, v in leaf.items()})
    ref = O.combine_gradients(tower)
    for r in range(world):
        assert res[r][2] == 1, "the early bucket hook must fire exactly once per backward"
        for n in ref:
            assert torch.allclose(res[r][0][n], ref[n], rtol=1e-6, atol=1e-7), n
    # SUM, not mean
    mean = {n: ref[n] / world for n in ref}
    assert not torch.allclose(res[0][0]["tower/hidden1_weights"], mean["tower/hidden1_weights"])
