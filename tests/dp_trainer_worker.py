"""One rank of tests/test_gpu_dp_trainer.py (not a test module).  Started as a fresh child process BEFORE it touches the GPU:

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dp_trainer_worker.py <dir> <side_stream 0|1> [<factored 0|1>]

Default (LPM_DP_BACKEND unset or "gloo"): all ranks share GPU 0 and exchange gradients over gloo (the boxes of this pool have one
GPU; the code path -- arena buckets gathered and all-reduced from autograd hooks, early hidden1_weights bucket, per-variable clip
+ Adam on the summed arena -- is the one RCCL drives with one rank per GPU).  LPM_DP_BACKEND=nccl: one rank per GPU (cuda:LOCAL_RANK)
over RCCL, the measured configuration; needs torch.cuda.device_count() >= WORLD_SIZE.  Reads <dir>/inputs.pt, writes <dir>/rank<r>.pt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_dir, side = sys.argv[1], sys.argv[2] == "1"
    route = sys.argv[3] if len(sys.argv) >= 4 else "1"       # "1": factored, "0": generic (bucket all-reduce), "sharded": route C
    factored = route == "1"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("LPM_DP_BACKEND", "gloo")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)) if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":                               # nccl == RCCL on ROCm; one rank per GPU
        import datetime
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=300))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from learnablepoolingmethods_amd import FLAGS, registry
    from learnablepoolingmethods_amd.train import Trainer
    inp = torch.load(os.path.join(out_dir, "inputs.pt"))
    c = inp["cfg"]
    FLAGS.moe_l2 = c["moe_l2"]
    FLAGS.audio_side_stream = side
    FLAGS.hidden1_factored_update = factored
    if route == "sharded":
        FLAGS.hidden1_sharded_min_towers = 2            # (the default starts above hidden1_factored_max_towers = 4 towers)
    per = inp["per_tower"]
    sl = slice(rank * per, (rank + 1) * per)
    x, nf, lab = inp["x"][sl], inp["nf"][sl], inp["lab"][sl]
    tr = Trainer(registry.get_model("NetVladV1"), vocab_size=c["vocab_size"], batch_size=per, base_learning_rate=c["base_learning_rate"],
                 learning_rate_decay=c["learning_rate_decay"], learning_rate_decay_examples=c["learning_rate_decay_examples"],
                 device=dev, seed=100 + rank,           # ranks start DIFFERENT: build() must broadcast rank 0's weights
                 model_kwargs=dict(iterations=c["iterations"], cluster_size=c["cluster_size"], hidden_size=c["hidden_size"]))
    tr.build(x, nf, lab)
    assert tr.sync.active and tr.num_towers == world
    assert (tr.sharded is not None) == (route == "sharded"), "the route of hidden1_weights' gradient is not the requested one"
    if tr.sharded is not None:
        tr.sharded.keep_summed = True
    res = {"early_buckets": tr.bucket_gather.early if tr.bucket_gather is not None else [], "steps": []}
    if rank == 0:                                       # only rank 0 gets the oracle's weights: the others must receive them
        tr.store.load({"tower/" + k: v for k, v in inp["params"].items()})
    dist.broadcast(tr.arena.param, src=0)
    for n, v in tr.store.vars.items():
        if not tr.store.trainable[n]:
            dist.broadcast(v, src=0)
    tr.invalidate_compute_copies()                      # (a collective into the arena does not bump torch's version counters: ADVICE r5)
    names = list(tr.arena.names)
    for s in range(inp["steps"]):
        tr.wait_pending()                               # (read below past get_variable: the parameter all-gather of the previous step)
        before = {n: tr.store.vars[n].detach().double().cpu() for n in names}
        o = tr.step(x, nf, lab)
        torch.cuda.synchronize()
        grads = {}
        used = tr.factored is not None and tr.factored.pending
        if used:
            # hidden1_weights' gradient was never written: what the optimiser consumed is the product of the all-gathered factors
            assert tr.factored.R == world * per, (tr.factored.R, world, per)
        for n in names:
            if tr.sharded is not None and n == names[0]:
                continue                                # (Trainer.gradient raises there: the sum exists only as the owners' shards)
            grads[n] = tr.gradient(n).double().cpu()
        if tr.sharded is not None:
            # the summed gradient of hidden1_weights exists shard by shard only: put the ranks' shards together (and their Adam moments)
            sh = tr.sharded
            parts = [torch.empty_like(sh.summed_shard) for _ in range(world)]
            dist.all_gather(parts, sh.summed_shard)
            t = tr.arena.views[names[0]]
            grads[names[0]] = torch.cat(parts)[:t.numel()].view(t.shape).double().cpu()
            sh.gather_moments()
        slots = {}
        for n in names:
            a0, _ = tr.arena.segment(n)
            k, shape = tr.arena.views[n].numel(), tr.arena.views[n].shape
            slots[n] = (tr.arena.m[a0:a0 + k].reshape(shape).double().cpu(), tr.arena.v[a0:a0 + k].reshape(shape).double().cpu())
        res["steps"].append(dict(factored=used, sharded=tr.sharded is not None, loss=o["loss"].double().cpu(), predictions=o["predictions"].double().cpu(), lr=o["learning_rate"],
                                 summed=grads, before=before, adam=slots,
                                 gathered=sorted(tr.bucket_gather.gathered) if tr.bucket_gather is not None else []))
    res["local_stats"] = {n: v.detach().double().cpu() for n, v in tr.store.vars.items() if not tr.store.trainable[n]}
    sd = tr.state_dict(sync=True)                       # sync=True is a collective (every rank calls it): moving statistics averaged over ranks
    res["state"] = {n: (v.double() if torch.is_tensor(v) else v) for n, v in sd.items()}
    torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
