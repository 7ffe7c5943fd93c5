"""Data-parallel consistency on the GPU box: N ranks (one per GPU over RCCL; LPM_SHARE_GPU=1: all on GPU 0 over gloo) take
steps on different shards; afterwards every rank must hold bit-identical parameters and Adam slots, and the summed-gradient
step must equal what one rank computes on the concatenated batch's tower gradients (checked through the loss trajectory).
This is a cross-rank CONSISTENCY diagnostic only (an error common to all ranks passes it): the comparison of two real-Trainer ranks with
oracle.train_step(num_towers=2) is tests/test_gpu_dp_trainer.py, which pytest collects.
Launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tests/diagnostics/dp_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from learnablepoolingmethods_amd import registry
from learnablepoolingmethods_amd.train import Trainer
from oracle import lpm_oracle as O

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
share = os.environ.get("LPM_SHARE_GPU") == "1"
dev = torch.device("cuda", 0 if share else int(os.environ["LOCAL_RANK"]))
torch.cuda.set_device(dev)
dist.init_process_group("gloo" if share else "nccl", **({} if share else {"device_id": dev}))
B, MF = 16, 40
tr = Trainer(registry.get_model("NetVladV1"), vocab_size=50, batch_size=B, base_learning_rate=1e-3, device=dev, seed=5,
             model_kwargs=dict(iterations=32, cluster_size=256, hidden_size=64))
x, nf, lab = O.make_synthetic_batch(B, MF, 1152, 50, seed=100 + rank, min_frames=10)
losses = [float(tr.step(x, nf, lab)["loss"]) for _ in range(3)]
torch.cuda.synchronize()
worst = 0.0
for name, t in (("param", tr.arena.param), ("m", tr.arena.m), ("v", tr.arena.v)):
    ref = t.clone()
    dist.broadcast(ref, src=0)
    d = (t - ref).abs().max()
    dist.all_reduce(d, op=dist.ReduceOp.MAX)
    worst = max(worst, float(d))
    if rank == 0:
        print(f"{name}: max |rank_i - rank_0| = {float(d):.3e}")
if rank == 0:
    print("losses rank 0:", [round(v, 5) for v in losses], "OK" if worst == 0.0 else "MISMATCH")
dist.destroy_process_group()
sys.exit(0 if worst == 0.0 else 1)
