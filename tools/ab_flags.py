"""A/B a boolean FLAGS switch (or an ops.* module attribute with --ops) inside ONE process on ONE box: the cfg-2 training
step of bench.py, alternating the two settings (box-to-box variance is ~3 %, more than most single optimisations).
Usage: python tools/ab_flags.py fused_encoder_blocks [rounds] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from learnablepoolingmethods_amd import FLAGS, registry
from learnablepoolingmethods_amd.train import Trainer

name = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device("cuda:0")
trainer = Trainer(registry.get_model("NetVladV1"), vocab_size=bench.VOCAB, batch_size=bench.PER_GPU_BATCH, device=dev, seed=1234,
                  model_kwargs=bench.CFG, **bench.TRAIN)
raw, nf, labels = bench.synthetic_batch(bench.PER_GPU_BATCH, dev, seed=0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 4.0:
    trainer.step(raw, nf, labels)
torch.cuda.synchronize()
res = {True: [], False: []}
for r in range(rounds):
    for val in (True, False):
        setattr(FLAGS, name, val)
        for _ in range(5):
            trainer.step(raw, nf, labels)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            trainer.step(raw, nf, labels)
        torch.cuda.synchronize()
        res[val].append((time.perf_counter() - t) / steps * 1e3)
for val in (True, False):
    xs = sorted(res[val])
    print(f"{name}={val}: median {xs[len(xs) // 2]:.3f} ms/step  min {xs[0]:.3f}  all {[round(x, 3) for x in res[val]]}")
