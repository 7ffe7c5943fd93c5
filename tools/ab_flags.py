"""A/B a boolean FLAGS switch (or, as ops.NAME, a module attribute of ops.py) inside ONE process on ONE box: the training step of a
bench.py workload, alternating the two settings (box-to-box variance is ~3 %, more than most single optimisations).
Usage: python tools/ab_flags.py fused_encoder_blocks|ops.PROJ_STREAM [rounds] [steps] [cfg2|cfg5]
       python tools/ab_flags.py ops.PROJ_DX_STREAM_MIN_N=512,1024 ...      (two explicit values instead of True / False)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from learnablepoolingmethods_amd import FLAGS, ops, registry
from learnablepoolingmethods_amd.train import Trainer

name = sys.argv[1]
VALS = (True, False)
if "=" in name:
    name, vv = name.split("=")
    VALS = tuple(int(v) for v in vv.split(","))
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
wl = bench.WORKLOADS[sys.argv[4] if len(sys.argv) > 4 else "cfg2"]
bench.set_flags(wl)
dev = torch.device("cuda:0")
trainer = Trainer(registry.get_model(wl.get("model", "NetVladV1")), vocab_size=bench.VOCAB, batch_size=wl["batch"], device=dev, seed=1234,
                  model_kwargs=wl["model_kwargs"], **bench.TRAIN)
raw, nf, labels = bench.synthetic_batch(wl["batch"], dev, seed=0)


def setattr(_flags, key, val):          # ops.NAME: a module attribute of ops.py instead of a flag
    if key.startswith("ops."):
        ops.__dict__[key[4:]] = val
    else:
        _flags.__setattr__(key, val)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 4.0:
    trainer.step(raw, nf, labels)
torch.cuda.synchronize()
res = {v: [] for v in VALS}
for r in range(rounds):
    for val in VALS:
        setattr(FLAGS, name, val)
        for _ in range(5):
            trainer.step(raw, nf, labels)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            trainer.step(raw, nf, labels)
        torch.cuda.synchronize()
        res[val].append((time.perf_counter() - t) / steps * 1e3)
for val in VALS:
    xs = sorted(res[val])
    print(f"{name}={val}: median {xs[len(xs) // 2]:.3f} ms/step  min {xs[0]:.3f}  all {[round(x, 3) for x in res[val]]}")
