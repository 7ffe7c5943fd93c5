"""How long does the HOST take to enqueue one training step, against how long the GPU takes to run it?  If the host is ahead, the GPU never
waits for a launch; if a step's enqueue takes as long as its execution, something in the step blocks the host.
  python tools/host_enqueue_time.py [cfg2|cfg3|cfg5]          (prints the median host time per step call and the GPU time per step)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from learnablepoolingmethods_amd import registry
from learnablepoolingmethods_amd.train import Trainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
wl = bench.WORKLOADS[cfg]
bench.set_flags(wl)
dev = torch.device("cuda:0")
trainer = Trainer(registry.get_model(wl.get("model", "NetVladV1")), vocab_size=bench.VOCAB, batch_size=wl["batch"], device=dev, seed=1234,
                  model_kwargs=wl["model_kwargs"], **bench.TRAIN)
raw, nf, labels = bench.synthetic_batch(wl["batch"], dev, seed=0)
for _ in range(60):
    trainer.step(raw, nf, labels)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
N = 40
for _ in range(N):
    a = time.perf_counter()
    trainer.step(raw, nf, labels)
    host.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
first = host[:4]
host.sort()
print(f"{cfg}: host time per step call median {host[N // 2] * 1e3:.2f} ms (min {host[0] * 1e3:.2f}, max {host[-1] * 1e3:.2f}; first four "
      f"{[round(h * 1e3, 2) for h in first]}); all {N} calls returned after {t_enq * 1e3:.1f} ms, the GPU finished after {t_all * 1e3:.1f} ms "
      f"= {t_all / N * 1e3:.2f} ms per step")

# ... and with the queue EMPTY at every call (synchronise first): the call's own cost, nothing blocking it
alone = []
for _ in range(20):
    torch.cuda.synchronize()
    a = time.perf_counter()
    trainer.step(raw, nf, labels)
    alone.append(time.perf_counter() - a)
torch.cuda.synchronize()
alone.sort()
print(f"{cfg}: host time of a step call into an empty queue: median {alone[10] * 1e3:.2f} ms (min {alone[0] * 1e3:.2f}, max {alone[-1] * 1e3:.2f})")
