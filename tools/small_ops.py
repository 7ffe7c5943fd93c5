"""Which Python lines issue the small at::native launches of a training step?  One step of a bench.py workload under torch.profiler with
stacks; prints, for the aten ops that launch the fill / copy / add / mul / sum / cat kernels, their count per step and the innermost
repo frames.  Usage: python tools/small_ops.py [cfg2|cfg3|cfg5]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
import bench
from learnablepoolingmethods_amd import registry
from learnablepoolingmethods_amd.train import Trainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
wl = bench.WORKLOADS[cfg]
bench.set_flags(wl)
dev = torch.device("cuda:0")
tr = Trainer(registry.get_model(wl.get("model", "NetVladV1")), vocab_size=bench.VOCAB, batch_size=wl["batch"], device=dev, seed=1234,
             model_kwargs=wl["model_kwargs"], **bench.TRAIN)
raw, nf, labels = bench.synthetic_batch(wl["batch"], dev, seed=0)
for _ in range(5):
    tr.step(raw, nf, labels)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        tr.step(raw, nf, labels)
    torch.cuda.synchronize()
WANT = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::sum", "aten::cat", "aten::clone",
        "aten::contiguous", "aten::div", "aten::sub", "aten::neg", "aten::sigmoid", "aten::_foreach_copy_", "aten::_foreach_add_", "aten::zeros",
        "aten::full", "aten::to", "aten::_to_copy", "aten::stack", "aten::index", "aten::select", "aten::mean", "aten::sqrt", "aten::rsqrt")
sites = collections.Counter()
for ev in prof.events():
    if ev.name in WANT and ev.device_time_total > 0 or (ev.name in WANT and any(k.device_time > 0 for k in getattr(ev, "kernels", []))):
        frames = [f for f in (ev.stack or []) if "/root/repo" in f or "learnablepoolingmethods_amd" in f or "bench.py" in f]
        where = " <- ".join(f.split("/")[-1] for f in frames[:3]) or "(no repo frame)"
        sites[(ev.name, where)] += 1
print(f"{cfg}: aten ops with device kernels, per step (x{N} steps profiled)")
for (name, where), c in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"{c / N:6.1f}  {name:22s} {where}")
