"""The shader clock INSIDE a training step.  A one-wave sampler kernel (lpm_clock_sampler) sits on a stream of its own and stores, every 10 us,
the constant 100 MHz counter and the shader-clock counter; markers (lpm_clock_marker) on the main stream stamp the step boundaries in the
same time base.  Steps alternate between two variants (default: the projection's input gradient on the own kernel / on the library's
GEMM), so both are seen in one process and one thermal state.  Prints, per variant, the mean shader clock in 20 equal slices of the
step and the step's length.
  python tools/clock_trace.py [cfg2|cfg3|cfg5] [variant: dx (default) | none]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from learnablepoolingmethods_amd import _capi, ops, registry
from learnablepoolingmethods_amd.ops import ptr
from learnablepoolingmethods_amd.train import Trainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
variant = sys.argv[2] if len(sys.argv) > 2 else "dx"
wl = bench.WORKLOADS[cfg]
bench.set_flags(wl)
dev = torch.device("cuda:0")
lib = _capi.load()
tr = Trainer(registry.get_model(wl.get("model", "NetVladV1")), vocab_size=bench.VOCAB, batch_size=wl["batch"], device=dev, seed=1234,
             model_kwargs=wl["model_kwargs"], **bench.TRAIN)
raw, nf, labels = bench.synthetic_batch(wl["batch"], dev, seed=0)


def setv(i):
    if variant == "dx":
        ops.PROJ_DX_STREAM_MIN_N = 512 if i % 2 == 0 else 1 << 30


for i in range(300):
    setv(i)
    tr.step(raw, nf, labels)
torch.cuda.synchronize()
NSTEP, PERIOD = 24, 1000                    # 10 us per sample
NS = 40000                                  # 400 ms of samples: covers the steps below
samples = torch.zeros(2 * NS, dtype=torch.int64, device=dev)
marks = torch.zeros(2 * (NSTEP + 1), dtype=torch.int64, device=dev)
side = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(side):
    lib.check(lib._lpm_clock_sampler(ptr(samples), NS, PERIOD, side.cuda_stream), "lpm_clock_sampler")
main = torch.cuda.current_stream().cuda_stream
lib.check(lib._lpm_clock_marker(ptr(marks), 0, main), "lpm_clock_marker")
for i in range(NSTEP):
    setv(i)
    tr.step(raw, nf, labels)
    lib.check(lib._lpm_clock_marker(ptr(marks), i + 1, main), "lpm_clock_marker")
torch.cuda.synchronize()
s = samples.cpu().numpy().reshape(-1, 2).astype(np.int64)
m = marks.cpu().numpy().reshape(-1, 2).astype(np.int64)
s = s[s[:, 0] > 0]
t = s[:, 0] / 100.0                         # us
f = np.diff(s[:, 1]) / np.diff(s[:, 0]) * 100.0      # MHz between neighbouring samples
tm = m[:, 0] / 100.0
print(f"{cfg}: {len(s)} samples, sampling gaps > 30 us: {(np.diff(t) > 30).sum()} (the sampler wave was descheduled / starved there)")
NB = 20
for v in (0, 1):
    prof, lens = np.zeros(NB), []
    cnt = 0
    for i in range(4, NSTEP):               # skip the first steps behind the launch of the sampler
        if i % 2 != v:
            continue
        a, b = tm[i], tm[i + 1]
        lens.append(b - a)
        edges = np.linspace(a, b, NB + 1)
        mid = (t[:-1] + t[1:]) / 2
        for k in range(NB):
            sel = (mid >= edges[k]) & (mid < edges[k + 1])
            prof[k] += f[sel].mean() if sel.any() else np.nan
        cnt += 1
    prof /= cnt
    name = {0: "own dx kernel", 1: "library dx GEMM"}[v] if variant == "dx" else f"steps {v} mod 2"
    print(f"  {name:16s}: step {np.mean(lens):7.1f} us (n={cnt}); mean shader clock {np.nanmean(prof):6.0f} MHz; per 5 % slice of the step:")
    print("     " + " ".join(f"{x:5.0f}" for x in prof))
