"""One process, one thermal state: the projection's input gradient alternates between the own weight-stream kernel (~135 us at cfg-2) and the
library's fp32 GEMM (~205 us) in blocks of `blk` steps; every step is bracketed by events.  If step time were the sum of its kernels'
times, the two classes of steps would differ by the two kernels' difference (~70 us).
  python tools/dx_toggle_probe.py [cfg2|cfg3] [block length, default 1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from learnablepoolingmethods_amd import ops, registry
from learnablepoolingmethods_amd.train import Trainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
blk = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wl = bench.WORKLOADS[cfg]
bench.set_flags(wl)
dev = torch.device("cuda:0")
tr = Trainer(registry.get_model(wl.get("model", "NetVladV1")), vocab_size=bench.VOCAB, batch_size=wl["batch"], device=dev, seed=1234,
             model_kwargs=wl["model_kwargs"], **bench.TRAIN)
raw, nf, labels = bench.synthetic_batch(wl["batch"], dev, seed=0)
for i in range(400):
    ops.PROJ_DX_STREAM_MIN_N = 512 if (i // blk) % 2 == 0 else 1 << 30
    tr.step(raw, nf, labels)
torch.cuda.synchronize()
N = 400
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record()
for i in range(N):
    ops.PROJ_DX_STREAM_MIN_N = 512 if (i // blk) % 2 == 0 else 1 << 30
    tr.step(raw, nf, labels)
    ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
own = sorted(t[i] for i in range(N) if (i // blk) % 2 == 0 and i % blk == blk - 1 or blk == 1 and (i // blk) % 2 == 0)
lib = sorted(t[i] for i in range(N) if (i // blk) % 2 == 1 and (i % blk == blk - 1 or blk == 1))
med = lambda v: v[len(v) // 2]
print(f"{cfg}, blocks of {blk}: steps with the own dx kernel median {med(own):.3f} ms (n={len(own)}), with the library's {med(lib):.3f} ms (n={len(lib)}): "
      f"difference {1e3 * (med(lib) - med(own)):+.0f} us (means {sum(own) / len(own):.3f} / {sum(lib) / len(lib):.3f})")
