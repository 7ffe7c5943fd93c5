"""Is the step's time the sum of its kernels' times, or does it follow the power budget?  An idle gap (torch.cuda._sleep: one spinning thread,
next to no power) is put in front of every training step and the step time measured against the gap's length.  Slope 1: the gap simply
adds.  Slope < 1: the rest of the step got FASTER by what the idle chip saved -- the step runs against the power / thermal limit.
  python tools/idle_probe.py [cfg2|cfg3|cfg5]"""
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
tr = Trainer(registry.get_model(wl.get("model", "NetVladV1")), vocab_size=bench.VOCAB, batch_size=wl["batch"], device=dev, seed=1234,
             model_kwargs=wl["model_kwargs"], **bench.TRAIN)
raw, nf, labels = bench.synthetic_batch(wl["batch"], dev, seed=0)
for _ in range(300):
    tr.step(raw, nf, labels)
torch.cuda.synchronize()
# calibrate the spin kernel: cycles per microsecond
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 20_000_000 / (e0.elapsed_time(e1) * 1e3)
res = []
for rnd in range(3):
    for gap_us in (0, 200, 400, 800, 1600):
        n = 80
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            if gap_us:
                ev[i][0].record()
                torch.cuda._sleep(int(gap_us * cyc_per_us))      # (spins on the SHADER clock: its real length is measured, below)
                ev[i][1].record()
            tr.step(raw, nf, labels)
        torch.cuda.synchronize()
        step_ms = (time.perf_counter() - t0) / n * 1e3
        gap_ms = sum(a.elapsed_time(b) for a, b in ev) / n if gap_us else 0.0
        res.append((gap_us, step_ms, gap_ms))
base = sorted(t for g, t, _ in res if g == 0)[1]
print(f"{cfg}: spin kernel {cyc_per_us:.0f} cycles/us when calibrated alone; step without a gap {base:.3f} ms")
for gap_us in (200, 400, 800, 1600):
    rows = sorted((t, gm) for g, t, gm in res if g == gap_us)
    med, gap = rows[1]
    rest = med - gap
    print(f"  idle gap asked {gap_us:5d} us, measured {gap * 1e3:6.0f} us per step -> step {med:.3f} ms; the rest of the step: {rest:.3f} ms "
          f"({(rest - base) * 1e3:+6.0f} us = {(rest - base) / base * 100:+.1f} %)")
