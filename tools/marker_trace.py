"""Where does a step's time go WITHOUT a profiler?  Every launching call of the library (lpm_* with a stream argument) is followed by a
one-thread marker kernel on the same stream (lpm_clock_marker: the constant 100 MHz counter), also torch.mm / addmm / bmm; steps alternate
between two variants (default: the projection's input gradient on the own kernel / on the library's GEMM).  Printed: the intervals between
consecutive markers of the main stream, averaged per variant, where the variants differ most -- i.e. which launches absorb (or give)
time.  The markers cost ~2 us each and serialise nothing that was not serial before; both variants carry the same ones.
  python tools/marker_trace.py [cfg2|cfg3|cfg5] [control]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
import bench
from learnablepoolingmethods_amd import _capi, ops, registry
from learnablepoolingmethods_amd.ops import ptr
from learnablepoolingmethods_amd.train import Trainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
wl = bench.WORKLOADS[cfg]
bench.set_flags(wl)
dev = torch.device("cuda:0")
lib = _capi.load()
tr = Trainer(registry.get_model(wl.get("model", "NetVladV1")), vocab_size=bench.VOCAB, batch_size=wl["batch"], device=dev, seed=1234,
             model_kwargs=wl["model_kwargs"], **bench.TRAIN)
raw, nf, labels = bench.synthetic_batch(wl["batch"], dev, seed=0)


CONTROL = len(sys.argv) > 2 and sys.argv[2] == "control"       # no toggle: even and odd steps are the same code (the noise floor of the method)


def setv(i):
    ops.PROJ_DX_STREAM_MIN_N = 512 if (i % 2 == 0 or CONTROL) else 1 << 30


for i in range(200):
    setv(i)
    tr.step(raw, nf, labels)
torch.cuda.synchronize()

MAXM = 1 << 16
marks = torch.zeros(2 * MAXM, dtype=torch.int64, device=dev)
log = []            # (slot, name, stream)
state = {"on": False, "n": 0}
marker = lib._lpm_clock_marker
main_stream = torch.cuda.current_stream().cuda_stream


def mark(name, stream):
    if not state["on"] or state["n"] >= MAXM:
        return
    marker(ptr(marks), state["n"], stream)
    log.append((state["n"], name, stream or 0))
    state["n"] += 1


for name, (res, args) in _capi.SIGNATURES.items():
    if name in ("lpm_clock_marker", "lpm_clock_sampler") or not args or args[-1] is not C.c_void_p or res is not C.c_int:
        continue
    if name.endswith(("_bytes", "_supported", "_nblk", "_slabs")):
        continue
    orig = getattr(lib, "_" + name)

    def wrapped(*a, _o=orig, _n=name):
        r = _o(*a)
        mark(_n, a[-1])
        return r
    setattr(lib, "_" + name, wrapped)
for tname in ("mm", "addmm", "bmm"):
    o = getattr(torch, tname)

    def tw(*a, _o=o, _n=tname, **k):
        r = _o(*a, **k)
        mark("torch." + _n, torch.cuda.current_stream().cuda_stream)
        return r
    setattr(torch, tname, tw)

for i in range(20):                      # settle with the markers in
    setv(i)
    state["on"] = True
    tr.step(raw, nf, labels)
torch.cuda.synchronize()
state["n"] = 0
log.clear()
NSTEP = 40
bounds = []
for i in range(NSTEP):
    setv(i)
    bounds.append(state["n"])
    mark("<step start>", main_stream)
    tr.step(raw, nf, labels)
bounds.append(state["n"])
torch.cuda.synchronize()
t = marks.cpu().numpy().reshape(-1, 2)[:, 0].astype(np.int64) / 100.0       # us
res = {0: {}, 1: {}}
steplen = {0: [], 1: []}
for i in range(NSTEP):
    v = i % 2
    ent = [e for e in log[bounds[i]:bounds[i + 1]] if e[2] == (main_stream or 0)]
    seen = {}
    for a, b in zip(ent[:-1], ent[1:]):
        k = (b[1], seen.get(b[1], 0)); seen[b[1]] = k[1] + 1
        res[v].setdefault(k, []).append(t[b[0]] - t[a[0]])
    if i + 1 < NSTEP:
        steplen[v].append(t[bounds[i + 1]] - t[bounds[i]])
print(f"{cfg}: step (marker to marker) own dx {np.mean(steplen[0]):.1f} us, library dx {np.mean(steplen[1]):.1f} us: difference "
      f"{np.mean(steplen[1]) - np.mean(steplen[0]):+.1f} us; {len(log) // NSTEP} markers per step")
diff = []
for k in res[0]:
    if k in res[1]:
        a, b = float(np.mean(res[0][k])), float(np.mean(res[1][k]))
        diff.append((a - b, a, b, k))
diff.sort()
print(f"sum over the main stream's intervals present in both (own - library): {sum(d[0] for d in diff):+.1f} us")
for d, a, b, k in diff[:8] + diff[-12:]:
    print(f"  {d:+7.1f} us   own {a:7.1f}  library {b:7.1f}   interval ending with {k[0]} #{k[1]}")
if os.environ.get("MARKER_TRACE_ALL") == "1":          # every interval of an even step, in launch order
    i = 2
    ent = [e for e in log[bounds[i]:bounds[i + 1]] if e[2] == (main_stream or 0)]
    seen = {}
    acc = 0.0
    print("main stream, launch order: offset in the step, mean interval (both variants where equal), launch")
    for a, b in zip(ent[:-1], ent[1:]):
        k = (b[1], seen.get(b[1], 0)); seen[b[1]] = k[1] + 1
        m = float(np.mean(res[0][k]))
        acc += m
        print(f"  {acc:8.1f}  {m:7.1f}  {k[0]} #{k[1]}")
only0 = [(float(np.mean(v)), k) for k, v in res[0].items() if k not in res[1]]
only1 = [(float(np.mean(v)), k) for k, v in res[1].items() if k not in res[0]]
print("only in own-dx steps:", [(round(a, 1), k[0]) for a, k in only0], " only in library-dx steps:", [(round(a, 1), k[0]) for a, k in only1])
