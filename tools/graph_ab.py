"""the N = 1 training step captured ONCE in a hipGraph and replayed, against the same step launched eagerly, on one box.
A MEASUREMENT, not a product path: a replayed graph repeats the captured kernel arguments -- the learning rate, Adam's bias correction,
the operand scales and the batch pointers are those of the captured step -- so it answers "what do 167 launches cost the step" and
nothing else.  Host-side reads the step normally makes (the delayed operand-scale read-back, the min |gamma| watch) are stubbed for the
capture.
  python tools/graph_ab.py [cfg2|cfg3|cfg5] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from learnablepoolingmethods_amd import frame_level_models, ops, registry
from learnablepoolingmethods_amd.train import Trainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
wl = bench.WORKLOADS[cfg]
bench.set_flags(wl)
if os.environ.get("LPM_SINGLE_STREAM") == "1":     # as bench.py: the audio branch and the hidden1 update on the main stream
    from learnablepoolingmethods_amd import FLAGS as _flags
    _flags.audio_side_stream = False
    _flags.hidden1_update_stream = False
dev = torch.device("cuda:0")
tr = Trainer(registry.get_model(wl.get("model", "NetVladV1")), vocab_size=bench.VOCAB, batch_size=wl["batch"], device=dev, seed=1234,
             model_kwargs=wl["model_kwargs"], **bench.TRAIN)
batches = [bench.synthetic_batch(wl["batch"], dev, seed=i) for i in range(8)]
for i in range(60):
    tr.step(*batches[i % 8])
torch.cuda.synchronize()


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager_rot = timed(lambda i: tr.step(*batches[i % 8]), steps)
eager_one = timed(lambda i: tr.step(*batches[0]), steps)
print(f"{cfg}: eager {eager_rot:.3f} ms/step (8 rotating batches), {eager_one:.3f} ms/step (one batch)", flush=True)

# ---- capture
sc = tr.operand_scales
if sc is not None:
    sc._harvest(wait=True)

    def frozen_begin_step():
        sc.step += 1
        sc.sites = {}
        sc.fp16_now = bool(sc.enabled and sc.slots)
        if sc.fp16_now:
            sc.steps_fp16 += 1
    sc.begin_step = frozen_begin_step
frame_level_models._GammaWatch.ok = lambda self, gamma: True
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
graph = torch.cuda.CUDAGraph()
try:
    with torch.cuda.stream(side):
        for _ in range(3):                      # warm the allocator on the capture stream
            tr.step(*batches[0])
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="relaxed"):
            out = tr.step(*batches[0])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
except Exception as e:
    print(f"{cfg}: capture FAILED: {type(e).__name__}: {str(e)[:600]}", flush=True)
    sys.exit(0)
loss0 = float(out["loss"])
graph.replay()
torch.cuda.synchronize()
replay = timed(lambda i: graph.replay(), steps)
loss1 = float(out["loss"])
eager_one2 = timed(lambda i: tr.step(*batches[0]), steps)
print(f"{cfg}: hipGraph replay {replay:.3f} ms/step (captured step repeated {steps} x on one batch; loss {loss0:.4f} -> {loss1:.4f}); "
      f"eager, same batch, after: {eager_one2:.3f} ms/step; graph / eager = {replay / min(eager_one, eager_one2):.3f}", flush=True)
