"""Is a training step a pure function of its inputs?  No kernel on this path uses floating-point atomics, so from the same weights,
optimiser state and batch every run must produce the same bits -- anything else is a race (or memory read before it was written).
One process: build the trainer, remember (parameters, Adam slots, moving statistics), then N times: restore, step, compare the
gradient arena / updated parameters / loss with the first run bit for bit.
  python tools/determinism_check.py [cfg2|cfg3|cfg5|blocks] [runs] [side 0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from learnablepoolingmethods_amd import FLAGS, registry
from learnablepoolingmethods_amd.train import Trainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
side = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
TAP = len(sys.argv) > 4 and sys.argv[4] == "tap"      # copies of the video pooling backward's intermediates are compared too (ops.DEBUG_TAP)
from learnablepoolingmethods_amd import ops
dev = torch.device("cuda:0")
if cfg == "blocks":                                   # the small tower of the two-rank tests (tests/dp_cases.py "blocks"): 16 clips x 40 frames,
    FLAGS.audio_side_stream = side                    # K = 256, both encoders as block Functions, closed-form input_bn gradients
    per, max_frames, vocab = 16, 40, 50
    tr = Trainer(registry.get_model("NetVladV1"), vocab_size=vocab, batch_size=per, base_learning_rate=2e-4, device=dev, seed=100,
                 model_kwargs=dict(iterations=32, cluster_size=256, hidden_size=64))
    g = torch.Generator(device=dev).manual_seed(33)
    nf = torch.randint(max_frames // 3, max_frames + 1, (per,), device=dev, generator=g, dtype=torch.int32)
    raw = torch.randn(per, max_frames, 1152, device=dev, generator=g) * (torch.arange(max_frames, device=dev)[None, :, None] < nf[:, None, None])
    labels = torch.rand(per, vocab, device=dev, generator=g) < 0.06
else:
    wl = bench.WORKLOADS[cfg]
    bench.set_flags(wl)
    FLAGS.audio_side_stream = side
    tr = Trainer(registry.get_model(wl.get("model", "NetVladV1")), vocab_size=bench.VOCAB, batch_size=wl["batch"], device=dev, seed=1234,
                 model_kwargs=wl["model_kwargs"], **bench.TRAIN)
    raw, nf, labels = bench.synthetic_batch(wl["batch"], dev, seed=0)
torch.manual_seed(7)
tr.step(raw, nf, labels)                              # builds; the state after this step is the starting point
torch.cuda.synchronize()
# NetVladV1: the repeated step runs its encoder GEMMs on fp16 planes with scales measured on this batch (ops.OperandScales: a run's scales
# are a function of the maxima of earlier steps only -- here every repetition sees the same ones)
torch.manual_seed(11)                                 # (the calibration pass draws the dropout masks of the repeated step: same maxima, same scales)
if tr.calibrate_operand_scales(raw, nf, labels):
    print(f"operand scales calibrated: {len(tr.operand_scales.slots)} sites", flush=True)
a = tr.arena
start = dict(param=a.param.clone(), m=a.m.clone(), v=a.v.clone(), step=tr.global_step,
             stats={n: t.detach().clone() for n, t in tr.store.vars.items() if not tr.store.trainable[n]})
first, bad = None, 0
for r in range(runs):
    a.param.copy_(start["param"]); a.m.copy_(start["m"]); a.v.copy_(start["v"]); tr.global_step = start["step"]
    for n, t in start["stats"].items():
        tr.store.vars[n].data.copy_(t)
    torch.manual_seed(11)                             # dropout draws
    if TAP:
        ops.DEBUG_TAP = {}
    out = tr.step(raw, nf, labels)
    torch.cuda.synchronize()
    got = dict(grad=a.grad.clone(), param=a.param.clone(), loss=out["loss"].detach().clone(), pred=out["predictions"].detach().clone())
    if TAP:
        t = ops.DEBUG_TAP
        if t.get("dots") is not None:                # the k-major form's column dots <dO,N>, <dO,W2>, <N,W2> against torch, run by run
            Bk, Kk, Dk = t["colsq"].shape[0], t["colsq"].shape[1], t["nrm"].shape[-1]
            Nn = t["nrm"].float().view(Bk, Kk, Dk) * torch.rsqrt(t["colsq"].clamp_min(1e-12))[..., None]
            dO, W2 = t["dout"].view(Bk, Kk, Dk), t["centres"].view(Dk, Kk)
            ref = torch.stack([(dO * Nn).sum(-1), torch.einsum("bkd,dk->bk", dO, W2), torch.einsum("bkd,dk->bk", Nn, W2)], 1)
            dots = t["dots"].view(Bk, 3, Kk)
            nbad = ((dots - ref).abs() > 1e-4 * ref.abs().amax(dim=(0, 2), keepdim=True)).sum(dim=(0, 2)).tolist()
            if any(nbad):
                err = ((dots - ref).abs().amax(dim=(0, 2)) / ref.abs().amax(dim=(0, 2))).tolist()
                print(f"run {r}: K3 column dots vs torch: entries off by > 1e-4 per component {nbad}, relative max error {[f'{e:.1e}' for e in err]}")
        got.update({"tap/" + k: v for k, v in ops.DEBUG_TAP.items() if v is not None})
        ops.DEBUG_TAP = None
    if first is None:
        first = got
        continue
    for k, v in got.items():
        if not torch.equal(v, first[k]):
            bad += 1
            d = (v.double() - first[k].double()).abs()
            where = ""
            if k in ("grad", "param"):
                i = int(d.argmax())
                name = next((n for n in a.names if a.segment(n)[0] <= i < a.segment(n)[0] + a.views[n].numel()), "?")
                differing = sorted({n for n in a.names if not torch.equal(v[a.segment(n)[0]:a.segment(n)[0] + a.views[n].numel()],
                                                                          first[k][a.segment(n)[0]:a.segment(n)[0] + a.views[n].numel()])})
                where = f" largest in {name}; differing variables: {differing[:12]}{' ...' if len(differing) > 12 else ''} ({len(differing)})"
            elif k.startswith("tap/"):
                idx = torch.nonzero(d.flatten() > 0).flatten()
                where = f" {idx.numel()} of {d.numel()} entries, flat indices {idx[:6].tolist()} .. {idx[-3:].tolist()}, shape {tuple(v.shape)}"
            print(f"run {r}: {k} differs from run 0: max abs {float(d.max()):.3e} (scale {float(first[k].double().abs().max()):.3e}){where}")
print(f"{cfg} side_stream={side}: {runs} runs, {bad} differing tensors" + ("" if bad else " -- bit-identical"))
sys.exit(1 if bad else 0)
