"""K2 forms inside the model: one NetVladV1 step with ops.VLAD_CLIP on / off, per-variable gradient differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import lpm_oracle as O
from learnablepoolingmethods_amd import FLAGS, ops, registry
from learnablepoolingmethods_amd.train import Trainer

dev = torch.device("cuda:0")
B, MF, it = 16, 40, int(sys.argv[1]) if len(sys.argv) > 1 else 32
x, nf, lab = O.make_synthetic_batch(B, MF, 1152, 50, seed=23, min_frames=10)
res = {}
mode = sys.argv[2] if len(sys.argv) > 2 else "clip"
for clip in (True, False):
    if mode == "clip":
        ops.VLAD_CLIP = clip
    else:                      # the round-3 one-launch form (another summation order of the assignment sums) against the 128 x 128 chain
        ops.VLAD_CLIP = False
        ops.VLAD_KMAJOR_SCALED = clip
    tr = Trainer(registry.get_model("NetVladV1"), vocab_size=50, batch_size=B, base_learning_rate=1e-3, device=dev, seed=13,
                 model_kwargs=dict(iterations=it, cluster_size=256, hidden_size=64))
    tr.build(x, nf, lab)
    tr.store.summaries = {}
    loss = tr.step(x, nf, lab)["loss"].item()
    torch.cuda.synchronize()
    res[clip] = (loss, {n: tr.gradient(n).clone() for n in tr.arena.names}, {k: v.clone() for k, v in tr.store.summaries.items()})
print("loss", res[True][0], res[False][0])
for n in res[True][1]:
    a, b = res[True][1][n], res[False][1][n]
    d = float((a - b).norm() / b.norm().clamp_min(1e-30))
    if d > 1e-6:
        print(f"{n:60s} rel l2 {d:.3e}")
for k in res[True][2]:
    a, b = res[True][2][k].float(), res[False][2][k].float()
    print(f"summary {k:30s} rel {float((a - b).abs().max() / b.abs().max()):.3e}")
