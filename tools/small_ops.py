"""Which Python lines issue the small torch launches of a training step?  (torch.profiler has no Python stacks in this build.)  The
Python entry points of the usual suspects are wrapped for ONE step and every call on a CUDA tensor is counted under its innermost repo
frames; what the autograd engine issues from C++ (AccumulateGrad copies, zero-filled undefined gradients, the backward of plain torch
ops) does not pass through here -- the difference to the profiler's per-step counts is theirs.  Usage: python tools/small_ops.py [cfg2|cfg3|cfg5]"""
import collections, os, sys, traceback
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
for _ in range(5):
    tr.step(raw, nf, labels)
torch.cuda.synchronize()
sites = collections.Counter()


def where():
    fr = [f for f in traceback.extract_stack()[:-2] if "learnablepoolingmethods_amd" in f.filename]
    return " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr[-3:])) or "(no repo frame)"


def wrap(owner, name, label):
    orig = getattr(owner, name)

    def f(*a, **k):
        t = next((x for x in list(a) + list(k.values()) if torch.is_tensor(x)), None)
        devs = str(k.get("device", "")) + (str(t.device) if t is not None else "")
        if name in ("contiguous", "float", "to", "t") and t is not None:
            # only the calls that launch something: a real copy / cast
            noop = (name == "t") or (name == "contiguous" and t.is_contiguous()) or (name == "float" and t.dtype == torch.float32) \
                or (name == "to" and not any(isinstance(x, torch.dtype) and x != t.dtype for x in list(a) + list(k.values()))
                    and not any(isinstance(x, (torch.device, str)) for x in list(a[1:]) + list(k.values())))
            if noop:
                return orig(*a, **k)
            label2 = f"{label} {tuple(t.shape)} {t.numel() * t.element_size() >> 20} MiB"
            sites[(label2, where())] += 1
            return orig(*a, **k)
        if "cuda" in devs or (t is None and name in ("zeros", "full", "empty")):
            sites[(label, where())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)
    return orig


saved = []
for owner, names in ((torch, ("zeros", "zeros_like", "full", "cat", "sum", "stack", "where", "sigmoid", "clamp")),
                     (torch.Tensor, ("zero_", "fill_", "sum", "add_", "copy_", "mul", "__mul__", "__rmul__", "__add__", "__radd__", "__sub__", "__truediv__",
                                     "to", "contiguous", "clone", "float", "mean", "sigmoid", "matmul", "t"))):
    for n in names:
        saved.append((owner, n, wrap(owner, n, f"{owner.__name__}.{n}")))
tr.step(raw, nf, labels)
torch.cuda.synchronize()
for owner, n, orig in saved:
    setattr(owner, n, orig)
print(f"{cfg}: Python-level calls on CUDA tensors in one step (contiguous / to / t may be no-ops)")
for (label, w), c in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{c:4d}  {label:22s} {w}")
