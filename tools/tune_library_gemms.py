"""Regenerates learnablepoolingmethods_amd/_tunable/gfx950_fp32_gemm.csv: PyTorch's TunableOp searches the hipBLASLt / rocBLAS solutions of every
fp32 library GEMM a training step of the three benchmark configurations issues (MoE head, context gating, small-batch projections) and
records the fastest per shape; candidates whose result differs from the default solution's are rejected.  Run on the GPU box:
    python tools/tune_library_gemms.py [out.csv]        (default gpurun_out/tunable/gfx950_fp32_gemm.csv; copy it into the package afterwards)
ops.enable_library_gemm_selection() reads the file with tuning OFF."""
import os, sys
out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tunable/gfx950_fp32_gemm.csv")
os.makedirs(os.path.dirname(out), exist_ok=True)
if os.path.exists(out):
    os.remove(out)
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"           # (also keeps ops.enable_library_gemm_selection out of the way)
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = out
os.environ["PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS"] = "100"
os.environ["PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS"] = "10"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tunable
import bench
from learnablepoolingmethods_amd import FLAGS, registry
from learnablepoolingmethods_amd.train import Trainer

tunable.set_filename(out, insert_device_ordinal=False)
try:
    tunable.set_numerical_check_tolerances(True, 1e-3, 1e-4)
except Exception as e:
    print("numerical check not available:", e)
dev = torch.device("cuda:0")
for cfg in ("cfg2", "cfg3", "cfg5"):
    wl = bench.WORKLOADS[cfg]
    FLAGS.reset()
    bench.set_flags(wl)
    tr = Trainer(registry.get_model(wl.get("model", "NetVladV1")), vocab_size=bench.VOCAB, batch_size=wl["batch"], device=dev, seed=1234,
                 model_kwargs=wl["model_kwargs"], **bench.TRAIN)
    raw, nf, labels = bench.synthetic_batch(wl["batch"], dev, seed=0)
    for _ in range(4):
        tr.step(raw, nf, labels)
    torch.cuda.synchronize()
    print(cfg, "tuned;", len(tunable.get_results()), "entries so far", flush=True)
    del tr
    torch.cuda.empty_cache()
print("results are written to", out, "when the process exits")
