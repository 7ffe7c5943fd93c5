"""Run PyTorch TunableOp over every library GEMM shape of a bench.py workload's training step (cfg2 | cfg3 | cfg5) (hipBLASLt / rocBLAS solution search)
and write the selections to gpurun_out/tunableop_results.csv.  Prints the step time before and after."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tunable
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
from learnablepoolingmethods_amd import registry
from learnablepoolingmethods_amd.train import Trainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
wl = bench.WORKLOADS[cfg]
bench.set_flags(wl)
trainer = Trainer(registry.get_model(wl.get("model", "NetVladV1")), vocab_size=bench.VOCAB, batch_size=wl["batch"], device=dev, seed=1234,
                  model_kwargs=wl["model_kwargs"], **bench.TRAIN)
raw, nf, labels = bench.synthetic_batch(wl["batch"], dev, seed=0)


def timed(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        trainer.step(raw, nf, labels)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    trainer.step(raw, nf, labels)
print("before: %.3f ms/step" % timed(30), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
tunable.set_filename("gpurun_out/tunableop_results.csv")
tunable.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "30")))
tunable.set_max_tuning_iterations(int(os.environ.get("TUNE_ITERS", "20")))
tunable.enable(True)
tunable.tuning_enable(True)
t0 = time.perf_counter()
trainer.step(raw, nf, labels)
torch.cuda.synchronize()
print("tuning step took %.1f s, %d entries" % (time.perf_counter() - t0, len(tunable.get_results())), flush=True)
tunable.tuning_enable(False)
for _ in range(2):
    trainer.step(raw, nf, labels)
print("after: %.3f ms/step" % timed(30), flush=True)
tunable.write_file() if hasattr(tunable, "write_file") else None
for r in tunable.get_results()[:60]:
    print(r)
