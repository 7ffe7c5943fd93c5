"""Run PyTorch TunableOp over every library GEMM shape of the cfg-2 training step (hipBLASLt / rocBLAS solution search)
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

model = registry.get_model("NetVladV1")
trainer = Trainer(model, vocab_size=bench.VOCAB, batch_size=bench.PER_GPU_BATCH, device=dev, seed=1234, model_kwargs=bench.CFG, **bench.TRAIN)
raw, nf, labels = bench.synthetic_batch(bench.PER_GPU_BATCH, dev, seed=0)


def timed(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        trainer.step(raw, nf, labels)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    trainer.step(raw, nf, labels)
print("before: %.3f ms/step" % timed(8), flush=True)
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
print("after: %.3f ms/step" % timed(8), flush=True)
tunable.write_file() if hasattr(tunable, "write_file") else None
for r in tunable.get_results()[:60]:
    print(r)
