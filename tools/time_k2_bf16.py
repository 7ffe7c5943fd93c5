"""K2 for bf16 storage alone at BASELINE configs[4]'s per-GPU shape (B = 128, T = 300, D = 1024, K = 512): the clip-wide form
(csrc/vlad_clip16.hip) against the 128 x 128 form, with and without the residual term (the centres' reads), a cache-evicting pass between
launches (as the step's other kernels do).  HIP events around single launches, median of `iters`.
  python tools/time_k2_bf16.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import _capi, ops

lib = _capi.load()
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 15
B, T, D, K = 128, 300, 1024, 512
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B * T, D, device=dev, generator=g)
a = torch.rand(B * T, K, device=dev, generator=g).to(torch.bfloat16)
cen = torch.randn(D, K, device=dev, generator=g) / D ** 0.5
st = ops.stream_ptr()
steps = lib._lpm_frame_steps_bf16(T)
xt = torch.empty(lib._lpm_frame_tiles_bf16_bytes(B, T, D) // 4, dtype=torch.int32, device=dev)
lib.check(lib._lpm_split_frames_bf16(ops.ptr(x), D, B, T, D, ops.ptr(xt), st), "split_frames")
at = torch.empty(B * (K // 32) * steps * 256, dtype=torch.int32, device=dev)
lib.check(lib._lpm_assign_tiles_bf16(ops.ptr(a), None, None, B, T, K, 0, ops.ptr(at), st), "assign_tiles")
nrm = torch.empty((B, D, K), dtype=torch.bfloat16, device=dev)
asum = torch.empty((B, K), device=dev)
evict = torch.empty(512 * 1024 * 1024 // 4, device=dev)
alg = 2 * (B * T * K + B * T * D + B * D * K) + 4 * D * K


def run(form, flags):
    P = lib._lpm_vlad_clip16_slabs(D, K) if form == "clip" else D // 128
    part = torch.empty((B, P, K), device=dev)
    fn = lib._lpm_vlad_aggregate_clip_fwd_bf16 if form == "clip" else lib._lpm_vlad_aggregate_tiles3_fwd_bf16
    ts = []
    for _ in range(iters):
        evict.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.check(fn(ops.ptr(at), ops.ptr(xt), ops.ptr(cen), B, T, D, K, flags, ops.ptr(nrm), ops.ptr(asum), ops.ptr(part), st), "k2")
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for form in ("tiles3", "clip"):
    for flags, name in ((_capi.LPM_VLAD_RESIDUAL, "with residual"), (0, "no residual (no centre reads)")):
        t = run(form, flags)
        print(f"{form:7s} {name:32s} {t:7.1f} us   {alg / t / 1e6:6.2f} TB/s algorithmic = {alg / t / 1e6 / 8:.3f} of 8 TB/s", flush=True)
# the assignment tiles' own kernel (bf16 logits -> softmax -> tiles)
ts = []
for _ in range(iters):
    evict.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.check(lib._lpm_assign_tiles_bf16(ops.ptr(a), None, None, B, T, K, _capi.LPM_VLAD_SOFTMAX, ops.ptr(at), st), "assign_tiles")
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"assign_tiles_bf16 {ts[len(ts) // 2]:7.1f} us", flush=True)
