"""K2 at cfg-2's video shape (80 x 300 x 1024 x 256), the forms side by side in ONE process, interleaved rounds (guide 5.4 rule 24):
  chain : lpm_vlad_aggregate_raw_kmajor_fwd (128 x 128 items) + lpm_vlad_row_scales         -- two launches
  none / rounds / all : lpm_vlad_aggregate_kmajor_scaled_fwd with no / whole rounds of / all clips as wide (256 x 128) items
  clip  : lpm_vlad_aggregate_clip_kmajor_fwd (round 4: all clusters x a third of a clip's columns) + lpm_vlad_row_scales
(K2_FORMS=chain,... restricts the library forms; LPM_VC_DBG ablates the clip form)
Kernel durations from launch-attached HIP events (the library's timing tags)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import _capi, ops
from learnablepoolingmethods_amd.ops import ptr, stream_ptr

dev = torch.device("cuda:0")
lib = _capi.load()
B, T, D, K = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (80, 300, 1024, 256)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B * T, D, device=dev, generator=g)
x = x / x.norm(dim=1, keepdim=True)
logits = torch.randn(B * T, K, device=dev, generator=g) * 3
centres = torch.randn(D, K, device=dev, generator=g) * 0.05
xt = torch.empty(lib._lpm_xt_bytes(B, T, D) // 4, dtype=torch.int32, device=dev)
lib.check(lib._lpm_split_frames(ptr(x), D, B, T, D, ptr(xt), stream_ptr()), "split")
at = torch.empty(lib._lpm_at_bytes(B, T, K) // 4, dtype=torch.int32, device=dev)
lib.check(lib._lpm_assign_tiles(ptr(logits), None, None, B, T, K, ops.LPM_VLAD_SOFTMAX | ops.LPM_VLAD_RESIDUAL, ptr(at), stream_ptr()), "at")
P = D // 128
raw = torch.empty(B, K, D, device=dev)
asum, rs, colsq, csq = (torch.empty(B, K, device=dev) for _ in range(4))
gsq = torch.empty(B, device=dev)
part = torch.empty(B, P, K, device=dev)
wsb = lib._lpm_vlad_kmajor_workspace_bytes(B, D, K)
ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
filler = torch.empty(256 << 20, dtype=torch.uint8, device=dev)      # evicts the operands from the caches between launches, as the step does


def chain():
    lib.check(lib._lpm_vlad_aggregate_raw_kmajor_fwd(ptr(at), ptr(xt), ptr(centres), B, T, D, K, ops.LPM_VLAD_RESIDUAL, ptr(raw), ptr(asum),
                                                     ptr(part), stream_ptr()), "raw")
    lib.check(lib._lpm_vlad_row_scales(ptr(part), P, B, K, ptr(rs), ptr(colsq), ptr(csq), ptr(gsq), stream_ptr()), "rs")


def one(flag):
    def f():
        lib.check(lib._lpm_vlad_aggregate_kmajor_scaled_fwd(ptr(at), ptr(xt), ptr(centres), B, T, D, K, ops.LPM_VLAD_RESIDUAL | flag, ptr(raw),
                                                            ptr(rs), ptr(asum), ptr(colsq), ptr(csq), ptr(gsq), ptr(ws), wsb, stream_ptr()), "one")
    return f


Pc = lib._lpm_vlad_clip_slabs(D, K)
partc = torch.empty(B, max(Pc, 1), K, device=dev)


def clip():
    lib.check(lib._lpm_vlad_aggregate_clip_kmajor_fwd(ptr(at), ptr(xt), ptr(centres), B, T, D, K, ops.LPM_VLAD_RESIDUAL, ptr(raw), ptr(asum),
                                                      ptr(partc), stream_ptr()), "clip")
    lib.check(lib._lpm_vlad_row_scales(ptr(partc), Pc, B, K, ptr(rs), ptr(colsq), ptr(csq), ptr(gsq), stream_ptr()), "rs")


forms = {"chain": chain, "none": one(_capi.LPM_VLAD_WIDE_NONE), "rounds": one(0), "all": one(_capi.LPM_VLAD_WIDE_ALL)}
if os.environ.get("K2_FORMS"):
    forms = {k: v for k, v in forms.items() if k in os.environ["K2_FORMS"].split(",")}
if Pc:
    forms["clip"] = clip
res = {k: [] for k in forms}
buf = (ctypes.c_float * 64)()
for rnd in range(12):
    for name, fn in forms.items():
        filler.fill_(rnd & 255)
        lib._lpm_kernel_timing_enable(1)
        fn()
        torch.cuda.synchronize()
        lib._lpm_kernel_timing_enable(0)
        t = sum(buf[i] for i in range(lib._lpm_kernel_timing_read(2, buf, 64)))
        t4 = sum(buf[i] for i in range(lib._lpm_kernel_timing_read(4, buf, 64)))
        if rnd >= 2:
            res[name].append((t * 1e3, t4 * 1e3))
alg = 4 * (B * T * K + B * T * D + B * D * K) + 4 * D * K
for name, v in res.items():
    k2 = sorted(a for a, _ in v)
    tot = sorted(a + b for a, b in v)
    print(f"{name:7s} K2 median {k2[len(k2) // 2]:6.1f} us (min {k2[0]:6.1f})  K2 + row scales median {tot[len(tot) // 2]:6.1f} us  -> "
          f"{alg / tot[len(tot) // 2] / 1e6:5.2f} TB/s on {alg / 1e6:.1f} MB algorithmic", flush=True)
