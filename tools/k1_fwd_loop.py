"""Run K1's forward (tile form) alone at cfg-2's shape N times: for rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnablepoolingmethods_amd import _capi
from learnablepoolingmethods_amd._capi import ptr, stream_ptr

lib = _capi.load()
dev = torch.device("cuda:0")
B, T, D, K = 80, 300, 1024, 256
M = B * T
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
zero = os.environ.get("K1_ZERO") == "1"
x = torch.zeros(M, D, device=dev) if zero else torch.randn(M, D, device=dev)
W = torch.zeros(D, K, device=dev) if zero else torch.randn(D, K, device=dev) / 32
logits = torch.empty(M, K, device=dev)
st = stream_ptr()
xr = torch.empty(lib._lpm_row_tiles_bytes(B, T, D) // 4, dtype=torch.int32, device=dev)
wt = torch.empty(lib._lpm_weight_tiles_bytes(D, K) // 4, dtype=torch.int32, device=dev)
p1 = torch.empty(lib._lpm_assign_gemm_tiles_nblk(B, T), 2, K, device=dev)
lib._lpm_split_rows_tiles(ptr(x), x.stride(0), B, T, D, ptr(xr), st)
lib._lpm_split_weight_tiles(ptr(W), D, K, 0, ptr(wt), st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(iters + 5):
    if i == 5:
        e0.record()
    lib.check(lib._lpm_assign_gemm_tiles_fwd(ptr(xr), ptr(wt), B, T, D, K, ptr(logits), ptr(p1), st), "k1")
e1.record()
torch.cuda.synchronize()
print(f"k1 fwd back-to-back {e0.elapsed_time(e1) / iters * 1e3:.1f} us; err vs fp32 matmul "
      f"{float((logits - x @ W).abs().max()):.2e}")
