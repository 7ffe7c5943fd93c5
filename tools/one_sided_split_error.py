"""VERDICT r2 item 4, the side experiment: would a 2-MFMA one-sided split (one operand hi + lo, the other hi only) do for the weight /
input gradient GEMMs under the 1e-3 bar?  Exact emulation in fp64 of what the matrix pipe would accumulate, at an encoder dW shape
(reduction over M = 20 480 rows).  Runs on the CPU in seconds.  Result (committed in DESIGN.md section 4): 1.7e-3 Frobenius for either
one-sided form against 4.4e-6 for the three-product form -- the bf16 rounding of the un-split operand (2^-9 / sqrt 3 per element) goes
straight into the result because gradient sums cancel; the one-sided split does not fit the bar and was not built."""
import torch

torch.manual_seed(0)
M, K, N = 20480, 512, 512
x = torch.randn(M, K, dtype=torch.float64)
dy = torch.randn(M, N, dtype=torch.float64) * torch.rand(M, 1, dtype=torch.float64)


def split(t):
    hi = t.float().bfloat16()
    lo = (t.float() - hi.float()).bfloat16()
    return hi.double(), lo.double()


xh, xl = split(x)
dh, dl = split(dy)
ref = x.t() @ dy
forms = [("3 products (x_h dy_h + x_l dy_h + x_h dy_l)", xh.t() @ dh + xl.t() @ dh + xh.t() @ dl),
         ("2 products, dy hi only", xh.t() @ dh + xl.t() @ dh),
         ("2 products, x hi only", xh.t() @ dh + xh.t() @ dl),
         ("1 product (plain bf16)", xh.t() @ dh)]
for name, v in forms:
    print(f"{name}: Frobenius {float((v - ref).norm() / ref.norm()):.2e}, max / scale {float((v - ref).abs().max() / ref.abs().max()):.2e}")
