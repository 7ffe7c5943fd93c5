#!/usr/bin/env python3
"""Round-5 probe for the two-product fp16 form of NetVladV1's dense GEMMs, run on the GPU box:

  1. do the matrix cores keep fp16 subnormals (library GEMM on subnormal operands)?
  2. library GEMM time: bf16 images with a 3K reduction against fp16 images with a 2K reduction, at the encoder's shapes;
  3. the magnitudes of every tensor the encoder GEMMs read as a split operand (activations and gradients), over a short cfg-2 run on
     rotating batches -- what a power-of-two loss scale has to hold inside fp16's range.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def denormals():
    dev = "cuda"
    out = {}
    for n in (256, 1024):
        a = torch.full((n, n), 2.0 ** -20, dtype=torch.float16, device=dev)      # subnormal in fp16 (min normal 2^-14)
        b = torch.zeros((n, n), dtype=torch.float16, device=dev)
        b.fill_diagonal_(1024.0)
        c = torch.mm(a, b, out_dtype=torch.float32)
        out[f"n{n}"] = {"expected": 2.0 ** -10, "got": float(c[3, 5]), "kept": bool(abs(float(c[3, 5]) - 2.0 ** -10) < 1e-9)}
    return out


def gemm_times():
    dev = "cuda"
    res = []

    def t(fn, n=20):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    M = 20480
    for (K, N, what) in ((1024, 3072, "qkv fwd"), (1024, 1024, "o fwd"), (4096, 1024, "ffn2 fwd"), (3072, 1024, "qkv dx"),
                         (1024, 4096, "ffn1 fwd (own tile GEMM today)")):
        row = {"what": what, "M": M, "K": K, "N": N}
        for dt, planes, name in ((torch.bfloat16, 3, "bf16x3"), (torch.float16, 2, "fp16x2")):
            a = torch.randn(M, planes * K, device=dev).to(dt)
            w = torch.randn(N, planes * K, device=dev).to(dt)
            us = t(lambda: torch.mm(a, w.t(), out_dtype=torch.float32))
            row[name + "_us"] = round(us, 1)
            row[name + "_executed_pflops"] = round(2.0 * M * planes * K * N / us / 1e9, 3)
        # strided form: an fp16 image that keeps the 3K row stride (first two planes read)
        a = torch.randn(M, 3 * K, device=dev).to(torch.float16)[:, :2 * K]
        w = torch.randn(N, 3 * K, device=dev).to(torch.float16)[:, :2 * K]
        row["fp16x2_strided_us"] = round(t(lambda: torch.mm(a, w.t(), out_dtype=torch.float32)), 1)
        res.append(row)
    # weight gradient: dW = x^T dy.  bf16x3: [3M, K]^T [3M, N] in S slices; fp16x2: x image [M, 2K] against dy_h [M, N] (row stride 2N) in S slices
    for (K, N, what) in ((1024, 3072, "qkv dW"), (1024, 1024, "o dW"), (4096, 1024, "ffn2 dW"), (1024, 4096, "ffn1 dW")):
        row = {"what": what, "M": M, "K": K, "N": N}
        S = 8 if K * N <= (1 << 20) else 4
        x3 = torch.randn(M, 3 * K, device=dev).to(torch.bfloat16)
        d3 = torch.randn(M, 3 * N, device=dev).to(torch.bfloat16)
        xb, db = x3.view(S, 3 * M // S, K), d3.view(S, 3 * M // S, N)
        row["bf16x3_us"] = round(t(lambda: torch.bmm(xb.transpose(1, 2), db, out_dtype=torch.float32)), 1)
        x2 = torch.randn(M, 2 * K, device=dev).to(torch.float16)
        d2 = torch.randn(M, 2 * N, device=dev).to(torch.float16)
        for S2 in (2, 4, 8):
            xb2 = x2.view(S2, M // S2, 2 * K)
            db2 = d2.view(S2, M // S2, 2 * N)[:, :, :N]
            row[f"fp16x2_S{S2}_us"] = round(t(lambda: torch.bmm(xb2.transpose(1, 2), db2, out_dtype=torch.float32)), 1)
        # the other one-sided form: x_h^T [dy_h | dy_l]  (output [K, 2N])
        for S2 in (2, 4, 8):
            xb2 = x2.view(S2, M // S2, 2 * K)[:, :, :K]
            db2 = d2.view(S2, M // S2, 2 * N)
            row[f"fp16x2_xh_S{S2}_us"] = round(t(lambda: torch.bmm(xb2.transpose(1, 2), db2, out_dtype=torch.float32)), 1)
        res.append(row)
    return res


def magnitudes(steps=60):
    import bench
    from learnablepoolingmethods_amd import FLAGS, ops, registry
    from learnablepoolingmethods_amd.train import Trainer
    dev = torch.device("cuda", 0)
    wl = bench.WORKLOADS["cfg2"]
    FLAGS.ln_gradient_image = False
    FLAGS.mha_gradient_image = False
    ops.FFN_TILES = False
    log = {}
    cur = {"step": 0}
    orig = ops._split_rows

    def spy(x2d, bias=None, relu=False, grad=False, row_scale=None):
        if cur["step"] in cur["watch"]:
            v = x2d.detach()
            if row_scale is not None:
                v = v * row_scale[:, None]
            if bias is not None:
                v = v + bias
            if relu:
                v = v.clamp_min(0)
            a = v.abs()
            amax = float(a.max())
            nz = a[a > 0]
            key = f"{'grad' if grad else 'act'} [{x2d.shape[0]}, {x2d.shape[1]}]"
            rec = {"step": cur["step"], "amax": amax, "rms": float((v.double() ** 2).mean().sqrt()),
                   "min_nonzero": float(nz.min()) if nz.numel() else 0.0,
                   "frac_below_2^-24_amax": float((a < amax * 2.0 ** -24).float().mean()),
                   "frac_below_2^-12_amax": float((a < amax * 2.0 ** -12).float().mean())}
            log.setdefault(key, []).append(rec)
        return orig(x2d, bias=bias, relu=relu, grad=grad, row_scale=row_scale)
    ops._split_rows = spy
    # the FFN hidden gradient enters through lpm_split_rows_relu_bwd (df in fp32): watched where _FFNX3.backward makes it
    orig_mm3 = ops._mm3

    def spy_mm3(a3, w3, acc=None):
        out = orig_mm3(a3, w3, acc)
        if cur["step"] in cur["watch"] and cur.get("bwd"):
            a = out.detach().abs()
            key = f"gemm out in backward [{out.shape[0]}, {out.shape[1]}]"
            log.setdefault(key, []).append({"step": cur["step"], "amax": float(a.max()), "rms": float((out.double() ** 2).mean().sqrt())})
        return out
    ops._mm3 = spy_mm3
    trainer = Trainer(registry.get_model("NetVladV1"), vocab_size=bench.VOCAB, batch_size=wl["batch"], device=dev, seed=1234,
                      model_kwargs=wl["model_kwargs"], **bench.TRAIN)
    batches = [bench.synthetic_batch(wl["batch"], dev, seed=i) for i in range(8)]
    cur["watch"] = {0, 1, 5, 20, steps - 1}
    losses = []
    for s in range(steps):
        cur["step"] = s
        cur["bwd"] = True          # (forward mm3 outputs are logged too: harmless)
        out = trainer.step(*batches[s % 8])
        losses.append(round(float(out["loss"]), 4))
    ops._split_rows, ops._mm3 = orig, orig_mm3
    # gradient arena magnitudes per variable (last step)
    per_var = {}
    for n in trainer.arena.names[1:]:
        g = trainer.gradient(n)
        per_var[n] = {"amax": float(g.abs().max()), "norm": float(g.norm())}
    return {"losses": losses, "operands": log, "gradients_last_step": per_var}


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    out = {}
    if what in ("all", "denormals"):
        out["fp16_subnormals_in_library_gemm"] = denormals()
        print(json.dumps(out["fp16_subnormals_in_library_gemm"]), flush=True)
    if what in ("all", "gemms"):
        out["gemm_times"] = gemm_times()
        for r in out["gemm_times"]:
            print(json.dumps(r), flush=True)
    if what in ("all", "magnitudes"):
        out["magnitudes"] = magnitudes()
        print(json.dumps(out["magnitudes"]["losses"]))
        for k, v in out["magnitudes"]["operands"].items():
            print(k)
            for r in v:
                print("   ", json.dumps(r))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fp16_probe.json"), "w"), indent=1)
