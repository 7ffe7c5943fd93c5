"""The committed bench lines (profiles/r06_bench_line.json, r06_bench_line_cfg3.json, r06_bench_line_cfg5.json, written by bench.py on the GPU box) keep the driver's contract:
required keys, BASELINE.json's metric, and internally consistent roofline / throughput figures.  CPU-only: it reads the
committed artefact, it does not run the bench."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r06_bench_line.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_bench_line_contract_keys():
    d = _line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == base["metric"]
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None                      # BASELINE.md publishes no number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["n_gpus"] == 1 and d["config"]["global_batch"] == 80


def test_bench_line_figures_are_consistent():
    d = _line()
    # whole-job throughput = clips per step / step time
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic bytes per launch / average launch duration (2.60 MB per clip x 80 clips + W2)
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["avg_kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 2e-2
    assert r["algorithmic_bytes"] == 4 * (80 * 300 * 256 + 80 * 300 * 1024 + 80 * 1024 * 256 + 1024 * 256)
    assert r["traffic"] is not None and 0.9 * r["algorithmic_bytes"] <= r["traffic"] <= 2.0 * r["algorithmic_bytes"]      # PMC-derived (profiles/a5_hbm_traffic_cfg2.json)
    pj = json.load(open(os.path.join(ROOT, "profiles", "a5_hbm_traffic_cfg2.json")))
    assert r["traffic"] == pj["k2_bytes_per_launch"] and pj["algorithmic_bytes"] == r["algorithmic_bytes"] and pj["commit"] != "unknown"
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    a = d["assign_gemm"]
    assert abs(a["mfma_util_vs_bf16_peak"] - a["executed_bf16_tflops"] / 2500.0) < 5e-3


def test_bench_line_round2_objects():
    """The whole a5 function beside the K2 kernel, the parity check of the run, and the cfg-5 line (bf16 storage, 2.165 MB/clip)."""
    d = _line()
    a5 = d["roofline"]["a5_function"]
    assert set(a5["kernel_ms"]) in ({"assign_tiles", "vlad_aggregate", "vlad_finalize"}, {"assign_tiles", "vlad_aggregate"})
    assert abs(a5["total_ms"] - sum(a5["kernel_ms"].values())) < 1e-3
    assert abs(a5["frac"] - a5["algorithmic_bytes"] / (a5["total_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-3
    assert a5["frac"] <= d["roofline"]["frac"]                   # the chain cannot beat its dominant kernel
    assert d["parity"]["ok"] is True and d["parity"]["predictions_max_rel_err"] <= d["parity"]["tolerance"] == 1e-3
    with open(os.path.join(ROOT, "profiles", "r06_bench_line_cfg5.json")) as f:
        c = json.loads(f.read().strip().splitlines()[-1])
    assert c["dtype"] == "bf16" and c["config"]["global_batch"] == 128 and "configs[4]" in c["metric"]
    r = c["roofline"]
    B, T, D, K = 128, 300, 1024, 512
    assert r["algorithmic_bytes"] == 2 * (B * T * K + B * T * D + B * D * K) + 4 * D * K        # video part of 2.165 MB/clip, bf16
    assert abs(r["algorithmic_bytes"] / B / 1e6 - 1.99) < 0.02
    assert c["parity"]["ok"] is True and c["parity"]["tolerance"] == 2e-2
    with open(os.path.join(ROOT, "profiles", "r06_bench_line_cfg3.json")) as f:
        v2 = json.loads(f.read().strip().splitlines()[-1])
    assert v2["dtype"] == "f32" and v2["config"]["global_batch"] == 80 and "configs[2]" in v2["metric"] and "NetVladV2" in v2["metric"]
    assert abs(v2["value"] - 80 / (v2["ms_per_step"] * 1e-3)) < 0.01 * v2["value"]
    assert v2["parity"]["ok"] is True and v2["parity"]["tolerance"] == 1e-3
    for line, cfg in ((c, "cfg5"), (v2, "cfg3")):          # every configuration's roofline carries PMC-derived traffic now
        pj = json.load(open(os.path.join(ROOT, "profiles", f"a5_hbm_traffic_{cfg}.json")))
        assert line["roofline"]["traffic"] == pj["k2_bytes_per_launch"] >= line["roofline"]["algorithmic_bytes"]


def test_bench_line_round5_objects():
    """Round 5: the timed steps rotate over distinct resident batches, the gradient is alive at the end of the timed
    region, the default run carries the other two configurations, and the line says which steps ran on fp16 planes."""
    d = _line()
    assert d["batches"]["distinct"] >= 4
    g = d["gradient_state"]
    assert g["zero_fraction"] < 0.5 and g["max_abs"] > 0 and g["entries"] > 2e7
    f = d["operand_formats"]
    assert f["sites"] > 0 and f["steps_run"] - f["encoder_gemm_steps_on_fp16_planes"] == 2     # the two calibration-free first steps
    assert "fp16" in d["dtype_detail"]
    with open(os.path.join(ROOT, "profiles", "r06_bench_line_default_run.json")) as fh:
        dflt = json.loads(fh.read().strip().splitlines()[-1])
    assert dflt["metric"] == d["metric"] and abs(dflt["value"] - d["value"]) < 0.05 * d["value"]
    o = dflt["other_configs"]
    assert set(o) == {"cfg3", "cfg5"}
    for k, tag in (("cfg3", "configs[2]"), ("cfg5", "configs[4]")):
        assert tag in o[k]["metric"] and o[k]["value"] > 0 and o[k]["roofline"]["frac"] > 0 and o[k]["steps"] >= 10
        assert abs(o[k]["value"] - (80 if k == "cfg3" else 128) / (o[k]["ms_per_step"] * 1e-3)) < 0.01 * o[k]["value"]


def test_bench_self_launch_spawns_before_any_gpu_call(monkeypatch):
    """`python bench.py --gpus N` with no launcher: the parent must start the ranks as fresh children via
    torch.distributed.run BEFORE anything initialises HIP -- only torch.cuda.device_count() is allowed -- relay the launcher's exit
    code, and never exec.  Every GPU-initialising entry point is booby-trapped here."""
    import importlib
    import subprocess
    import sys

    import torch
    bench = importlib.import_module("bench")

    def boom(*a, **k):
        raise AssertionError("the self-launching parent touched the GPU")
    for name in ("is_available", "init", "set_device", "synchronize", "current_device", "get_device_properties"):
        monkeypatch.setattr(torch.cuda, name, boom)
    for name in ("execv", "execve", "execvp", "execl", "execlp"):
        monkeypatch.setattr(os, name, boom)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("LPM_SHARE_GPU", raising=False)
    seen = {}

    def fake_call(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    try:
        bench.main()
        raise AssertionError("main() must exit with the launcher's code")
    except SystemExit as e:
        assert e.code == 7                                  # the worst child rc, as torch.distributed.run reports it
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["MASTER_ADDR"] == "127.0.0.1"
    # fewer GPUs than ranks: non-zero exit with the phase named, nothing spawned
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    seen.clear()
    try:
        bench.main()
        raise AssertionError
    except SystemExit as e:
        assert e.code == 4 and not seen


def test_bench_self_launch_on_this_box_names_its_phase():
    """The real command on a box without (enough) GPUs: exit code 4 and the phase on stderr, no traceback."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LPM_SHARE_GPU")}
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 4 and "self-launch: device count" in r.stderr and "Traceback" not in r.stderr, r.stderr[-2000:]
