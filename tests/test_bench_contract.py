"""The committed bench line (profiles/r01_bench_line.json, written by bench.py on the GPU box) keeps the driver's contract:
required keys, BASELINE.json's metric, and internally consistent roofline / throughput figures.  CPU-only: it reads the
committed artefact, it does not run the bench."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r01_bench_line.json")) as f:
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
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    a = d["assign_gemm"]
    assert abs(a["mfma_util_vs_bf16_peak"] - a["executed_bf16_tflops"] / 2500.0) < 5e-3
