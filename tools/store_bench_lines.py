"""Splits the output of `python bench.py --config all` (three JSON lines) into profiles/r06_bench_line{,_cfg3,_cfg5}.json.
usage: python tools/store_bench_lines.py gpurun_out/<tag>/bench_all.jsonl [prefix, default profiles/r06_bench_line]"""
import json, sys
prefix = sys.argv[2] if len(sys.argv) > 2 else "profiles/r06_bench_line"
for l in open(sys.argv[1]):
    l = l.strip()
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    w = d["config"]["workload"]
    suffix = "_cfg3" if "NetVladV2" in w else ("_cfg5" if "gated NetVLAD" in w else "")
    open(f"{prefix}{suffix}.json", "w").write(l + "\n")
    print(f"{prefix}{suffix}.json", d["value"], d["ms_per_step"], d["roofline"]["frac"])
