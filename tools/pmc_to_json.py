"""pass1.csv (FETCH_SIZE) + pass2.csv (WRITE_SIZE) of tools/pmc_a5.sh -> the HBM traffic of every kernel of the a5 chain per launch, as
JSON on stdout.  Corrections exactly as MI355X_MICROARCH.md prescribes for gfx950: the counters are in KB (1024 bytes); FETCH_SIZE
reports half of wide coalesced reads -> x 2; WRITE_SIZE as reported.  bench.py reads the result (profiles/a5_hbm_traffic_<cfg>.json)
for `roofline.traffic`.   usage: pmc_to_json.py <dir with pass1.csv, pass2.csv> <cfg2|cfg3|cfg5> <commit>"""
import collections, csv, json, os, sys

d, cfg, commit = sys.argv[1], sys.argv[2], sys.argv[3]
ALG = {"cfg2": 4 * (80 * 300 * 256 + 80 * 300 * 1024 + 80 * 1024 * 256) + 4 * 1024 * 256,
       "cfg3": 4 * (80 * 300 * 256 + 80 * 300 * 1024 + 80 * 1024 * 256) + 4 * 1024 * 256,
       "cfg5": 2 * (128 * 300 * 512 + 128 * 300 * 1024 + 128 * 1024 * 512) + 4 * 1024 * 512}[cfg]
CHAIN = ("assign_tiles", "softmax_stats", "vlad_aggregate", "vlad_kmajor", "vlad_clip", "vlad_finalize", "vlad_row_scales")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    for r in csv.DictReader(open(os.path.join(d, f"pass{i}.csv"))):
        acc[(r["kernel"], r["grid"])][r["counter"]].append(float(r["value"]))
kern, total, k2 = {}, 0, None
for (k, grid), c in sorted(acc.items()):
    if not any(p in k for p in CHAIN):
        continue
    rd = 2 * 1024 * sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [0])), 1)
    wr = 1024 * sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [0])), 1)
    kern[f"{k} (grid {grid})"] = {"read_bytes": int(rd), "write_bytes": int(wr), "bytes": int(rd + wr), "launches_averaged": len(c.get("FETCH_SIZE", []))}
    total += rd + wr
    if ("vlad_aggregate" in k or "vlad_kmajor" in k or "vlad_clip" in k) and (k2 is None or rd + wr > k2[1]):
        k2 = (k, rd + wr)
print(json.dumps({"config": cfg, "commit": commit, "algorithmic_bytes": ALG,
                  "correction": "MI355X_MICROARCH.md HBM section: counters in KB (1024 B); gfx950 FETCH_SIZE x 2; WRITE_SIZE as reported",
                  "kernels": kern, "k2_kernel": k2[0] if k2 else None, "k2_bytes_per_launch": int(k2[1]) if k2 else None,
                  "chain_bytes_per_launch": int(total), "chain_vs_algorithmic": round(total / ALG, 3),
                  "k2_vs_algorithmic": round(k2[1] / ALG, 3) if k2 else None,
                  "source": "separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of tools/run_k2_only.py 6 " + cfg + " (tools/pmc_a5.sh)"}, indent=1))
