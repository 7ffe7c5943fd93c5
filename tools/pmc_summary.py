"""Average the counters of a rocprofv3 --pmc CSV per (kernel name pattern, grid size).  Usage: pmc_summary.py file.csv pattern [...]"""
import csv, collections, sys
acc = collections.defaultdict(list)
for path in sys.argv[1:]:
    if not path.endswith(".csv"):
        continue
    for r in csv.DictReader(open(path)):
        if any(p in r["Kernel_Name"] for p in sys.argv[1:] if not p.endswith(".csv")):
            acc[(r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, g, c), v in sorted(acc.items()):
    print(f"{k} | grid {g} | {c} | n={len(v)} | {sum(v) / len(v):.5g}")
