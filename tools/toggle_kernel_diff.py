"""From a rocprofv3 --kernel-trace database of tools/dx_toggle_probe.py (steps alternate between the own input-gradient kernel and the
library's GEMM): per kernel occurrence within the step, the mean duration in the steps of either class and the difference.
  python tools/toggle_kernel_diff.py results.db"""
import re, sqlite3, sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else cols[0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "l2_normalize_rows_kernel" in r[0]]
steps = [rows[marks[i]:marks[i + 1]] for i in range(len(marks) - 1)]
steps = steps[len(steps) // 2:]                     # the measured half (the first half is the probe's warm-up)
cls = {0: defaultdict(list), 1: defaultdict(list)}
length = {0: [], 1: []}
for st in steps:
    own = any("proj_dx2_kernel" in r[0] for r in st)
    c = 0 if own else 1
    seen = defaultdict(int)
    for n, a, b in st:
        n = re.sub(r"\(.*", "", n).replace("void ", "")[:70]
        if "proj_dx2" in n or "MT256x80x32" in n:
            n = "<the input-gradient kernel>"
        k = (n, seen[n]); seen[n] += 1
        cls[c][k].append((b - a) / 1e3)
    length[c].append((st[-1][2] - st[0][1]) / 1e3)
print(f"steps: own {len(length[0])}, library {len(length[1])}; first launch to last kernel's end: own {sum(length[0]) / len(length[0]):.1f} us, "
      f"library {sum(length[1]) / len(length[1]):.1f} us")
diff = []
for k in cls[0]:
    if k in cls[1]:
        a, b = sum(cls[0][k]) / len(cls[0][k]), sum(cls[1][k]) / len(cls[1][k])
        diff.append((a - b, a, b, k))
diff.sort()
tot = sum(d[0] for d in diff)
print(f"sum over all kernels of (own - library) = {tot:+.1f} us per step")
for d, a, b, k in diff[:6] + diff[-14:]:
    print(f"  {d:+7.1f} us   own {a:7.1f}  library {b:7.1f}   {k[0]} #{k[1]}")
