#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) into a per-kernel table
(name, calls, total ms, avg us, % of GPU time) -- EVERY kernel, no cut.  Usage: rocpd_stats.py results.db [out.md [steps]]
steps: the number of training steps the trace covers (warm-up + spin-up + timed), to print dispatches per step."""
import re
import sqlite3
import sys


def main(path, out=None, steps=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows)
    nd = sum(r[1] for r in rows)
    per = f" = {nd / int(steps):.0f} dispatches per step over {steps} steps (incl. the first, variable-creating ones)" if steps else ""
    lines = [f"# rocprofv3 --kernel-trace summary of {path}", "", f"total kernel time {total/1e6:.3f} ms over {nd} dispatches{per}; {len(rows)} distinct kernels, all listed", "",
             "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for n, c, s, a, mn, mx in rows:
        n = n.replace("(anonymous namespace)::", "")           # (otherwise at::native::(anonymous namespace)::X is cut at its first parenthesis)
        n = re.sub(r"\(.*", "", n)
        n = n.replace("void ", "")[:110]
        lines.append(f"| `{n}` | {c} | {s/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*s/total:.1f} |")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:4])
