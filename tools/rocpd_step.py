#!/usr/bin/env python3
"""The kernels of ONE steady-state training step in launch order, from a rocprofv3 rocpd database (--kernel-trace): start offset
within the step, duration, stream (queue), name.  A step = from one lpm::l2_normalize_rows_kernel (a1, the step's first launch) to the
next; the step printed is the `which`-th from the end.  Usage: rocpd_step.py results.db [out.md [which]]"""
import re
import sqlite3
import sys


def main(path, out=None, which=3):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
    rows = cur.execute(f"select {name_col}, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "l2_normalize_rows_kernel" in r[0]]
    if len(marks) < int(which) + 1:
        raise SystemExit("not enough steps in the trace")
    a, b = marks[-int(which) - 1], marks[-int(which)]
    t0 = rows[a][1]
    step = rows[a:b]
    busy = sum(r[2] - r[1] for r in step)
    queues = sorted({r[3] for r in step}) if qcol else []
    lines = [f"# one training step in launch order ({path}; step {which} from the end)", "",
             f"{len(step)} launches, {(rows[b][1] - t0) / 1e3:.1f} us from its first launch to the next step's, kernel time {busy / 1e3:.1f} us"
             + (f", {len(queues)} queues" if qcol else ""), "", "| start us | us | queue | kernel |", "|---:|---:|---:|---|"]
    for r in step:
        n = r[0].replace("(anonymous namespace)::", "")
        n = re.sub(r"\(.*", "", n).replace("void ", "")[:120]
        q = queues.index(r[3]) if qcol else 0
        lines.append(f"| {(r[1] - t0) / 1e3:.1f} | {(r[2] - r[1]) / 1e3:.1f} | {q} | `{n}` |")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:4])
