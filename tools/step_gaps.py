"""Idle time of the GPU inside one training step: reads a step table written by tools/rocpd_step.py and lists the gaps (no kernel of any
queue running) longer than a threshold, with the launch that ends each gap.  usage: step_gaps.py <step.md> [min gap us, default 8]"""
import re, sys
rows = []
lines = open(sys.argv[1]).read().splitlines()
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
for l in lines:
    m = re.match(r"\| ([\d.]+) \| ([\d.]+) \| (\d+) \| (.*) \|", l)
    if m:
        rows.append((float(m.group(1)), float(m.group(2)), int(m.group(3)), m.group(4)[:70]))
print(lines[2])
end, tot, small = 0.0, 0.0, 0.0
for s, d, q, n in rows:
    gap = s - end
    if gap > 0:
        tot += gap
        if gap <= thr:
            small += gap
    if gap > thr:
        print(f"gap {gap:7.1f} us before {s:8.1f} q{q} {n}")
    end = max(end, s + d)
print(f"idle inside the step: {tot:.1f} us ({small:.1f} us of it in gaps <= {thr} us)")
