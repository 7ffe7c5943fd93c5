#!/usr/bin/env python3
"""Prints a table of per-kernel VGPR/AGPR/spill/LDS/occupancy from hipcc's
-Rpass-analysis=kernel-resource-usage for the given .hip files (CPU-only, no GPU needed)."""
import re
import subprocess
import sys

KEYS = ["VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"]


def main(paths):
    for p in paths:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950",
                            "-Rpass-analysis=kernel-resource-usage", "-c", p, "-o", "/dev/null"],
                           capture_output=True, text=True)
        cur = None
        rows = {}
        for line in r.stderr.splitlines():
            m = re.search(r"remark: (?:\s*)([^:\[]+): (\S+)", line)
            if not m:
                continue
            k, v = m.group(1).strip(), m.group(2)
            if k == "Function Name":
                cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
                cur = re.sub(r"\(.*", "", cur).replace("void lpm::", "")
                rows[cur] = {}
            elif cur:
                rows[cur][k] = v
        print(f"== {p}")
        print(f"{'kernel':60s} vgpr agpr scratch spill sgpr   lds occ")
        for name, d in rows.items():
            print(f"{name[:60]:60s} {d.get('VGPRs','?'):>4s} {d.get('AGPRs','?'):>4s} {d.get('ScratchSize [bytes/lane]','?'):>7s} "
                  f"{d.get('VGPRs Spill','?'):>5s} {d.get('SGPRs','?'):>4s} {d.get('LDS Size [bytes/block]','?'):>6s} {d.get('Occupancy [waves/SIMD]','?'):>3s}")


if __name__ == "__main__":
    main(sys.argv[1:])
