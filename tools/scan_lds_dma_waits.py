"""Compiler-inserted s_waitcnt vmcnt(N) inside loops that issue LDS-DMA loads (global_load_lds / buffer_load ... lds).
LLVM's memory model treats an LDS-DMA load as a write to LDS that any later LDS read may alias: a plain C++ read of the ring in the same
loop gets an `s_waitcnt vmcnt(0)` in front of it, i.e. the loop waits for EVERY stage in flight, also the one it has just requested --
a ring of any depth degenerates to "request, wait a full memory round trip, compute".  Kernels that keep stages in flight read the ring
with inline-assembly ds_reads behind hand-counted waits; this tool lists the loops where the compiler put its own wait.
  python tools/scan_lds_dma_waits.py learnablepoolingmethods_amd/csrc/proj_gemm.hip [...]       (CPU only: hipcc -S)"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learnablepoolingmethods_amd import _build

DMA = re.compile(r"\b(global_load_lds_\w+|buffer_load_dword\w*\s.*\blds\b)")


def scan(src, smem=False):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        flags = [f for f in _build.FLAGS if f != "-fPIC"]
        r = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "--cuda-device-only", "-S", src, "-o", out], capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(r.stderr[-2000:])
        asm = open(out).read()
    res = []
    for name, body in re.findall(r"^(_Z\w+):\s*;[^\n]*\n(.*?)s_endpgm", asm, flags=re.S | re.M):
        lines = body.splitlines()
        if not any(DMA.search(l) for l in lines):
            continue
        dem = re.sub(r"\(.*", "", subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()).replace("void ", "")
        # loops as the compiler annotates them: a block header line (".LBBx_y:" or "; %bb.N:") carries "Loop Header" / "in Loop: Header=BBx_y"
        loops, cur = {}, None
        for i, l in enumerate(lines):
            if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l):
                m = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
                if m:
                    cur = m.group(1)
                elif "Loop Header" in l:
                    cur = l.split(":")[0].strip().lstrip(".L")
                else:
                    cur = None
            if cur is not None:
                loops.setdefault(cur, []).append((i, l))
        found, scalar = [], []
        for hdr, seg in loops.items():
            if not any(DMA.search(l) for _, l in seg):
                continue
            scalar += [(i, l.strip()) for i, l in seg if re.search(r"\bs_(buffer_)?load_dword", l)]
            inasm = False
            nd, nm = sum(bool(DMA.search(l)) for _, l in seg), sum("v_mfma" in l for _, l in seg)
            for i, l in seg:
                if "ASMSTART" in l: inasm = True
                elif "ASMEND" in l: inasm = False
                elif not inasm:
                    m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
                    if m:
                        found.append((i, int(m.group(1)), nd, nm))
        res.append((dem, sorted(set(found)), sorted(set(scalar))) if smem else (dem, sorted(set(found))))
    return res


if __name__ == "__main__":
    for src in sys.argv[1:]:
        print("==", src)
        for dem, found in scan(src):
            print(f"  {dem}: " + ("clean" if not found else "; ".join(f"line {i}: vmcnt({n}) in a loop of {d} LDS-DMA loads / {m} MFMAs" for i, n, d, m in found[:6])))
