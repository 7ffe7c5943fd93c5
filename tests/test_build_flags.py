"""The library is built WITHOUT packed fp32 instructions (learnablepoolingmethods_amd/_build.py, NO_PACKED_FP32): with them, the low
half of a compiler-generated v_pk_fma_f32 chain was wrong by a few per cent in ~3 % of training steps when a second process shared
the GPU (tools/determinism_check.py).  hipcc cross-compiles gfx950 here: the device assembly of one source file
under the build's flags must hold no v_pk_{fma,mul,add}_f32 -- and the same file without the flag must, or this test sees nothing."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from learnablepoolingmethods_amd import _build

PACKED = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")
SRC = os.path.join(_build.CSRC, "clip_adam.hip")


def _device_asm(flags, tmp_path, name):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / name
    dev = [f for f in flags if f not in ("-fPIC",)]
    r = subprocess.run([hipcc, *dev, "--cuda-device-only", "-S", SRC, "-o", str(out)], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


@pytest.mark.timeout(600)
def test_library_flags_produce_no_packed_fp32_instructions(tmp_path):
    assert _build.NO_PACKED_FP32 and all(f in _build.FLAGS for f in _build.NO_PACKED_FP32)
    asm = _device_asm(_build.FLAGS, tmp_path, "with_flag.s")
    assert "s_endpgm" in asm and not PACKED.search(asm), "packed fp32 instructions in the device code of the library build"
    without = [f for f in _build.FLAGS if f not in _build.NO_PACKED_FP32]
    # (-Xclang appears twice in NO_PACKED_FP32; the filter above drops every -Xclang, which is what is wanted here)
    asm2 = _device_asm(without, tmp_path, "without_flag.s")
    assert PACKED.search(asm2), "the control compile holds no packed fp32 instruction either: this test cannot see the flag's effect"


@pytest.mark.timeout(600)
def test_clip_wide_aggregation_loop_carries_its_fragments_in_place(tmp_path):
    """vlad_clip.hip requests the NEXT step's fragments with inline-assembly ds_reads before the loop's back edge and waits for them at
    the top of the next iteration (hand-counted lgkmcnt).  That is only correct while the compiler keeps every such fragment in the SAME
    registers at both ends of the back edge: a register copy behind the request reads bits that have not arrived yet (the round-4
    determinism failure: a v_mov of the next assignment fragment, stale whenever LDS was slower than the copy).  The main loop of every
    production instantiation must hold no vector register copy and no scratch access."""
    global SRC
    keep = SRC
    try:
        SRC = os.path.join(_build.CSRC, "vlad_clip.hip")
        asm = _device_asm(_build.FLAGS, tmp_path, "vlad_clip.s")
    finally:
        SRC = keep
    kernels = re.findall(r"^(_ZN3lpm16vlad_clip_kernelILi\dELi\dELi0EEEvNS_6VCArgsE):[^\n]*\n(.*?)s_endpgm", asm, flags=re.S | re.M)
    assert len(kernels) >= 2, "production instantiations of vlad_clip_kernel not found in the device assembly"
    for name, body in kernels:
        lines = body.splitlines()
        head = [i for i, l in enumerate(lines) if "Inner Loop Header" in l]
        assert head, name
        label = lines[head[0]].split(":")[0].strip()
        back = [i for i, l in enumerate(lines) if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\s*$", l) and i > head[0]]
        assert back, f"{name}: no back edge to {label}"
        loop = [l for l in lines[head[0]:back[-1] + 1] if not l.strip().startswith(";")]
        assert sum("v_mfma_f32_32x32x16_bf16" in l for l in loop) == 33, f"{name}: expected the 33 MFMAs of one step in the loop"
        bad = [l.strip() for l in loop if re.search(r"\b(v_mov_b(32|64)|v_accvgpr_(read|write)\w*|scratch_(load|store)\w*)\b", l)]
        assert not bad, f"{name}: register copies / scratch traffic inside the main loop: {bad[:6]}"


@pytest.mark.timeout(600)
def test_bf16_clip_wide_aggregation_loop_carries_its_fragments_in_place(tmp_path):
    """vlad_clip16.hip (round 6: K2 for bf16 storage) uses vlad_clip.hip's technique -- fragments requested by inline-assembly ds_reads
    across the loop's back edge, waited for with hand-counted lgkmcnt -- over two rotating B-pair register sets and two alternating A sets
    (the loop holds two steps).  The same condition holds: in the main loop of the production instantiations no vector register copy and
    no scratch access, the 24 MFMAs of two steps, and no s_waitcnt vmcnt(0) other than the ones the source places at the end of the
    frame loop (the compiler must not drain the LDS-DMA ring on its own)."""
    global SRC
    keep = SRC
    try:
        SRC = os.path.join(_build.CSRC, "vlad_clip16.hip")
        asm = _device_asm(_build.FLAGS, tmp_path, "vlad_clip16.s")
    finally:
        SRC = keep
    kernels = re.findall(r"^(_ZN3lpm18vlad_clip16_kernelILi\dELi0ELi\dEEEvNS_6VBArgsE):[^\n]*\n(.*?)s_endpgm", asm, flags=re.S | re.M)
    assert len(kernels) >= 2, "production instantiations of vlad_clip16_kernel not found in the device assembly"
    for name, body in kernels:
        assert "scratch_" not in body, f"{name}: scratch traffic (spilled registers)"
        lines = body.splitlines()
        head = [i for i, l in enumerate(lines) if "Inner Loop Header" in l]
        assert head, name
        label = lines[head[0]].split(":")[0].strip()
        back = [i for i, l in enumerate(lines) if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\s*$", l) and i > head[0]]
        assert back, f"{name}: no back edge to {label}"
        loop = [l for l in lines[head[0]:back[-1] + 1] if not l.strip().startswith(";")]
        assert sum("v_mfma_f32_32x32x16_bf16" in l for l in loop) == 24, f"{name}: expected the 24 MFMAs of two steps in the loop"
        assert sum("ds_read_b128" in l for l in loop) == 16, f"{name}: expected the 16 fragment reads of two steps in the loop"
        bad = [l.strip() for l in loop if re.search(r"\b(v_mov_b(32|64)|v_accvgpr_(read|write)\w*|scratch_(load|store)\w*)\b", l)]
        assert not bad, f"{name}: register copies / scratch traffic inside the main loop: {bad[:6]}"


@pytest.mark.timeout(900)
def test_projection_kernels_keep_their_lds_dma_rings_in_flight():
    """To LLVM an LDS-DMA load (global_load_lds) is a store to LDS that any later LDS read may alias: a plain C++ read of such a ring gets an
    `s_waitcnt vmcnt(0)` in front of it and the loop waits for every stage in flight -- also the one it has just requested.  The projection
    kernels ran that way for three rounds (cfg-2 forward 144 us instead of 107, cfg-5 input gradient 634 us instead of 455).  They now read
    their rings with inline-assembly ds_reads behind counted waits; this test keeps the compiler's own vmcnt waits out of every loop of
    proj_gemm.hip that issues LDS-DMA loads, and scalar loads (which share lgkmcnt with the ds_reads and return out of order) out of them."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(_build.CSRC), "..", "tools"))
    try:
        import scan_lds_dma_waits as scan
    finally:
        sys.path.pop(0)
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    rep = scan.scan(os.path.join(_build.CSRC, "proj_gemm.hip"), smem=True)
    names = [n for n, _, _ in rep]
    assert sum("proj_fwd_kernel" in n for n in names) >= 4 and sum("proj_dx2_kernel" in n for n in names) == 4, names
    for name, waits, smem in rep:
        assert not waits, f"{name}: compiler-inserted vmcnt waits inside an LDS-DMA loop: {waits[:4]}"
        assert not smem, f"{name}: scalar loads inside an LDS-DMA loop (they share lgkmcnt with the counted ds_reads): {smem[:4]}"


@pytest.mark.timeout(900)
def test_flat_k1_loop_keeps_loads_in_flight_and_copies_no_registers():
    """assign_flat.hip hands registers that asynchronous loads fill (global_load_dwordx4 / ds_read_b128 issued by inline assembly) to the
    MFMAs behind hand-counted waits: the compiler must not add waits of its own to the loop, and it must not COPY such a register between
    its load and its wait -- a copy placed in front of the wait reads what the load has not written yet (seen once while the kernel was
    written: a tied operand in both arms of a branch made it do exactly that)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(_build.CSRC), "..", "tools"))
    try:
        import scan_lds_dma_waits as scan
    finally:
        sys.path.pop(0)
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    src = os.path.join(_build.CSRC, "assign_flat.hip")
    rep = scan.scan(src)
    assert sum("assign_flat_kernel" in n for n, _ in rep) >= 3
    for name, waits in rep:
        if re.search(r"assign_wide_kernel<[1-9]\d*>", name):
            continue                                            # timing-experiment instantiations (-DLPM_K1_WIDE_EXPERIMENTS builds only)
        assert not waits, f"{name}: compiler-inserted vmcnt waits inside the LDS-DMA loop: {waits[:4]}"
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        flags = [f for f in _build.FLAGS if f != "-fPIC"]
        r = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "--cuda-device-only", "-S", src, "-o", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        asm = open(out).read()
    kernels = re.findall(r"^(_ZN3lpm18assign_flat_kernelILi\dELi\dELb\dELb\dEEEvNS_14AssignFlatArgsE):", asm, flags=re.M)
    assert len(kernels) >= 4 and any(k.endswith("Lb1EEEvNS_14AssignFlatArgsE") for k in kernels), kernels      # (incl. the plain-bf16 form of round 5)
    for mangled in kernels:
        plain = mangled.endswith("Lb1EEEvNS_14AssignFlatArgsE")
        # plain: the 6 MFMAs of each of the 4 unrolled double steps; split: the 9 MFMAs of each of the 4 or 8 unrolled steps
        _check_async_loop(asm, mangled, (24,) if plain else (36, 72))
    # the 160-row x 512-column plain form (round 5): the 10 MFMAs of each of the 8 unrolled plain steps of a stage
    wide = re.findall(r"^(_ZN3lpm18assign_wide_kernelILi0EEEvNS_14AssignFlatArgsE):", asm, flags=re.M)
    assert len(wide) == 1, wide
    body = _check_async_loop(asm, wide[0], (80,))
    m = re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", asm[asm.index(wide[0] + ":"):])
    assert m and int(m.group(1)) <= 256, "two waves per SIMD need <= 256 registers each"
    loop = body["loop"]
    assert sum("ds_read_b128" in l for l in loop) == 40 and sum("global_load_dwordx4" in l for l in loop) == 16
    assert sum("global_load_lds_dwordx4" in l for l in loop) == 5 and sum("s_barrier" in l for l in loop) == 1
    # a fragment read never lands in a register that one of the wave's last four MFMAs (two pairs) read as an operand: with two waves
    # queueing on a SIMD's matrix core the MFMA in front of the read may not have started when the LDS answers (seen: wrong column
    # tiles once the timing shifted); the kernel rotates eight fragment registers and keeps old values alive to force the allocator
    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()
    recent = []
    for l in loop:
        m = re.match(r"\s*v_mfma\w+\s+(\S+),\s*(\S+),\s*(\S+),", l)
        if m:
            recent = (recent + [regs(m.group(2)) | regs(m.group(3))])[-4:]
            continue
        m = re.match(r"\s*ds_read_b128\s+(v\[\d+:\d+\])", l)
        if m:
            assert not any(regs(m.group(1)) & r for r in recent), f"fragment read into a register an MFMA just in front of it reads: {l.strip()}"


def _check_async_loop(asm, mangled, mfma_counts):
    """The main loop of `mangled` holds one of `mfma_counts` MFMAs, no register copies and no scratch traffic; between the kernel's first
    asynchronous load and its last MFMA no register such a load writes is the source of a copy."""
    body = asm[asm.index(mangled + ":"):]
    body = body[:body.index("s_endpgm")]
    lines = body.splitlines()
    head = [i for i, l in enumerate(lines) if "Loop Header" in l]
    assert head, f"{mangled}: no loop"
    label = lines[head[0]].split(":")[0].strip()
    back = [i for i, l in enumerate(lines) if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\s*$", l) and i > head[0]]
    assert back, f"{mangled}: no back edge to {label}"
    loop = [l for l in lines[head[0]:back[-1] + 1] if not l.strip().startswith(";")]
    n = sum("v_mfma_f32_32x32x16_bf16" in l for l in loop)
    assert n in mfma_counts, f"{mangled}: expected {mfma_counts} MFMAs in the loop, found {n}"
    bad = [l.strip() for l in loop if re.search(r"\b(v_mov_b(32|64)|v_accvgpr_(read|write)\w*|scratch_(load|store)\w*)\b", l)]
    assert not bad, f"{mangled}: register copies / scratch traffic inside the main loop: {bad[:6]}"

    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)", tok)
        return {int(m.group(1))} if m else set()
    first = min(i for i, l in enumerate(lines) if re.match(r"\s*(global_load_dwordx4|ds_read_b128|global_load_lds)", l))
    last = max(i for i, l in enumerate(lines) if "v_mfma" in l)
    # a register is "asynchronous" from the load that writes it up to the next MFMA that reads it (the hand-counted wait sits in
    # between); the compiler renames freely, so the set is tracked along the instruction stream
    pending, copies, seen = set(), [], set()
    for l in lines[first:last + 1]:
        m = re.match(r"\s*(global_load_dwordx4|ds_read_b128)\s+(v\[\d+:\d+\])", l)
        if m:
            pending |= regs(m.group(2))
            seen |= regs(m.group(2))
            continue
        m = re.match(r"\s*v_mfma\w+\s+(\S+),\s*(\S+),\s*(\S+),", l)
        if m:
            pending -= regs(m.group(2)) | regs(m.group(3))
            continue
        m = re.match(r"\s*v_mov_b(32|64)\w*\s+(\S+),\s*(\S+)", l)
        if m and (regs(m.group(3)) & pending):
            copies.append(l.strip())
    assert len(seen) >= 48, f"{mangled}: the scan found only {len(seen)} registers written by asynchronous loads"
    assert not copies, f"{mangled}: copies of registers that asynchronous loads write: {copies[:6]}"
    return {"loop": loop, "lines": lines}
