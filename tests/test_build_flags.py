"""The library is built WITHOUT packed fp32 instructions (learnablepoolingmethods_amd/_build.py, NO_PACKED_FP32): with them, the low
half of a compiler-generated v_pk_fma_f32 chain was wrong by a few per cent in ~3 % of training steps when a second process shared
the GPU (tools/determinism_check.py).  hipcc cross-compiles gfx950 here: the device assembly of one source file
under the build's flags must hold no v_pk_{fma,mul,add}_f32 -- and the same file without the flag must, or this test sees nothing."""
import os
import re
import shutil
import subprocess

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
