"""Build _lib/liblpm_hip_packed.so: the library with the compiler free to emit packed fp32 instructions (what it was before round 3's
determinism finding) -- only for A/B runs (LPM_HIP_LIBRARY=..., tools/packed_ab.sh) and to show that tests/test_gpu_determinism.py
fails on it."""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from learnablepoolingmethods_amd import _build

flags = [f for f in _build.FLAGS if f not in _build.NO_PACKED_FP32]
tmp = tempfile.mkdtemp(prefix="lpm_packed_")
hipcc = _build._hipcc()


def one(src):
    obj = os.path.join(tmp, os.path.basename(src)[:-4] + ".o")
    r = subprocess.run([hipcc, *flags, *_build.EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr)
    return obj


with ThreadPoolExecutor(max_workers=4) as ex:
    objs = list(ex.map(one, _build.sources()))
out = os.path.join(_build.LIBDIR, "liblpm_hip_packed.so")
subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={_build.ARCH}", *objs, "-o", out], check=True)
print(out)
