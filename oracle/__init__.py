"""CPU oracle for the NetVLAD / attention-pooling training-step path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``learnablepoolingmethods_amd/`` may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker / the
timed CPU stand-in, never as the thing shipped.

PARITY UNPINNED: the reference (pomonam/LearnablePoolingMethods) is TF1
graph-mode Python, TensorFlow is not installable in this environment, the
reference holds no tests or golden vectors for this path, and NetVladV1/V2 do
not run as written.  The restatement below follows
the reference line by line (each function cites file:line) with TF1 op
semantics from SURVEY.md App. B and the defect resolutions of App. C, and is
pinned only by analytic known-answer tests, fp64-vs-fp32 agreement and
finite-difference gradient checks (tests/test_oracle_*.py).
"""
from .lpm_oracle import *  # noqa: F401,F403
