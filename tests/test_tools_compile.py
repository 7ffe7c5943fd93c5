"""-m "not gpu": every script under tools/ at least byte-compiles, and none of them imports the oracle (tools are measurement and
diagnosis helpers of the product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/)."""
import glob
import os
import py_compile
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tools_compile_and_leave_the_oracle_alone():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
    assert len(files) > 10
    for f in files:
        py_compile.compile(f, doraise=True)
        src = open(f).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{os.path.basename(f)} imports the oracle"
        assert not re.search(r"^\s*from\s+tests\b.*\bdp_cases\b", src, re.M), f"{os.path.basename(f)} reaches the oracle through tests.dp_cases"
