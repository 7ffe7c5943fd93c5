"""-m "not gpu": the C-ABI library loads on a CPU-only box and exports every symbol include/lpm_hip.h
declares, the ctypes table in _capi.py covers exactly those symbols, and the product path refuses CPU
tensors (no silent fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lpm_hip.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lpm_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for must in ("lpm_assign_gemm_fwd", "lpm_vlad_aggregate_fwd", "lpm_vlad_aggregate_bwd", "lpm_mha_fwd", "lpm_mha_bwd",
                 "lpm_multi_tensor_clip_adam", "lpm_frame_apply", "lpm_version", "lpm_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from learnablepoolingmethods_amd import _build, _capi
    if not os.path.exists(_capi.LIB_PATH):
        _build.build(verbose=False)
    dll = ctypes.CDLL(_capi.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(dll, n)]
    assert not missing, f"declared in lpm_hip.h but not exported: {missing}"


def test_ctypes_table_matches_header():
    from learnablepoolingmethods_amd import _capi
    assert sorted(_capi.SIGNATURES) == _declared()
    txt = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in _capi.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", txt, flags=re.S)
        assert m, name
        params = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        assert len(params) == len(args), f"{name}: header has {len(params)} parameters, ctypes table {len(args)}"


def test_version_and_error_string():
    from learnablepoolingmethods_amd import _capi
    lib = _capi.load()
    assert lib.version() == 100
    assert isinstance(lib.last_error(), str)


def test_product_path_refuses_cpu_tensors():
    from learnablepoolingmethods_amd import _capi, ops, registry
    from learnablepoolingmethods_amd import variables as vs
    with pytest.raises(_capi.LpmError):
        ops.netvlad(torch.zeros(8, 128), torch.zeros(128, 8), None, 4, bn=None, bias=torch.zeros(8))
    with pytest.raises(_capi.LpmError):
        ops.mha_core(torch.zeros(1, 16, 16), torch.zeros(1, 16, 16), torch.zeros(1, 16, 16), 1, 1.0)
    store = vs.VariableStore(device="cpu")
    with vs.use_store(store), pytest.raises(_capi.LpmError):
        registry.get_model("NetVladV1").create_model(torch.zeros(2, 10, 1152), vocab_size=10,
                                                     num_frames=torch.tensor([10, 10]), iterations=5,
                                                     cluster_size=8, hidden_size=8)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "learnablepoolingmethods_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"
