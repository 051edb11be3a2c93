"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/keisei_hip.h declares (no compute without a GPU); the product refuses to run without it."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    from shogidrl_amd import _lib, build
    build.build()
    hdr = open(os.path.join(ROOT, "include", "keisei_hip.h")).read()
    declared = set(re.findall(r"\b(ksh_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert L.ksh_version() >= 100


def test_product_has_no_cpu_path():
    import torch
    from shogidrl_amd import KeiseiHipError, ops
    from shogidrl_amd.env import VecShogiEnv
    with pytest.raises(KeiseiHipError):
        VecShogiEnv(4, device="cpu")
    with pytest.raises(KeiseiHipError):
        ops.gae(torch.zeros(2, 2), torch.zeros(2, 2), torch.zeros(2, 2, dtype=torch.bool), torch.zeros(2), 0.99, 0.95)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)|oracle/|libkeisei_oracle", re.M)
    pkg = os.path.join(ROOT, "shogidrl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), (dirpath, f)
