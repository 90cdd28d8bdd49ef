"""The C-ABI library loads on a CPU-only box, exports every symbol include/gretel_hip.h
declares, and fails LOUDLY (never silently falls back) when there is no GPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from gretel_amd import _lib


def _declared():
    src = open(os.path.join(ROOT, "include", "gretel_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gh_[a-z_0-9A-Z]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    names = _declared()
    assert len(names) >= 30
    so = ctypes.CDLL(_lib.SO_PATH)
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, missing


def test_binding_covers_the_header():
    L = _lib.load()
    for n in _declared():
        assert getattr(L, n).argtypes is not None or n == "gh_last_error", n


def test_no_silent_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gretel_amd.hansel import Hansel
    with pytest.raises(_lib.GretelHipError):
        Hansel(10, band=2)
    h = Hansel(10)                      # lazily staged: no device touched yet
    h.add_observation('A', 'C', 1, 2)
    with pytest.raises(_lib.GretelHipError):
        h.get_observation('A', 'C', 1, 2)


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "gretel_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt.replace("oracle/hansel_ref.py", "").replace("oracle/gretel_ref.py", ""):
                    bad.append(f)
    assert not bad, bad
    # ... nor do the measurement scripts (scratch/, profiles/): whatever calls the oracle lives under tests/; bench.py only in its
    # cpu_baseline legs, __graft_entry__.py only in smoke()
    for d in ("scratch", "profiles"):
        for f in os.listdir(os.path.join(ROOT, d)):
            if f.endswith((".py", ".sh")):
                txt = open(os.path.join(ROOT, d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                    bad.append(d + "/" + f)
    assert not bad, bad
    b = open(os.path.join(ROOT, "bench.py")).read()
    for m in re.finditer(r"^(\s*)(from|import)\s+oracle\b", b, flags=re.M):
        fn = re.findall(r"^def (\w+)", b[:m.start()], flags=re.M)[-1]
        assert fn.startswith("cpu_baseline"), fn
    g = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    for m in re.finditer(r"^(\s*)(from|import)\s+oracle\b", g, flags=re.M):
        assert re.findall(r"^def (\w+)", g[:m.start()], flags=re.M)[-1] == "smoke"


def test_io_library_exports_its_header():
    from gretel_amd import bamio
    src = open(os.path.join(ROOT, "include", "gretel_io.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(gio_[a-z_0-9]+)\s*\(", src)))
    assert len(names) >= 4
    so = ctypes.CDLL(bamio.IO_SO)
    assert not [n for n in names if not hasattr(so, n)]
