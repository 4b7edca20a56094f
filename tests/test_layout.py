"""CPU-side structural checks: the C-ABI library loads and exports every symbol include/melgpt.h declares
(no compute calls - there is no GPU here), and the product package never touches the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "melgpt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(melgpt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from melspec_gpt_vqvae_amd import _ffi, build

    lib_path = build.build()
    assert os.path.exists(lib_path)
    L = ctypes.CDLL(lib_path)
    syms = _declared_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/melgpt.h but not exported"
    assert _ffi.lib().melgpt_abi_version() == 1
    # the fp16 flavour (same sources, -DMELGPT_HALF_FP16) exports the same ABI
    L16 = ctypes.CDLL(build.lib_path("fp16"))
    assert all(hasattr(L16, s) for s in syms) and L16.melgpt_abi_version() == 1
    # every bound prototype is declared in the header and vice versa
    assert sorted(_ffi._PROTOS) == syms
    # ... and the libraries export NO melgpt_ symbol the header does not declare (dynamic symbol table, `nm -D`)
    import shutil
    import subprocess

    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    for flavour in ("bf16", "fp16"):
        out = subprocess.run([nm, "-D", "--defined-only", build.lib_path(flavour)], capture_output=True, text=True, check=True).stdout
        exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("melgpt_")})
        assert exported == syms, (flavour, sorted(set(exported) ^ set(syms)))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "melspec_gpt_vqvae_amd")
    bad = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt and f.endswith(".py"):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch):
    from melspec_gpt_vqvae_amd import _ffi

    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", "/nonexistent/libmelgpt_hip.so")
    with pytest.raises(_ffi.MelgptError):
        _ffi.lib()
