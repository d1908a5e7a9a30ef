"""CPU: the C-ABI library loads and exports every symbol include/cudf_amd_c.h declares (no compute calls)."""
import os
import re

from cudf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "cudf_amd_c.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cudf_amd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"libcudf_amd.so does not export {n}"
    # and the ctypes prototypes cover exactly the declared set
    assert sorted(_lib.SYMBOLS) == names


def test_version_and_error_string():
    lib = _lib.load()
    assert b"gfx950" in lib.cudf_amd_version()
    assert lib.cudf_amd_last_error() is not None


def test_product_does_not_reference_oracle():
    """The product path must not import, link or call anything under oracle/."""
    for base, _, files in os.walk(os.path.join(ROOT, "cudf_amd")):
        if os.sep + "build" in base:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), f"{os.path.join(base, f)} mentions the oracle"
