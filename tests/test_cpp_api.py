"""The source-compatible C++ boundary (include/cudf/*.hpp + libcudf_amd.so) used directly from C++, the way a
libcudf user or a Cython .pxd does (tests/cpp/api_compat.cpp). CPU: it must compile and link; GPU: it must run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "api_compat.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "api_compat.bin")


def _build():
    lib_dir = os.path.join(ROOT, "cudf_amd", "lib")
    assert os.path.exists(os.path.join(lib_dir, "libcudf_amd.so")), "build the library first (__graft_entry__.build())"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++20", "-O1", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                           SRC, "-o", EXE, "-L", lib_dir, "-lcudf_amd", f"-Wl,-rpath,{lib_dir}"])


def test_cpp_api_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_api_runs():
    if not os.path.exists(EXE):
        _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "api_compat OK" in out.stdout
