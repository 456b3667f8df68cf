"""tests/cpp/test_device_api.cpp: every algorithms::spmv wrapper of the C++ header API on a matrix
battery (f32 + f64) vs reference::spmv, plan reuse, device conversions, SpMM.  The binary is built
wherever hipcc is (no GPU needed to compile) and run only on a GPU box."""
import os
import subprocess

import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "build", "test_device_api")
SRC = os.path.join(ROOT, "tests", "cpp", "test_device_api.cpp")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    deps = [SRC] + [os.path.join(b, f) for b, _, fs in os.walk(os.path.join(ROOT, "include")) for f in fs]
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(d) for d in deps):
        return  # up to date with the test source AND every header it can include
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O2", "-x", "hip",
                           "-Wno-unused-result", "-I" + os.path.join(ROOT, "include"), SRC, "-o", EXE])


def test_cpp_device_api_compiles():
    build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_device_api_runs():
    build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " 0 failures" in r.stdout
