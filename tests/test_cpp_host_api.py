"""Builds and runs tests/cpp/test_host_api.cpp: the host-only part of the C++ header API
(layout contract, ranges, Matrix-Market loader, container conversions, generator, reference::spmv
on chesapeake).  No GPU: every container lives in memory_space_t::host."""
import os
import subprocess

from conftest import GOLDEN, ROOT


def test_cpp_host_api():
    exe = os.path.join(ROOT, "build", "test_host_api")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O1", "-x", "hip", "-Wno-unused-result",
           "-I" + os.path.join(ROOT, "include"), src, "-o", exe]
    subprocess.check_call(cmd)
    r = subprocess.run([exe, os.path.join(GOLDEN, "chesapeake.mtx")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " 0 failures" in r.stdout
