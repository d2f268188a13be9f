"""Compiles and runs the C++ veneer example against the in-tree library (GPU box only)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_veneer(tmp_path):
    import sliceslice_rs_amd as ss
    so = ss.build()
    exe = str(tmp_path / "veneer_test")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "veneer_test.cpp"), "-o", exe,
                           "-L", os.path.dirname(so), "-lsliceslice_hip", "-Wl,-rpath," + os.path.dirname(so)])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "veneer_test ok" in out.stdout
