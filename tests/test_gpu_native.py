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


def test_c_grep_example(tmp_path):
    """tools/grep_hip.c: examples/grep.rs in plain C (gcc, no HIP headers needed by the caller)."""
    import sliceslice_rs_amd as ss
    so = ss.build()
    exe = str(tmp_path / "grep_hip")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "grep_hip.c"),
                           "-o", exe, "-L", os.path.dirname(so), "-lsliceslice_hip", "-Wl,-rpath," + os.path.dirname(so)])
    data = os.path.join(ROOT, "tests", "golden", "data", "i386.txt")
    out = subprocess.run([exe, "privilege", data], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("true"), out.stdout + out.stderr
    out = subprocess.run([exe, "no such phrase in the manual", data], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("false"), out.stdout + out.stderr
