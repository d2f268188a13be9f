"""Builds csrc/ into the in-tree C-ABI library with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.path.join(_CSRC, "libsliceslice_hip.so")
_SOURCES = ["sliceslice_hip.hip"]
_DEPS = ["sliceslice_hip.hip", "scan_kernels.hpp", os.path.join("..", "..", "include", "sliceslice_hip.h")]


def library_path():
    return _SO


def _stale():
    if not os.path.exists(_SO):
        return True
    t = os.path.getmtime(_SO)
    return any(os.path.getmtime(os.path.join(_CSRC, d)) > t for d in _DEPS)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> csrc/libsliceslice_hip.so.  Returns the path."""
    if not force and not _stale():
        return _SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the HIP library cannot be built (and there is no CPU fallback)")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall",
           "-Wl,-rpath,/opt/rocm/lib", "-o", _SO] + [os.path.join(_CSRC, s) for s in _SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return _SO
