"""Builds csrc/ into the in-tree C-ABI library with hipcc for gfx950 (cross-compiles without a GPU).
The kernel families are separate translation units and are compiled in parallel."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.path.join(_CSRC, "libsliceslice_hip.so")
_SOURCES = ["sliceslice_hip.hip", "scan_inst_u4.hip", "scan_inst_u8.hip", "scan_inst_find.hip"]
_HEADERS = ["scan_kernels.hpp", "scan_launch.hpp", os.path.join("..", "..", "include", "sliceslice_hip.h")]
_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall"]


def library_path():
    return _SO


def _mtime(rel):
    return os.path.getmtime(os.path.join(_CSRC, rel))


def _hipcc():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the HIP library cannot be built (and there is no CPU fallback)")
    return hipcc


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> csrc/libsliceslice_hip.so.  Returns the path."""
    newest_header = max(_mtime(h) for h in _HEADERS)
    todo = []
    for src in _SOURCES:
        obj = os.path.join(_CSRC, src[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(_mtime(src), newest_header):
            todo.append((src, obj))
    if not todo and os.path.exists(_SO) and not force:
        return _SO
    hipcc = _hipcc()

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + _FLAGS + ["-c", os.path.join(_CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1) or 1) as pool:
        list(pool.map(compile_one, todo))
    objs = [os.path.join(_CSRC, s[:-4] + ".o") for s in _SOURCES]
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-rpath,/opt/rocm/lib", "-o", _SO] + objs + ["-ldl", "-pthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return _SO
