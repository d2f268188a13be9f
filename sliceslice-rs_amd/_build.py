"""Builds csrc/ into the in-tree C-ABI library with hipcc for gfx950 (cross-compiles without a GPU).
The kernel families are separate translation units and are compiled in parallel.

Concurrency: every rank of a `torch.distributed.run` job may call build() at once on a fresh clone.  The
whole build runs under an exclusive fcntl lock, objects and the library are written to temporary names
and os.replace()d into place, so no process can ever CDLL a half-written file, and the up-to-date check
compares the library with its objects (a failed link leaves an old .so behind newer .o files)."""
import fcntl
import json
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_ROOT = os.path.dirname(_HERE)
_SO = os.path.join(_CSRC, "libsliceslice_hip.so")
# The default library: what the constructors and ss_searcher_set_filter* can select (26 scan kernels, scan_launch.hpp::kernel_built).
_SOURCES = ["sliceslice_hip.hip", "scan_inst_u4_nt0.hip", "scan_inst_u4_nt1.hip", "scan_inst_find_nt0.hip", "scan_inst_find_nt1.hip"]
# The tuning build (-DSS_TUNING_VARIANTS): every variant ss_searcher_set_variant can name, incl. the U = 8 families.
_TUNING_SOURCES = _SOURCES + ["scan_inst_u8_nt0.hip", "scan_inst_u8_nt1.hip"]
_HEADERS = ["scan_kernels.hpp", "scan_launch.hpp", os.path.join("..", "..", "include", "sliceslice_hip.h")]
_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall"]
# Host-side sanitizer build (the reference's guard on its unsafe code is its ASAN CI job,
# .github/workflows/check.yml:42-58): ASan + UBSan on the HOST code of the same sources, device code untouched.
_SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g", "-O1"]
_NATIVE_BENCH_SRC = os.path.join(_ROOT, "tools", "native_bench.cpp")
_NATIVE_BENCH = os.path.join(_ROOT, "tools", "native_bench")


def library_path():
    return _SO


def native_bench_path():
    return _NATIVE_BENCH


def _mtime(rel):
    return os.path.getmtime(os.path.join(_CSRC, rel))


def _hipcc():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the HIP library cannot be built (and there is no CPU fallback)")
    return hipcc


class _Lock:
    def __init__(self, name):
        self._path = os.path.join(_CSRC, name)

    def __enter__(self):
        self._fh = open(self._path, "w")
        fcntl.flock(self._fh, fcntl.LOCK_EX)
        return self

    def __exit__(self, *a):
        fcntl.flock(self._fh, fcntl.LOCK_UN)
        self._fh.close()
        return False


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _build_variant(so, obj_suffix, extra_flags, link_flags, force, verbose, host_only=False, sources=None):
    """host_only: only the API translation unit (sliceslice_hip.hip - all of the host logic) is compiled with
    `extra_flags`; the kernel-instantiation units come from the regular build (their host side is launch stubs)."""
    sources = sources or _SOURCES
    newest_header = max(_mtime(h) for h in _HEADERS)
    own = sources[:1] if host_only else sources
    objs = [os.path.join(_CSRC, s[:-4] + (obj_suffix if s in own else ".o")) for s in sources]
    todo = []
    for src, obj in zip(sources, objs):
        if src not in own:
            continue
        stale = obj_suffix == ".o" and not os.path.exists(obj + ".res")         # objects from before the resource record
        if force or stale or not os.path.exists(obj) or os.path.getmtime(obj) < max(_mtime(src), newest_header):
            todo.append((src, obj))
    if (not todo and os.path.exists(so) and os.path.getmtime(so) >= max(os.path.getmtime(o) for o in objs) and
            (obj_suffix != ".o" or os.path.exists(_RESOURCES))):
        return so
    hipcc = _hipcc()
    tag = ".tmp%d" % os.getpid()

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + _FLAGS + extra_flags + ["-c", os.path.join(_CSRC, src), "-o", obj + tag]
        if obj_suffix != ".o":
            _run(cmd, verbose)
        else:
            # the regular build also records what the register allocator did with every kernel (see _resources)
            if verbose:
                print(" ".join(cmd), flush=True)
            p = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True)
            rows = _resources(p.stderr, src)
            if p.returncode != 0:
                sys.stderr.write(p.stderr)
                raise subprocess.CalledProcessError(p.returncode, cmd)
            with open(obj + ".res" + tag, "w") as f:
                json.dump(rows, f)
            os.replace(obj + ".res" + tag, obj + ".res")
        os.replace(obj + tag, obj)

    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1) or 1) as pool:
            list(pool.map(compile_one, todo))
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-rpath,/opt/rocm/lib", "-o", so + tag] + link_flags + objs +
         ["-ldl", "-pthread"], verbose)
    os.replace(so + tag, so)
    if obj_suffix == ".o":
        rows = []
        for o in objs:
            if os.path.exists(o + ".res"):
                rows += json.load(open(o + ".res"))
        with open(_RESOURCES + tag, "w") as f:
            json.dump(rows, f, indent=0)
        os.replace(_RESOURCES + tag, _RESOURCES)
    return so


_RESOURCES = os.path.join(_CSRC, "kernel_resources.json")
_RES_KEYS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
             "Occupancy [waves/SIMD]": "waves_per_simd", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills",
             "LDS Size [bytes/block]": "lds_bytes"}


def _resources(remarks, tu):
    """-Rpass-analysis=kernel-resource-usage remarks -> one row per kernel.  Kept next to the library
    (csrc/kernel_resources.json, `kernel_resources()`): the scan kernels are fast at four waves per SIMD (<= 128 VGPRs) and
    14 % slower at three, and which side of 128 the allocator lands on moves with unrelated edits."""
    rows, cur = [], None
    for line in remarks.splitlines():
        m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+): (\S+) \[-Rpass-analysis", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = {"kernel": val, "tu": tu}
            rows.append(cur)
        elif cur is not None and key in _RES_KEYS:
            cur[_RES_KEYS[key]] = int(val)
    if rows:
        names = subprocess.run(["c++filt"] + [r["kernel"] for r in rows], capture_output=True, text=True)
        if names.returncode == 0:
            for r, n in zip(rows, names.stdout.splitlines()):
                r["name"] = n
    return rows


def kernel_resources():
    """Rows of csrc/kernel_resources.json (written by build()): kernel, name, vgprs, waves_per_simd, spills, ..."""
    build()
    return json.load(open(_RESOURCES))


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> csrc/libsliceslice_hip.so.  Returns the path."""
    with _Lock(".build.lock"):
        return _build_variant(_SO, ".o", [], [], force, verbose)


def build_sanitized(force=False, verbose=False):
    """The host logic (sliceslice_hip.hip) with ASan + UBSan -> csrc/libsliceslice_hip_asan.so (test builds only:
    load it with SLICESLICE_HIP_LIB=<path> and the ASan runtime preloaded; see tests/test_gpu_native.py)."""
    so = os.path.join(_CSRC, "libsliceslice_hip_asan.so")
    build(verbose=verbose)                                  # the kernel-instantiation objects are shared with the regular build
    with _Lock(".build_asan.lock"):
        return _build_variant(so, ".asan.o", _SAN_FLAGS, ["-fsanitize=address,undefined", "-shared-libsan"], force, verbose,
                              host_only=True)


_TSAN_FLAGS = ["-fsanitize=thread", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g", "-O1"]


def build_tsan(force=False, verbose=False):
    """ThreadSanitizer build of the host code -> csrc/libsliceslice_hip_tsan.so (tests/test_gpu_native.py)."""
    so = os.path.join(_CSRC, "libsliceslice_hip_tsan.so")
    build(verbose=verbose)
    with _Lock(".build_tsan.lock"):
        return _build_variant(so, ".tsan.o", _TSAN_FLAGS, ["-fsanitize=thread", "-shared-libsan"], force, verbose, host_only=True)


def tsan_runtime():
    out = subprocess.check_output([_hipcc(), "-print-file-name=libclang_rt.tsan-x86_64.so"], text=True).strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def asan_runtime():
    """Path of the clang ASan runtime that build_sanitized() links against (to LD_PRELOAD into python)."""
    out = subprocess.check_output([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def build_native_bench(force=False, verbose=False):
    """tools/native_bench.cpp -> tools/native_bench: the measurements that must not have Python or torch in the
    loop (per-call latencies, the config-1 per-needle loop), linked against the in-tree library."""
    so = build(verbose=verbose)
    with _Lock(".build_tools.lock"):
        if (not force and os.path.exists(_NATIVE_BENCH) and
                os.path.getmtime(_NATIVE_BENCH) >= max(os.path.getmtime(_NATIVE_BENCH_SRC), os.path.getmtime(so))):
            return _NATIVE_BENCH
        tmp = _NATIVE_BENCH + ".tmp%d" % os.getpid()
        _run([_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(_ROOT, "include"), _NATIVE_BENCH_SRC, "-o", tmp,
              "-L", _CSRC, "-lsliceslice_hip", "-Wl,-rpath,$ORIGIN/../sliceslice-rs_amd/csrc", "-Wl,-rpath,/opt/rocm/lib", "-pthread"], verbose)
        os.replace(tmp, _NATIVE_BENCH)
        return _NATIVE_BENCH


def tuning_library_path():
    return os.path.join(_CSRC, "libsliceslice_hip_tuning.so")


def build_tuning(force=False, verbose=False):
    """Every kernel variant ss_searcher_set_variant can name (-DSS_TUNING_VARIANTS, plus the U = 8 translation units) ->
    csrc/libsliceslice_hip_tuning.so.  Not the product: tools/ and the variant tests load it with SLICESLICE_HIP_LIB=<path>."""
    so = tuning_library_path()
    with _Lock(".build_tuning.lock"):
        return _build_variant(so, ".tuning.o", ["-DSS_TUNING_VARIANTS=1"], [], force, verbose, sources=_TUNING_SOURCES)


def build_ab(name, defines, force=False, verbose=False):
    """A/B builds of the SAME sources with extra -D flags -> csrc/libsliceslice_hip_<name>.so (tuning only; load
    with SLICESLICE_HIP_LIB=<path>, see tools/ab_compare.py)."""
    so = os.path.join(_CSRC, "libsliceslice_hip_%s.so" % name)
    with _Lock(".build_ab_%s.lock" % name):
        return _build_variant(so, ".%s.o" % name, ["-D" + d for d in defines], [], force, verbose)
