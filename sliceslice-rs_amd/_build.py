"""Builds csrc/ into the in-tree C-ABI libraries with hipcc for gfx950 (cross-compiles without a GPU).

    libsliceslice_hip.so          the product: include/sliceslice_hip.h and nothing else (-fvisibility=hidden)
    libsliceslice_hip_service.so  the product's objects plus the resident search service (include/sliceslice_hip_service.h): an
                                  opt-in component outside the hot path - a process uses one library or the other
    libsliceslice_hip_tools.so    benchmark helpers (synthetic haystack generator, read ceiling, self-test): ss_tools.hip
    libsliceslice_hip_tuning.so   the product's sources with -DSS_TUNING_VARIANTS -DSS_TEST_HOOKS: every kernel variant,
                                  ss_searcher_set_variant / _set_grid, fault injection (tools/, the variant and hook tests)
    libsliceslice_hip_asan.so / _tsan.so   the host code under sanitizers (with the test hooks), device code untouched
    tests/native/libfake_rccl.so  the shared-memory RCCL stand-in of the multi-rank tests (test infrastructure)

Every translation unit is compiled in parallel.  Concurrency: every rank of a `torch.distributed.run` job may call build() at
once on a fresh clone.  Each build runs under an exclusive fcntl lock, objects and libraries are written to temporary names and
os.replace()d into place, so no process can ever CDLL a half-written file, and the up-to-date check compares the library with
its objects (a failed link leaves an old .so behind newer .o files)."""
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
_TOOLS_SO = os.path.join(_CSRC, "libsliceslice_hip_tools.so")
_SERVICE_SO = os.path.join(_CSRC, "libsliceslice_hip_service.so")
# host-side translation units (ss_internal.hpp lists what each holds) ...
_HOST_SOURCES = ["ss_core.hip", "ss_scan.hip", "ss_census.hip", "ss_host.hip", "ss_batched.hip", "ss_comm.hip"]
_SERVICE_SOURCES = ["ss_service.hip"]       # NOT in the product: libsliceslice_hip_service.so and the hooks builds
# ... and the scan kernel family, one explicit-instantiation unit per (U, load flavour, search / find).  The product holds what
# the constructors and ss_searcher_set_filter3 can select (scan_launch.hpp::kernel_built): U = 4, non-temporal loads.
_KERNEL_SOURCES = ["scan_inst_u4_nt1.hip", "scan_inst_find_nt1.hip"]
_SOURCES = _HOST_SOURCES + _KERNEL_SOURCES
# The tuning build adds every variant ss_searcher_set_variant can name: plain loads, U = 8.
_TUNING_SOURCES = _SOURCES + _SERVICE_SOURCES + ["scan_inst_u4_nt0.hip", "scan_inst_find_nt0.hip", "scan_inst_u8_nt0.hip", "scan_inst_u8_nt1.hip"]
_HEADERS = ["scan_filters.hpp", "scan_kernels.hpp", "scan_launch.hpp", "batched_kernels.hpp", "service_kernels.hpp", "aux_kernels.hpp",
            "ss_internal.hpp", os.path.join("..", "..", "include", "sliceslice_hip.h"),
            os.path.join("..", "..", "include", "sliceslice_hip_service.h"),
            os.path.join("..", "..", "include", "sliceslice_hip_tuning.h")]
_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-fvisibility=hidden"]
_HOOK_FLAGS = ["-DSS_TEST_HOOKS=1"]
# Host-side sanitizer builds (the reference's guard on its unsafe code is its ASAN CI job,
# .github/workflows/check.yml:42-58): ASan + UBSan / TSan on the HOST code of the same sources, device code untouched.
_SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g", "-O1"]
_TSAN_FLAGS = ["-fsanitize=thread", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g", "-O1"]
_NATIVE_BENCH_SRC = os.path.join(_ROOT, "tools", "native_bench.cpp")
_NATIVE_BENCH = os.path.join(_ROOT, "tools", "native_bench")
_FAKE_RCCL_SRC = os.path.join(_ROOT, "tests", "native", "fake_rccl.c")
_FAKE_RCCL = os.path.join(_ROOT, "tests", "native", "libfake_rccl.so")


def library_path():
    return _SO


def tools_library_path():
    return _TOOLS_SO


def service_library_path():
    return _SERVICE_SO


def native_bench_path():
    return _NATIVE_BENCH


def fake_rccl_path():
    return _FAKE_RCCL


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


def _build_variant(so, obj_suffix, extra_flags, link_flags, force, verbose, sources, own=None, record_resources=False):
    """Compiles `sources` (those in `own` - default: all - with `extra_flags` into <name><obj_suffix>; the others are taken as the
    regular build's <name>.o) and links them into `so`."""
    own = sources if own is None else own
    newest_header = max(_mtime(h) for h in _HEADERS)
    objs = [os.path.join(_CSRC, s[:-4] + (obj_suffix if s in own else ".o")) for s in sources]
    todo = []
    for src, obj in zip(sources, objs):
        if src not in own:
            continue
        stale = record_resources and not os.path.exists(obj + ".res")            # objects from before the resource record
        if force or stale or not os.path.exists(obj) or os.path.getmtime(obj) < max(_mtime(src), newest_header):
            todo.append((src, obj))
    if (not todo and os.path.exists(so) and os.path.getmtime(so) >= max(os.path.getmtime(o) for o in objs) and
            (not record_resources or os.path.exists(_RESOURCES))):
        return so
    hipcc = _hipcc()
    tag = ".tmp%d" % os.getpid()

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + _FLAGS + extra_flags + ["-c", os.path.join(_CSRC, src), "-o", obj + tag]
        if not record_resources:
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
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-Bsymbolic-functions", "-o", so + tag] + link_flags + objs +
         ["-ldl", "-pthread"], verbose)
    os.replace(so + tag, so)
    if record_resources:
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
    14 % slower at three, and which side of 128 the allocator lands on moves with unrelated edits; scalar-register spills
    (v_writelane traffic in front of every short-lived workgroup's first load) doubled unnoticed in round 3."""
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
    """Rows of csrc/kernel_resources.json (written by build()): kernel, name, vgprs, waves_per_simd, spills, ...  Product library
    only (the tools library's kernels are in csrc/ss_tools.o.res)."""
    build()
    return json.load(open(_RESOURCES))


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> csrc/libsliceslice_hip.so (+ libsliceslice_hip_tools.so).  Returns the product's path."""
    with _Lock(".build.lock"):
        so = _build_variant(_SO, ".o", [], [], force, verbose, _SOURCES, record_resources=True)
    build_tools(force=force, verbose=verbose)
    return so


def build_service(force=False, verbose=False):
    """The product's objects + csrc/ss_service.hip -> csrc/libsliceslice_hip_service.so (include/sliceslice_hip_service.h): the
    resident search service is an opt-in component outside the hot path and ships apart from the drop-in library."""
    build(verbose=verbose)                                  # the product's objects are shared
    with _Lock(".build_service.lock"):
        return _build_variant(_SERVICE_SO, ".o", [], [], force, verbose, _SOURCES + _SERVICE_SOURCES, own=_SERVICE_SOURCES)


def build_tools(force=False, verbose=False):
    """csrc/ss_tools.hip -> csrc/libsliceslice_hip_tools.so: the benchmark helpers of include/sliceslice_hip_tuning.h, group 1."""
    with _Lock(".build_toolslib.lock"):
        return _build_variant(_TOOLS_SO, ".o", [], [], force, verbose, ["ss_tools.hip"])


def build_sanitized(force=False, verbose=False):
    """The host logic with ASan + UBSan and the test hooks -> csrc/libsliceslice_hip_asan.so (test builds only:
    load it with SLICESLICE_HIP_LIB=<path> and the ASan runtime preloaded; see tests/test_gpu_native.py)."""
    so = os.path.join(_CSRC, "libsliceslice_hip_asan.so")
    build(verbose=verbose)                                  # the kernel-instantiation objects are shared with the regular build
    with _Lock(".build_asan.lock"):
        return _build_variant(so, ".asan.o", _SAN_FLAGS + _HOOK_FLAGS, ["-fsanitize=address,undefined", "-shared-libsan"], force, verbose,
                              _SOURCES + _SERVICE_SOURCES, own=_HOST_SOURCES + _SERVICE_SOURCES)


def build_tsan(force=False, verbose=False):
    """ThreadSanitizer build of the host code (with the test hooks) -> csrc/libsliceslice_hip_tsan.so (tests/test_gpu_native.py)."""
    so = os.path.join(_CSRC, "libsliceslice_hip_tsan.so")
    build(verbose=verbose)
    with _Lock(".build_tsan.lock"):
        return _build_variant(so, ".tsan.o", _TSAN_FLAGS + _HOOK_FLAGS, ["-fsanitize=thread", "-shared-libsan"], force, verbose,
                              _SOURCES + _SERVICE_SOURCES, own=_HOST_SOURCES + _SERVICE_SOURCES)


def tsan_runtime():
    out = subprocess.check_output([_hipcc(), "-print-file-name=libclang_rt.tsan-x86_64.so"], text=True).strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def asan_runtime():
    """Path of the clang ASan runtime that build_sanitized() links against (to LD_PRELOAD into python)."""
    out = subprocess.check_output([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def build_native_bench(force=False, verbose=False):
    """tools/native_bench.cpp -> tools/native_bench: the measurements that must not have Python or torch in the
    loop (per-call latencies, the config-1 per-needle loop, the multi-rank overheads), linked against the in-tree libraries."""
    build(verbose=verbose)
    so = build_service(verbose=verbose)         # (a superset of the product: the tool also times the resident service)
    with _Lock(".build_tools.lock"):
        if (not force and os.path.exists(_NATIVE_BENCH) and
                os.path.getmtime(_NATIVE_BENCH) >= max(os.path.getmtime(_NATIVE_BENCH_SRC), os.path.getmtime(so), os.path.getmtime(_TOOLS_SO))):
            return _NATIVE_BENCH
        tmp = _NATIVE_BENCH + ".tmp%d" % os.getpid()
        _run([_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(_ROOT, "include"), _NATIVE_BENCH_SRC, "-o", tmp,
              "-L", _CSRC, "-lsliceslice_hip_service", "-lsliceslice_hip_tools", "-Wl,-rpath,$ORIGIN/../sliceslice-rs_amd/csrc",
              "-Wl,-rpath,/opt/rocm/lib", "-pthread"], verbose)
        os.replace(tmp, _NATIVE_BENCH)
        return _NATIVE_BENCH


def build_fake_rccl(force=False, verbose=False):
    """tests/native/fake_rccl.c -> tests/native/libfake_rccl.so: the shared-memory stand-in that lets the native collective code
    run with several ranks on ONE GPU (real RCCL refuses that).  Test infrastructure; handed to the library with
    SLICESLICE_RCCL_LIB=<path>."""
    with _Lock(".build_fake_rccl.lock"):
        if not force and os.path.exists(_FAKE_RCCL) and os.path.getmtime(_FAKE_RCCL) >= os.path.getmtime(_FAKE_RCCL_SRC):
            return _FAKE_RCCL
        tmp = _FAKE_RCCL + ".tmp%d" % os.getpid()
        _run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-Wall", "-I/opt/rocm/include", _FAKE_RCCL_SRC, "-o", tmp,
              "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-pthread", "-Wl,-rpath,/opt/rocm/lib"], verbose)
        os.replace(tmp, _FAKE_RCCL)
        return _FAKE_RCCL


def tuning_library_path():
    return os.path.join(_CSRC, "libsliceslice_hip_tuning.so")


def build_tuning(force=False, verbose=False):
    """Every kernel variant ss_searcher_set_variant can name plus the test hooks (-DSS_TUNING_VARIANTS -DSS_TEST_HOOKS, with the plain-load
    and U = 8 translation units) -> csrc/libsliceslice_hip_tuning.so.  Not the product: tools/ and the variant / hook tests load it
    with SLICESLICE_HIP_LIB=<path>."""
    so = tuning_library_path()
    with _Lock(".build_tuning.lock"):
        return _build_variant(so, ".tuning.o", ["-DSS_TUNING_VARIANTS=1"] + _HOOK_FLAGS, [], force, verbose, _TUNING_SOURCES)


def build_ab(name, defines, force=False, verbose=False):
    """A/B builds of the SAME sources with extra -D flags -> csrc/libsliceslice_hip_<name>.so (tuning only; load
    with SLICESLICE_HIP_LIB=<path>, see tools/ab_compare.py)."""
    so = os.path.join(_CSRC, "libsliceslice_hip_%s.so" % name)
    with _Lock(".build_ab_%s.lock" % name):
        return _build_variant(so, ".%s.o" % name, ["-D" + d for d in defines] + _HOOK_FLAGS, [], force, verbose, _SOURCES + _SERVICE_SOURCES)
