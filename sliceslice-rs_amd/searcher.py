"""Python host-side mirror of the reference's searcher interface over the C ABI.

Names and argument meaning follow the reference (all paths under /root/reference):

    DynamicHipSearcher.new(needle)                  DynamicAvx2Searcher::new            src/x86.rs:454
    DynamicHipSearcher.with_position(needle, pos)   DynamicAvx2Searcher::with_position  src/x86.rs:468
    HipSearcher.new / .with_position                Avx2Searcher::new / ::with_position src/x86.rs:282,297
    MemchrHipSearcher.new(byte)                     MemchrSearcher::new                 src/lib.rs:124
    searcher.search_in(haystack) -> bool            DynamicAvx2Searcher::search_in      src/x86.rs:523
    PositionError                                   the `assert!` panics                src/x86.rs:300,473

``search_in`` accepts a CUDA/HIP ``torch.Tensor`` of dtype uint8 (device path, no copy), a raw
``(device_pointer, length)`` pair, or host bytes / bytearray / numpy uint8 (uploaded, then scanned on
the GPU).  There is no CPU search path: if the HIP library cannot be loaded, or no GPU is visible,
construction raises.

PyTorch is used only as plumbing (device memory, streams, torch.distributed); the C ABI has no torch
types in its signatures and this module does not import torch unless a tensor is passed in.
"""
import ctypes
import os
import sys

import numpy as np

from . import _build

SS_OK, SS_ERR_POSITION, SS_ERR_ARGUMENT, SS_ERR_NO_DEVICE, SS_ERR_HIP, SS_ERR_RCCL, SS_ERR_NOMEM, SS_ERR_PEER = range(8)

_lib = None

_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_u64 = ctypes.c_uint64
_int = ctypes.c_int
_pint = ctypes.POINTER(ctypes.c_int)
_pvp = ctypes.POINTER(ctypes.c_void_p)

_psz = ctypes.POINTER(_sz)
_pu64 = ctypes.POINTER(_u64)

# every symbol include/sliceslice_hip.h declares (the product library): name -> (restype, argtypes)
ABI = {
    "ss_searcher_new": (_int, [_vp, _sz, _pvp]),
    "ss_searcher_with_position": (_int, [_vp, _sz, _sz, _pvp]),
    "ss_searcher_free": (None, [_vp]),
    "ss_searcher_info": (_int, [_vp, _psz, _psz]),
    "ss_searcher_filter3": (_int, [_vp, _psz, _psz, _psz]),
    "ss_searcher_set_filter3": (_int, [_vp, _sz, _sz, _sz]),
    "ss_byte_histogram_device": (_int, [_vp, _sz, _sz, _vp, _vp]),
    "ss_choose_position": (_int, [_vp, _sz, _vp, _psz]),
    "ss_choose_filter_triple": (_int, [_vp, _sz, _vp, _psz, _psz, _psz]),
    "ss_search_device": (_int, [_vp, _vp, _sz, _vp, _pint]),
    "ss_search_device_async": (_int, [_vp, _vp, _sz, _vp, _vp]),
    "ss_find_device": (_int, [_vp, _vp, _sz, _vp, _pu64]),
    "ss_find_host": (_int, [_vp, _vp, _sz, _pu64]),
    "ss_find_device_async": (_int, [_vp, _vp, _sz, _u64, _vp, _vp]),
    "ss_search_host": (_int, [_vp, _vp, _sz, _pint]),
    "ss_search_file": (_int, [_vp, ctypes.c_char_p, _pint]),
    "ss_search_batched": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "ss_find_batched": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "ss_batch_plan_create": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _int, _vp, _pvp]),
    "ss_batch_plan_run": (_int, [_vp, _vp, _vp]),
    "ss_batch_plan_free": (None, [_vp]),
    "ss_search_pairs": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "ss_searcher_set_timing": (_int, [_vp, _int]),
    "ss_searcher_last_kernel_ms": (_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "ss_searcher_last_launch": (_int, [_vp, _pint, ctypes.POINTER(ctypes.c_uint)]),
    "ss_shard_range": (_int, [_sz, _sz, _int, _int, _psz, _psz]),
    "ss_comm_unique_id": (_int, [_vp]),
    "ss_comm_init_rank": (_int, [_vp, _int, _int, _pvp]),
    "ss_comm_free": (None, [_vp]),
    "ss_comm_count": (_int, [_vp, _pint]),
    "ss_comm_rccl_info": (_int, [ctypes.c_char_p, _sz, _pint]),
    "ss_search_sharded": (_int, [_vp, _vp, _sz, _vp, _vp, _pint]),
    "ss_find_sharded": (_int, [_vp, _vp, _sz, _u64, _vp, _vp, _pu64]),
    "ss_comm_init_all": (_int, [_int, _pint, _pvp]),
    "ss_comm_set_free": (None, [_vp]),
    "ss_comm_set_combine": (_int, [_vp, _int]),
    "ss_comm_set_issue": (_int, [_vp, _int]),
    "ss_comm_set_count": (_int, [_vp, _pint]),
    "ss_comm_set_last_kernel_ms": (_int, [_vp, ctypes.POINTER(ctypes.c_float), _int]),
    "ss_comm_set_last_issue_us": (_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "ss_search_sharded_all": (_int, [_vp, _pvp, _psz, _vp, _pint]),
    "ss_find_sharded_all": (_int, [_vp, _pvp, _psz, _pu64, _vp, _pu64]),
    "ss_set_autotune": (_int, [_int]),
    "ss_searcher_tuning_state": (_int, [_vp, _vp, _sz, _vp]),
    "ss_last_error": (ctypes.c_char_p, []),
    "ss_device_info": (_int, [ctypes.c_char_p, _sz, _pint, _psz]),
}
# include/sliceslice_hip_service.h: the resident search service, an opt-in component outside the hot path - NOT in the product
# library: libsliceslice_hip_service.so (the product's objects plus the service) and the hooks builds hold it
SERVICE_ABI = {
    "ss_service_start": (_int, [_int, ctypes.c_double, _pvp]),
    "ss_service_search": (_int, [_vp, _vp, _vp, _sz, _pint]),
    "ss_service_bind": (_int, [_vp, _vp, _sz]),
    "ss_service_stop": (None, [_vp]),
}
# include/sliceslice_hip_tuning.h, group 1: libsliceslice_hip_tools.so
TOOLS_ABI = {
    "ss_fill_random_device": (_int, [_vp, _u64, _sz, _u64, _vp]),
    "ss_fill_random_host": (_int, [_vp, _u64, _sz, _u64]),
    "ss_read_ceiling": (_int, [_vp, _sz, _vp, _int, ctypes.POINTER(ctypes.c_float)]),
    "ss_selftest_dpp": (_int, [_vp]),
    "ss_tools_last_error": (ctypes.c_char_p, []),
}
# ... group 2: builds with -DSS_TEST_HOOKS only (libsliceslice_hip_tuning.so, the sanitizer builds)
HOOKS_ABI = {
    "ss_version": (ctypes.c_char_p, []),
    "ss_searcher_set_variant": (_int, [_vp, _int]),
    "ss_searcher_set_grid": (_int, [_vp, _int]),
    "ss_choose_filter_for_position": (_int, [_vp, _sz, _sz, _psz, _psz, _psz]),
    "ss_debug_set_epochs": (_int, [_vp, _int]),
    "ss_debug_set_completion_state": (_int, [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]),
    "ss_debug_set_comm_epoch": (_int, [_vp, _vp, _int]),
    "ss_debug_late_answers": (_u64, []),
    "ss_debug_fail_next_scans": (_int, [_vp, _int]),
    "ss_debug_census": (_int, [_vp, _vp, _sz, ctypes.POINTER(ctypes.c_uint32)]),
    "ss_debug_census_stats": (_int, [_vp, _vp, _sz, ctypes.POINTER(ctypes.c_uint32), _pint]),
    "ss_debug_plan_filter": (_int, [_vp, _sz, ctypes.POINTER(ctypes.c_uint32)]),
    "ss_debug_plan_cold": (_int, [_vp, _sz, ctypes.POINTER(ctypes.c_uint32)]),
    "ss_debug_plan_layout": (_int, [_vp, ctypes.POINTER(ctypes.c_uint32)]),
    "ss_debug_batch_classes": (_int, [_vp, _vp, _sz, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint8)]),
    "ss_service_counters": (_int, [_vp, _pu64, _pu64, _pu64]),
}


class TuningState(ctypes.Structure):
    """ss_tuning_state (include/sliceslice_hip.h): every launch-tuning state a handle holds for one haystack."""
    _fields_ = [(k, ctypes.c_uint32) for k in ("autotune", "census_state", "census_age", "tiles", "tiles3", "tiles2", "match_tiles", "lanes",
                                               "pair_lanes", "triple_lanes", "deep_lanes", "triple_state", "on_trial", "trials", "accepted", "settled", "proposal")] + \
               [("own", ctypes.c_uint32 * 3), ("in_force", ctypes.c_uint32 * 3), ("order_measured", ctypes.c_uint32), ("norder", ctypes.c_uint32),
                ("order", ctypes.c_uint8 * 16), ("histogram_state", ctypes.c_uint32), ("workgroups_per_cu", ctypes.c_uint32),
                ("grid", ctypes.c_uint32), ("kernel_mode", ctypes.c_uint32), ("last_found", ctypes.c_uint32)]

    def as_dict(self):
        d = {}
        for k, _ in self._fields_:
            v = getattr(self, k)
            d[k] = list(v) if hasattr(v, "__len__") else int(v)
        d["order"] = d["order"][:d["norder"]]
        return d


def rccl_info():
    """(path of the librccl the native communicators use, its ncclGetVersion) - ss_comm_rccl_info."""
    buf = ctypes.create_string_buffer(512)
    ver = ctypes.c_int(0)
    _check(lib().ss_comm_rccl_info(buf, len(buf), ctypes.byref(ver)))
    return buf.value.decode("utf-8", "replace"), int(ver.value)


def set_autotune(enabled):
    """ss_set_autotune: launch tuning on (the default) or off, process-wide; returns the previous setting."""
    return bool(lib().ss_set_autotune(1 if enabled else 0))


class SlicesliceError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sliceslice_hip error %d: %s" % (code, msg))
        self.code = code


class PositionError(SlicesliceError, AssertionError):
    """The reference panics here (src/x86.rs:300 `assert!(position < needle.size())`,
    src/x86.rs:473 `assert_eq!(position, 0)`)."""


def _preload_torch():
    if "torch" in sys.modules or os.environ.get("SLICESLICE_PRELOAD_TORCH", "1") == "1":
        # torch wheels bundle their own libamdhip64 (same soname).  Loading torch first makes this
        # library bind to the SAME HIP runtime, so torch streams/pointers are valid in it.
        try:
            import torch  # noqa: F401
        except Exception:
            pass


def _bind(L, table, strict):
    for name, (res, args) in table.items():
        if not strict and not hasattr(L, name):
            continue
        fn = getattr(L, name)          # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    return L


def _load(path):
    """One build of the library: every product symbol must be there; the hooks are bound where the build has them."""
    _preload_torch()
    L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    _bind(L, ABI, strict=True)
    _bind(L, SERVICE_ABI, strict=False)
    _bind(L, HOOKS_ABI, strict=False)
    L.has_hooks = hasattr(L, "ss_debug_fail_next_scans")
    L.has_service = hasattr(L, "ss_service_start")
    return L


def lib():
    """Loads csrc/libsliceslice_hip.so (building it with hipcc if it is missing).  Fails loudly.
    SLICESLICE_HIP_LIB=<path> loads another build of the SAME library instead (the tuning build, an A/B build)."""
    global _lib
    if _lib is None:
        _lib = _load(os.environ.get("SLICESLICE_HIP_LIB") or _build.build())
    return _lib


_tools = None
_tuning = None
_service = None


def tools_lib():
    """csrc/libsliceslice_hip_tools.so: the benchmark helpers (synthetic haystack generator, read ceiling, self-test)."""
    global _tools
    if _tools is None:
        _build.build()
        _preload_torch()
        _tools = _bind(ctypes.CDLL(_build.tools_library_path(), mode=ctypes.RTLD_LOCAL), TOOLS_ABI, strict=True)
    return _tools


class tuning_build:
    """``with ss.tuning_build():`` - inside the block ``lib()`` is libsliceslice_hip_tuning.so (every kernel variant, the test
    hooks of include/sliceslice_hip_tuning.h).  Objects remember the library they were made with, so searchers created
    inside keep working (and are freed by the right library) after the block."""

    def __enter__(self):
        global _lib, _tuning
        if _tuning is None:
            _tuning = _load(_build.build_tuning())
        self._saved, _lib = _lib, _tuning
        return _tuning

    def __exit__(self, *a):
        global _lib
        _lib = self._saved
        return False


class service_build:
    """``with ss.service_build():`` - inside the block ``lib()`` is libsliceslice_hip_service.so: every function of the product
    library plus the resident search service (include/sliceslice_hip_service.h).  Searchers belong to the library that made
    them, so the searchers a SearchService is asked about must be created inside the block too."""

    def __enter__(self):
        global _lib, _service
        if _service is None:
            _service = _load(_build.build_service())
        self._saved, _lib = _lib, _service
        return _service

    def __exit__(self, *a):
        global _lib
        _lib = self._saved
        return False


def _hooks(L):
    if not getattr(L, "has_hooks", False):
        raise SlicesliceError(SS_ERR_ARGUMENT, "this entry point exists in builds with -DSS_TEST_HOOKS only (libsliceslice_hip_tuning.so: "
                                               "`with ss.tuning_build():` or SLICESLICE_HIP_LIB=<path>)")
    return L


def _check(rc, L=None):
    if rc != SS_OK:
        msg = (L or lib()).ss_last_error().decode("utf-8", "replace")
        raise (PositionError if rc == SS_ERR_POSITION else SlicesliceError)(rc, msg)


def _check_tools(rc):
    if rc != SS_OK:
        raise SlicesliceError(rc, tools_lib().ss_tools_last_error().decode("utf-8", "replace"))


def _host_view(b):
    """(keepalive, address, length) of a host buffer."""
    if isinstance(b, np.ndarray):
        a = np.ascontiguousarray(b, dtype=np.uint8)
        return a, (a.ctypes.data if a.size else 0), a.size
    if isinstance(b, bytes):
        a = np.frombuffer(b, dtype=np.uint8)
        return (a, b), (a.ctypes.data if a.size else 0), a.size
    if isinstance(b, (bytearray, memoryview)):
        a = np.frombuffer(b, dtype=np.uint8)
        return (a, b), (a.ctypes.data if a.size else 0), a.size
    raise TypeError("unsupported haystack/needle type %r" % type(b))


def _is_tensor(x):
    return type(x).__module__.split(".")[0] == "torch" and hasattr(x, "data_ptr")


def _current_stream_handle():
    if "torch" in sys.modules:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_stream().cuda_stream
    return 0


class _on_device_of:
    """The C side launches on the CURRENT device (its needle copy, its flag slots) and, by default, on torch's
    current stream OF THAT DEVICE: make the tensor's device current for the duration of the call, so that a
    haystack on cuda:1 is never scanned by a kernel launched on cuda:0."""

    def __init__(self, t):
        self._ctx = None
        if _is_tensor(t) and t.is_cuda:
            import torch
            if t.device.index != torch.cuda.current_device():
                self._ctx = torch.cuda.device(t.device)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self._ctx is not None:
            self._ctx.__exit__(*a)
        return False


class DynamicHipSearcher:
    """GPU counterpart of ``sliceslice::x86::DynamicAvx2Searcher`` (src/x86.rs:405-525)."""

    def __init__(self, needle, position=None):
        nb = needle.astype(np.uint8).tobytes() if isinstance(needle, np.ndarray) else bytes(needle)
        k, addr, n = _host_view(nb)
        self._h = ctypes.c_void_p()
        self._needle = nb
        L = self._L = lib()                 # the build this searcher belongs to (see tuning_build)
        if position is None:
            _check(L.ss_searcher_new(addr, n, ctypes.byref(self._h)), L)
        else:
            _check(L.ss_searcher_with_position(addr, n, position % (1 << 64), ctypes.byref(self._h)), L)

    def _ck(self, rc):
        _check(rc, self._L)

    # -- reference-shaped constructors ---------------------------------------------------------------
    @classmethod
    def new(cls, needle):
        return cls(needle)

    @classmethod
    def with_position(cls, needle, position):
        return cls(needle, position)

    @property
    def needle(self):
        return self._needle

    @property
    def position(self):
        n, pos = _sz(0), _sz(0)
        self._ck(self._L.ss_searcher_info(self._h, ctypes.byref(n), ctypes.byref(pos)))
        return pos.value

    # -- the hot path ------------------------------------------------------------------------------------
    def search_in(self, haystack, stream=None):
        """bool: does the needle occur in ``haystack``?  (DynamicAvx2Searcher::search_in)"""
        found = ctypes.c_int(0)
        if _is_tensor(haystack):
            if not haystack.is_cuda:
                return self.search_in(haystack.numpy())
            if haystack.dtype.itemsize != 1 or not haystack.is_contiguous():
                raise TypeError("device haystack must be a contiguous 1-byte tensor")
            with _on_device_of(haystack):
                st = stream if stream is not None else _current_stream_handle()
                self._ck(self._L.ss_search_device(self._h, haystack.data_ptr(), haystack.numel(), st, ctypes.byref(found)))
        elif isinstance(haystack, tuple):
            ptr, length = haystack
            st = stream if stream is not None else _current_stream_handle()
            self._ck(self._L.ss_search_device(self._h, ptr, length, st, ctypes.byref(found)))
        else:
            k, addr, n = _host_view(haystack)
            self._ck(self._L.ss_search_host(self._h, addr, n, ctypes.byref(found)))
        return bool(found.value)

    inlined_search_in = search_in       # src/x86.rs:498

    def find(self, haystack, stream=None):
        """Offset of the leftmost occurrence or None (row f1; the shape of tests/i386.rs:6-10
        `find_subsequence` and of the competitors in bench/benches/i386.rs).  Device tensors /
        (pointer, length) pairs use ss_find_device, host buffers ss_find_host."""
        pos = _u64(0)
        if isinstance(haystack, tuple) or (_is_tensor(haystack) and haystack.is_cuda):
            ptr, length = haystack if isinstance(haystack, tuple) else (haystack.data_ptr(), haystack.numel())
            with _on_device_of(haystack):
                st = stream if stream is not None else _current_stream_handle()
                self._ck(self._L.ss_find_device(self._h, ptr, length, st, ctypes.byref(pos)))
        else:
            if _is_tensor(haystack):
                haystack = haystack.numpy()
            k, addr, n = _host_view(haystack)
            self._ck(self._L.ss_find_host(self._h, addr, n, ctypes.byref(pos)))
        return None if pos.value == (1 << 64) - 1 else pos.value

    def find_async(self, haystack, d_best, base_offset=0, stream=None):
        """Enqueue only: atomicMin base_offset + offset into the uint64 device tensor d_best (init: all ones)."""
        with _on_device_of(haystack):
            st = stream if stream is not None else _current_stream_handle()
            self._ck(self._L.ss_find_device_async(self._h, haystack.data_ptr(), haystack.numel(), base_offset, st,
                                              d_best.data_ptr()))

    def search_in_async(self, haystack, d_flag, stream=None):
        """Enqueue only: OR the result into the int32 device tensor ``d_flag`` (caller-zeroed)."""
        with _on_device_of(haystack):
            st = stream if stream is not None else _current_stream_handle()
            self._ck(self._L.ss_search_device_async(self._h, haystack.data_ptr(), haystack.numel(), st, d_flag.data_ptr()))

    # -- tuning / measurement hooks ------------------------------------------------------------------
    @property
    def filter(self):
        """(first, second): indices of the first two needle bytes the filter tests."""
        return self.filter3[:2]

    @property
    def filter3(self):
        """(first, second, third): the bytes of the first-phase filter (third == second: a two-byte filter)."""
        a, b, c = _sz(0), _sz(0), _sz(0)
        self._ck(self._L.ss_searcher_filter3(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def set_filter(self, first, second, third=None):
        """ss_searcher_set_filter3; third=None: a plain two-byte filter (third == second)."""
        self._ck(self._L.ss_searcher_set_filter3(self._h, first, second, second if third is None else third))

    def set_timing(self, on=True):
        self._ck(self._L.ss_searcher_set_timing(self._h, int(on)))

    def last_kernel_ms(self):
        ms = ctypes.c_float(0)
        self._ck(self._L.ss_searcher_last_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def set_variant(self, variant):
        """Tuning builds only (ss_searcher_set_variant); 0 - the automatic choice - is accepted by every build."""
        if int(variant) != 0 or getattr(self._L, "has_hooks", False):
            self._ck(_hooks(self._L).ss_searcher_set_variant(self._h, int(variant)))

    def set_grid(self, blocks):
        if int(blocks) != 0 or getattr(self._L, "has_hooks", False):
            self._ck(_hooks(self._L).ss_searcher_set_grid(self._h, int(blocks)))

    def last_launch(self):
        """(workgroups per CU, workgroups in the grid) of the latest scan enqueued through this searcher on the current device."""
        w, g = ctypes.c_int(0), ctypes.c_uint(0)
        self._ck(self._L.ss_searcher_last_launch(self._h, ctypes.byref(w), ctypes.byref(g)))
        return w.value, g.value

    def device_triple(self):
        """The three first-phase bytes the device tests by default (ss_searcher_tuning_state.own): the searcher's triple, with the third byte
        the library adds to a plain pair."""
        st = TuningState()
        self._ck(self._L.ss_searcher_tuning_state(self._h, None, 0, ctypes.byref(st)))
        return tuple(int(x) for x in st.own)

    def tuning_state(self, haystack):
        """ss_searcher_tuning_state as a dict: what this handle has learnt about `haystack` (a device tensor) and goes by."""
        st = TuningState()
        self._ck(self._L.ss_searcher_tuning_state(self._h, haystack.data_ptr(), haystack.numel(), ctypes.byref(st)))
        return st.as_dict()

    def census_stats(self, haystack):
        """Hooks builds: the census's per-position match counts {pair_match, triple_match, pair_lanes, triple_lanes}, or None."""
        c = (ctypes.c_uint32 * 131)()
        have = ctypes.c_int(0)
        self._ck(_hooks(self._L).ss_debug_census_stats(self._h, haystack.data_ptr(), haystack.numel(), c, ctypes.byref(have)))
        if not have.value:
            return None
        self.stats_roles = have.value - 1                   # the slot the pair counts were gathered for
        return {"pair_match": list(c[:64]), "triple_match": list(c[64:128]), "pair_lanes": int(c[128]), "triple_lanes": int(c[129]), "deep_lanes": int(c[130])}

    def census(self, haystack):
        """Hooks builds: the candidate census of (this searcher, haystack) as a dict, or None when its counts are not in."""
        c = (ctypes.c_uint32 * 11)()
        ptr, n = haystack.data_ptr(), haystack.numel()
        self._ck(_hooks(self._L).ss_debug_census(self._h, ptr, n, c))
        self.last_mode = int(c[5])                          # kernel family of the latest launch (0, 2 or 3)
        self.device_filter = (int(c[6]), int(c[7]), int(c[8]))      # the bytes the device tests on this haystack
        self.triple_state, self.triple_trials = int(c[9]), int(c[10])  # 0 undecided / 1 own / 2 from the histogram; trials so far
        if c[0] == 0:
            return None
        return {"tiles": c[0], "tiles3": c[1], "tiles2": c[2], "match_tiles": c[3], "lanes": c[4]}

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        L = getattr(self, "_L", None)
        if h and L is not None and _lib is not None:
            L.ss_searcher_free(h)


class HipSearcher(DynamicHipSearcher):
    """GPU counterpart of ``sliceslice::x86::Avx2Searcher`` (src/x86.rs:266-382): the searcher for needles of at
    least one byte - an EMPTY needle panics in the reference (`position = size.wrapping_sub(1)` fails
    `assert!(position < size)`, src/x86.rs:285-300; test `avx2_empty_needle`, src/x86.rs:545-549), where the dynamic
    searcher answers `true`.  Everything else is the dynamic searcher's behaviour."""

    def __init__(self, needle, position=None):
        nb = needle.astype(np.uint8).tobytes() if isinstance(needle, np.ndarray) else bytes(needle)
        if len(nb) == 0:
            raise PositionError(SS_ERR_POSITION, "Avx2Searcher contract: the needle must not be empty (src/x86.rs:300)")
        if position is not None and position >= len(nb):
            raise PositionError(SS_ERR_POSITION, "position %d out of range for a needle of %d bytes" % (position, len(nb)))
        super().__init__(nb, position)


class MemchrHipSearcher:
    """GPU counterpart of ``sliceslice::MemchrSearcher`` (src/lib.rs:119-142): one byte, `search_in` is false for an
    empty haystack."""

    def __init__(self, needle):
        b = int(needle)
        if not 0 <= b <= 255:
            raise ValueError("MemchrHipSearcher takes one byte (0..255)")
        self._inner = DynamicHipSearcher(bytes([b]))

    @classmethod
    def new(cls, needle):
        return cls(needle)

    def search_in(self, haystack, stream=None):
        return self._inner.search_in(haystack, stream)

    inlined_search_in = search_in

    def find(self, haystack, stream=None):
        return self._inner.find(haystack, stream)


def shard_range(length, needle_len, nranks, rank):
    """Byte range [begin, end) of `rank`'s shard: S = ceil(len/G), n-1 bytes of overlap to the right."""
    b, e = _sz(0), _sz(0)
    _check(lib().ss_shard_range(length, needle_len, nranks, rank, ctypes.byref(b), ctypes.byref(e)))
    return b.value, e.value


class ShardedSearcher:
    """Range-sharded search over the GPUs of one node: one process per GPU, each holding its shard.

    ``search_in(shard)`` scans the local shard and combines the found flag with ONE all-reduce(MAX)
    (OR over {0,1}; RCCL has no OR).  Two transports:
      * ``backend="rccl"``  - native RCCL through the C ABI (ss_comm_*), all on one HIP stream;
      * ``backend="torch"`` - torch.distributed.all_reduce on the given process group (RCCL on GPUs,
        gloo on CPU - used by the CPU tests with an injected shard searcher).
    """

    def __init__(self, needle, position=None, group=None, backend="torch", local_search=None, local_find=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.nranks = dist.get_world_size(group)
        self.needle = bytes(needle)
        self.backend = backend
        self._local_search = local_search
        self._local_find = local_find
        self._best = None
        self._searcher = None if (local_search is not None or local_find is not None) else \
            DynamicHipSearcher(needle, position)
        self._comm = None
        self._flag = None
        self._flag_next = 0
        self._L = self._searcher._L if self._searcher is not None else lib()
        if backend == "rccl":
            self._init_rccl()

    def shard_range(self, total_len):
        return shard_range(total_len, len(self.needle), self.nranks, self.rank)

    def _ck(self, rc):
        _check(rc, self._L)

    def fail_next_scans(self, count=1):
        """Hooks builds: the next `count` scans of this rank fail before they reach the device (ss_debug_fail_next_scans)."""
        self._ck(_hooks(self._L).ss_debug_fail_next_scans(self._searcher._h, count))

    def _init_rccl(self):
        import torch
        uid = (ctypes.c_uint8 * 128)()
        box = [None]
        if self.rank == 0 and self._L.ss_comm_unique_id(uid) == 0:
            box = [bytes(uid)]
        self._dist.broadcast_object_list(box, src=0, group=self.group)     # None: rank 0 has no id to offer
        if box[0] is None:
            raise SlicesliceError(SS_ERR_RCCL, "rank 0 could not create an RCCL unique id" +
                                  (": " + self._L.ss_last_error().decode("utf-8", "replace") if self.rank == 0 else ""))
        uid = (ctypes.c_uint8 * 128).from_buffer_copy(box[0])
        comm = ctypes.c_void_p()
        # ncclCommInitRank is a collective: a rank on which it fails (or never returns) must not leave the others
        # behind in it, and all ranks must take the same road afterwards.  It runs on a worker thread with a time
        # limit (SLICESLICE_RCCL_INIT_TIMEOUT seconds, default 180); then the ranks agree - all-reduce(MIN) of "mine
        # worked" on the torch group - and either all keep their communicator or all raise.
        import threading
        result = {}
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None

        def work():
            try:
                if dev is not None:
                    torch.cuda.set_device(dev)
                result["rc"] = self._L.ss_comm_init_rank(uid, self.nranks, self.rank, ctypes.byref(comm))
                result["err"] = self._L.ss_last_error().decode("utf-8", "replace") if result["rc"] else ""
            except Exception as e:                                   # pragma: no cover - ctypes / loader failures
                result["rc"], result["err"] = -1, repr(e)

        t = threading.Thread(target=work, daemon=True)
        t.start()
        t.join(float(os.environ.get("SLICESLICE_RCCL_INIT_TIMEOUT", "180")))
        mine = 1 if (not t.is_alive() and result.get("rc") == 0) else 0
        on_gpu = self._dist.get_backend(self.group) == "nccl"
        ok = torch.tensor([mine], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
        self._dist.all_reduce(ok, op=self._dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) != 1:
            if mine:
                self._L.ss_comm_free(comm)
            why = "timed out" if t.is_alive() else (result.get("err") or "failed on another rank")
            raise SlicesliceError(SS_ERR_RCCL, "native RCCL communicator not built on every rank (this rank: %s)" % why)
        self._comm = comm
        torch.cuda.synchronize()

    def _peer_error(self):
        return SlicesliceError(SS_ERR_PEER, "another rank failed the local part of this sharded search; no answer")

    def search_in(self, shard, stream=None):
        """Collective-safe: a rank whose local part raises still takes part in the all-reduce (contributing "not found"
        and raising the second word of the flag pair), then re-raises; the other ranks raise SlicesliceError(SS_ERR_PEER)
        - nobody is left waiting in the collective."""
        import torch
        if self._local_search is not None:                       # CPU tests: injected shard searcher
            err = None
            try:
                f = 1 if self._local_search(shard) else 0
            except Exception as e:                                # noqa: BLE001 - re-raised behind the collective
                err, f = e, 0
            flag = torch.tensor([f, 1 if err is not None else 0], dtype=torch.int32)
            self._dist.all_reduce(flag, op=self._dist.ReduceOp.MAX, group=self.group)
            if err is not None:
                raise err
            if int(flag[1]):
                raise self._peer_error()
            return bool(flag[0])
        if self.backend == "rccl":
            found = ctypes.c_int(0)
            with _on_device_of(shard):
                st = stream if stream is not None else _current_stream_handle()
                self._ck(self._L.ss_search_sharded(self._searcher._h, shard.data_ptr(), shard.numel(), self._comm, st,
                                               ctypes.byref(found)))
            return bool(found.value)
        # torch transport: the scan, the flag housekeeping and the all-reduce must all be ordered on ONE stream -
        # torch's current stream.  A caller-supplied stream (object or raw handle) is made current for the duration.
        with _on_device_of(shard), self._as_current(stream):
            # a ring of pre-zeroed flag PAIRS {found, a rank failed}: one fresh pair per call, one zero_() launch per 256
            # calls instead of per call
            if self._flag is None or self._flag_next == self._flag.numel():
                if self._flag is None:
                    self._flag = torch.zeros(512, dtype=torch.int32, device=shard.device)
                else:
                    self._flag.zero_()
                self._flag_next = 0
            flag = self._flag[self._flag_next:self._flag_next + 2]
            self._flag_next += 2
            err = None
            try:
                self._searcher.search_in_async(shard, flag[0:1])
            except Exception as e:                                # noqa: BLE001 - re-raised behind the collective
                err = e
                flag[1:2].fill_(1)
            self._dist.all_reduce(flag, op=self._dist.ReduceOp.MAX, group=self.group)
            f, failed = flag.tolist()
            if err is not None:
                raise err
            if failed:
                raise self._peer_error()
            return bool(f)

    @staticmethod
    def _as_current(stream):
        import contextlib
        import torch
        if stream is None:
            return contextlib.nullcontext()
        if isinstance(stream, int):
            stream = torch.cuda.ExternalStream(stream)
        return torch.cuda.stream(stream)

    def find(self, shard, shard_begin, stream=None):
        """Global offset of the leftmost occurrence in the logical haystack, or None: every rank finds its
        local leftmost match (offset + shard_begin), ONE all-reduce(MIN) combines them."""
        import torch
        none = (1 << 63) - 1                                      # int64 stand-in for SS_NPOS in the reduce
        err = None
        if self._local_find is not None:                          # CPU tests: injected shard find
            try:
                p = self._local_find(shard)
            except Exception as e:                                # noqa: BLE001 - re-raised behind the collective
                err, p = e, None
            t = torch.tensor([none if p is None else p + shard_begin, -1 if err is not None else none], dtype=torch.int64)
        elif self.backend == "rccl":                              # native: ncclAllReduce(uint64 pair, ncclMin)
            pos = _u64(0)
            with _on_device_of(shard):
                st = stream if stream is not None else _current_stream_handle()
                self._ck(self._L.ss_find_sharded(self._searcher._h, shard.data_ptr(), shard.numel(), shard_begin, self._comm, st,
                                             ctypes.byref(pos)))
            return None if pos.value == (1 << 64) - 1 else pos.value
        else:
            with _on_device_of(shard), self._as_current(stream):   # everything ordered on one (the current) stream
                if self._best is None:
                    self._best = torch.empty(1, dtype=torch.int64, device=shard.device)
                self._best.fill_(-1)                               # all ones = SS_NPOS
                try:
                    self._searcher.find_async(shard, self._best, shard_begin)
                except Exception as e:                            # noqa: BLE001
                    err = e
                b = torch.where(self._best < 0, torch.full_like(self._best, none), self._best)
                t = torch.cat([b, torch.full_like(b, -1 if err is not None else none)])
                self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=self.group)
                v, status = (int(x) for x in t.tolist())
            if err is not None:
                raise err
            if status != none:
                raise self._peer_error()
            return None if v == none else v
        # second word of the pair: `none` unless a rank failed its local part (MIN brings the -1 to everybody)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=self.group)
        v, status = (int(x) for x in t.tolist())
        if err is not None:
            raise err
        if status != none:
            raise self._peer_error()
        return None if v == none else v

    def rccl_ranks(self):
        """Number of ranks RCCL itself reports for the native communicator (ncclCommCount), or None."""
        if self._comm is None:
            return None
        n = ctypes.c_int(0)
        self._ck(self._L.ss_comm_count(self._comm, ctypes.byref(n)))
        return n.value

    def close(self):
        if self._comm is not None and _lib is not None:
            self._L.ss_comm_free(self._comm)
            self._comm = None


class NodeSearcher:
    """Range-sharded search over several GPUs from ONE process (ss_comm_init_all / ss_search_sharded_all):
    what a drop-in ``search_in(&self, haystack) -> bool`` (src/x86.rs:523) over the GPUs of a node calls - no
    launcher, no rendezvous.  ``shards`` is a list of uint8 tensors, shard g resident on device g of the set
    (ranges from ``shard_range``)."""

    COMBINE_RCCL, COMBINE_HOST = 0, 1
    ISSUE_THREADS, ISSUE_SERIAL = 0, 1

    def __init__(self, needle, position=None, devices=None, ndev=None):
        import torch
        if devices is None:
            devices = list(range(ndev if ndev is not None else torch.cuda.device_count()))
        self.devices = list(devices)
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        self._set = ctypes.c_void_p()
        L = self._L = lib()
        _check(L.ss_comm_init_all(len(self.devices), arr, ctypes.byref(self._set)), L)
        self._searcher = DynamicHipSearcher(needle, position)
        self.needle = bytes(needle)

    def set_combine(self, mode):
        self._ck(self._L.ss_comm_set_combine(self._set, mode))

    def set_issue(self, mode):
        """ISSUE_THREADS: one issue thread per device (the default from two devices up); ISSUE_SERIAL: the calling thread."""
        self._ck(self._L.ss_comm_set_issue(self._set, mode))

    def rccl_ranks(self):
        """ncclCommCount of every communicator of the set (they must agree), or None for a set without communicators."""
        n = ctypes.c_int(0)
        if self._L.ss_comm_set_count(self._set, ctypes.byref(n)) != 0:
            return None
        return n.value

    def last_kernel_ms(self):
        """Every device's scan-kernel time of the latest search (the searcher's timing must be on)."""
        ms = (ctypes.c_float * len(self.devices))()
        self._ck(self._L.ss_comm_set_last_kernel_ms(self._set, ms, len(self.devices)))
        return [float(x) for x in ms]

    def last_issue_us(self):
        """Host microseconds the latest search spent issuing (scans, collective, answer words, all of it)."""
        us = (ctypes.c_float * 4)()
        self._ck(self._L.ss_comm_set_last_issue_us(self._set, us))
        return [float(x) for x in us]

    def _ck(self, rc):
        _check(rc, self._L)

    def set_epoch(self, value):
        """Hooks builds: move the set's "found" epoch (ss_debug_set_comm_epoch) so that a test can cross the 2^31 wrap."""
        self._ck(_hooks(self._L).ss_debug_set_comm_epoch(None, self._set, value))

    def shard_range(self, total_len, g):
        return shard_range(total_len, len(self.needle), len(self.devices), g)

    def _args(self, shards):
        G = len(self.devices)
        assert len(shards) == G
        for g, t in enumerate(shards):
            assert t.is_cuda and t.device.index == self.devices[g] and t.dtype.itemsize == 1 and t.is_contiguous()
        ptrs = (ctypes.c_void_p * G)(*[t.data_ptr() if t.numel() else None for t in shards])
        lens = (_sz * G)(*[t.numel() for t in shards])
        return ptrs, lens

    def search_in(self, shards):
        ptrs, lens = self._args(shards)
        found = ctypes.c_int(0)
        self._ck(self._L.ss_search_sharded_all(self._searcher._h, ptrs, lens, self._set, ctypes.byref(found)))
        return bool(found.value)

    def find(self, shards, begins):
        ptrs, lens = self._args(shards)
        b = (_u64 * len(begins))(*begins)
        pos = _u64(0)
        self._ck(self._L.ss_find_sharded_all(self._searcher._h, ptrs, lens, b, self._set, ctypes.byref(pos)))
        return None if pos.value == (1 << 64) - 1 else pos.value

    def close(self):
        if getattr(self, "_set", None) and _lib is not None:
            self._L.ss_comm_set_free(self._set)
            self._set = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SearchService:
    """A resident search service on the current device (ss_service_*): a small kernel that stays on the GPU and takes one search
    at a time from a mailbox in device memory that the host writes through the PCIe BAR - no launch per search (5 us instead
    of 8.5-9.5; ``bind`` a haystack that does not change between searches).  ``search_in(searcher, haystack)`` has the semantics of
    ``searcher.search_in(haystack)`` for a device haystack whose bytes are COMPLETE (the service is not ordered behind pending
    stream work)."""

    def __init__(self, workgroups=0, lease_ms=0.0):
        self._h = ctypes.c_void_p()
        L = self._L = lib()
        if not getattr(L, "has_service", False):
            raise SlicesliceError(SS_ERR_ARGUMENT, "the resident search service is not part of libsliceslice_hip.so: it lives in "
                                                   "libsliceslice_hip_service.so (`with ss.service_build():`, or SLICESLICE_HIP_LIB=<path>) "
                                                   "and in the hooks builds")
        _check(L.ss_service_start(int(workgroups), float(lease_ms), ctypes.byref(self._h)), L)

    def _ck(self, rc):
        _check(rc, self._L)

    def search_in(self, searcher, haystack):
        found = ctypes.c_int(0)
        ptr, n = (haystack.data_ptr(), haystack.numel()) if hasattr(haystack, "data_ptr") else haystack
        assert searcher._L is self._L, "searcher and service come from different builds of the library"
        self._ck(self._L.ss_service_search(self._h, searcher._h, ptr if n else None, n, ctypes.byref(found)))
        return bool(found.value)

    def bind(self, haystack):
        """The caller vouches that this device range stays unchanged until ``unbind()`` / the next ``bind``: searches inside it
        skip the per-request cache acquire (all but the first, and those whose searcher was uploaded after it)."""
        ptr, n = (haystack.data_ptr(), haystack.numel()) if hasattr(haystack, "data_ptr") else haystack
        self._ck(self._L.ss_service_bind(self._h, ptr if n else None, n))

    def unbind(self):
        self._ck(self._L.ss_service_bind(self._h, None, 0))

    def counters(self):
        """(requests served, kernel launches, requests that skipped the acquire) - hooks builds (ss_service_counters)."""
        r, k, t = _u64(0), _u64(0), _u64(0)
        self._ck(_hooks(self._L).ss_service_counters(self._h, ctypes.byref(r), ctypes.byref(k), ctypes.byref(t)))
        return r.value, k.value, t.value

    def stop(self):
        if getattr(self, "_h", None) and _lib is not None:
            self._L.ss_service_stop(self._h)
            self._h = None

    close = stop

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.stop()

    def __del__(self):
        try:
            self.stop()
        except Exception:
            pass


class BatchPlan:
    """Plan once, search many (ss_batch_plan_*): the per-problem set-up of ``search_batched`` / ``find_batched`` done once - the
    reference builds its 4,585 searchers once and times the searches (bench/benches/i386.rs:246-256).  ``run()`` is ONE kernel
    launch that also re-arms the outputs; it can be captured into a hipGraph.  Arguments as ``search_batched``; the tensors
    must stay alive (and their ranges / needle bytes unchanged) as long as the plan is used."""

    def __init__(self, haystacks, hay_off, needles, needle_off, position=None, find=False, stream=None, hay_ranges=None,
                 needle_ranges=None):
        hb, he, count = _ranges(hay_off, *(hay_ranges or (None, None)))
        nb, ne, ncount = _ranges(needle_off, *(needle_ranges or (None, None)))
        assert count == ncount
        self._keep = (haystacks, hay_off, needles, needle_off, position, hay_ranges, needle_ranges)
        self.count, self.find, self.device = count, bool(find), haystacks.device
        self._h = ctypes.c_void_p()
        L = self._L = lib()
        st = stream if stream is not None else _current_stream_handle()
        _check(L.ss_batch_plan_create(haystacks.data_ptr(), hb, he, needles.data_ptr(), nb, ne,
                                      position.data_ptr() if position is not None else None, count, int(self.find), st,
                                      ctypes.byref(self._h)), L)

    def run(self, out=None, stream=None):
        """Enqueues the search; returns the output tensor (int32 flags, or int64 offsets with -1 = absent for find plans)."""
        import torch
        if out is None:
            out = torch.empty(self.count, dtype=torch.int64 if self.find else torch.int32, device=self.device)
        st = stream if stream is not None else _current_stream_handle()
        _check(self._L.ss_batch_plan_run(self._h, st, out.data_ptr()), self._L)
        return out

    def filter_of(self, problem):
        """((first, second, third) indices in the needle, the packed bytes, slices that scan the problem) of one problem's
        descriptor - hooks builds (ss_debug_plan_filter)."""
        out = (ctypes.c_uint32 * 5)()
        _check(_hooks(self._L).ss_debug_plan_filter(self._h, int(problem), out), self._L)
        return (out[0], out[1], out[2]), out[3], out[4]

    def layout(self):
        """{"two": the plan holds two layouts, "slices": (first, second), "found_last": problems found in the latest tallied run,
        "next_is_second": the next run takes the contiguous-runs layout} - hooks builds (ss_debug_plan_layout)."""
        out = (ctypes.c_uint32 * 5)()
        _check(_hooks(self._L).ss_debug_plan_layout(self._h, out), self._L)
        return {"two": bool(out[0]), "slices": (out[1], out[2]), "found_last": out[3], "next_is_second": bool(out[4])}

    def cold_of(self, problem):
        """(schedule indices, schedule bytes, exact_len, bytes in front, the compare's 16 bytes) of one problem's ready-made cold
        part - hooks builds (ss_debug_plan_cold)."""
        out = (ctypes.c_uint32 * 14)()
        _check(_hooks(self._L).ss_debug_plan_cold(self._h, int(problem), out), self._L)
        n = out[0]
        idx = b"".join(int(out[2 + t]).to_bytes(4, "little") for t in range(4))[:n]
        val = b"".join(int(out[6 + t]).to_bytes(4, "little") for t in range(4))[:n]
        tail = b"".join(int(out[10 + t]).to_bytes(4, "little") for t in range(4))
        return list(idx), val, out[1] & 0xFF, out[1] >> 8, tail

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            self._L.ss_batch_plan_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def batch_classes(haystacks, hay_off=None, hay_ranges=None):
    """(state, classes) the library remembers for the UNPLANNED batch calls on these haystacks / ranges: state 0 unknown, 1 named
    once, 2 sampling in flight, 3 classes in (then a list of 256 rarity classes, 0 = rarest) - hooks builds (ss_debug_batch_classes)."""
    hb, _, count = _ranges(hay_off, *(hay_ranges or (None, None)))
    st, cls = ctypes.c_uint32(0), (ctypes.c_uint8 * 256)()
    L = lib()
    _check(_hooks(L).ss_debug_batch_classes(haystacks.data_ptr(), hb, count, ctypes.byref(st), cls), L)
    return st.value, (list(cls) if st.value == 3 else None)


def _ranges(off, begin, end):
    """(begin_ptr, end_ptr, count) from either CSR offsets (count+1 int64) or explicit begin/end arrays."""
    if off is not None:
        return off.data_ptr(), off.data_ptr() + 8, off.numel() - 1
    return begin.data_ptr(), end.data_ptr(), begin.numel()


def search_batched(haystacks, hay_off, needles, needle_off, position=None, stream=None, hay_ranges=None,
                   needle_ranges=None, pairs=False):
    """One launch for many (needle_i, haystack_i) problems; all arguments are device tensors (uint8
    blobs; int64 CSR offsets of length count+1, or explicit (begin, end) tensor pairs via *_ranges, which
    may alias).  ``pairs=True`` uses the lane-per-problem kernel for tiny haystacks (ss_search_pairs).
    Returns an int32 device tensor of flags."""
    import torch
    hb, he, count = _ranges(hay_off, *(hay_ranges or (None, None)))
    nb, ne, ncount = _ranges(needle_off, *(needle_ranges or (None, None)))
    assert count == ncount
    found = torch.empty(count, dtype=torch.int32, device=haystacks.device)
    st = stream if stream is not None else _current_stream_handle()
    fn = lib().ss_search_pairs if pairs else lib().ss_search_batched
    _check(fn(haystacks.data_ptr(), hb, he, needles.data_ptr(), nb, ne,
              position.data_ptr() if position is not None else None, count, st, found.data_ptr()))
    return found


def find_batched(haystacks, hay_off, needles, needle_off, stream=None, hay_ranges=None, needle_ranges=None):
    """Leftmost offset of needle i in haystack i for many problems in one call (ss_find_batched); -1 where absent.  Arguments as
    search_batched.  Returns an int64 device tensor."""
    import torch
    hb, he, count = _ranges(hay_off, *(hay_ranges or (None, None)))
    nb, ne, ncount = _ranges(needle_off, *(needle_ranges or (None, None)))
    assert count == ncount
    pos = torch.empty(count, dtype=torch.int64, device=haystacks.device)
    st = stream if stream is not None else _current_stream_handle()
    _check(lib().ss_find_batched(haystacks.data_ptr(), hb, he, needles.data_ptr(), nb, ne, count, st, pos.data_ptr()))
    return pos                                      # SS_NPOS (all ones) reads as -1


def search_file(searcher, path):
    """examples/grep.rs:42-56: map the file, one search_in (row f2)."""
    found = ctypes.c_int(0)
    _check(searcher._L.ss_search_file(searcher._h, os.fsencode(path), ctypes.byref(found)), searcher._L)
    return bool(found.value)


def byte_histogram(haystack, sample_bytes=0, stream=None):
    """256 byte-value counts of a device haystack (row f3)."""
    hist = np.zeros(256, dtype=np.uint64)
    st = stream if stream is not None else _current_stream_handle()
    _check(lib().ss_byte_histogram_device(haystack.data_ptr(), haystack.numel(), sample_bytes, st, hist.ctypes.data))
    return hist


def choose_position(needle, hist=None):
    """Index of the rarest needle byte under `hist` (ties: later byte); no histogram -> n-1 (row f3)."""
    nb = bytes(needle)
    pos = _sz(0)
    h = None if hist is None else np.ascontiguousarray(hist, dtype=np.uint64)
    _check(lib().ss_choose_position(nb, len(nb), None if h is None else h.ctypes.data, ctypes.byref(pos)))
    return pos.value


def choose_filter_triple(needle, hist=None):
    """(first, second, third) that `DynamicHipSearcher.new(needle)` lets the device filter test; with `hist` (256
    byte counts of the haystack, `byte_histogram`) the corpus-aware choice to apply with `set_filter`."""
    nb = bytes(needle)
    a, b, c = _sz(0), _sz(0), _sz(0)
    h = None if hist is None else np.ascontiguousarray(hist, dtype=np.uint64)
    _check(lib().ss_choose_filter_triple(nb, len(nb), None if h is None else h.ctypes.data, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    return a.value, b.value, c.value


def choose_filter_pair(needle):
    """(first, second): the first two of `choose_filter_triple(needle)`."""
    return choose_filter_triple(needle)[:2]


def choose_filter_for_position(needle, position):
    """(first, second, third) that `DynamicHipSearcher.with_position(needle, position)` lets the device filter test:
    `second == position`; `first == 0` (the reference's pair) when `position < 16`, else a byte at most 15 in front of it.
    Hooks builds (a pure host function; a constructed searcher's `filter3` says the same in every build)."""
    nb = bytes(needle)
    a, b, c = _sz(0), _sz(0), _sz(0)
    L = _hooks(lib())
    _check(L.ss_choose_filter_for_position(nb, len(nb), position, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), L)
    return a.value, b.value, c.value


def fill_random_device(tensor, seed, global_offset=0, stream=None):
    st = stream if stream is not None else _current_stream_handle()
    _check_tools(tools_lib().ss_fill_random_device(tensor.data_ptr(), global_offset, tensor.numel(), seed, st))
    return tensor


def fill_random_host(length, seed, global_offset=0):
    out = np.empty(length, dtype=np.uint8)
    _check_tools(tools_lib().ss_fill_random_host(out.ctypes.data, global_offset, length, seed))
    return out


def read_ceiling_gbps(tensor, reps=10, stream=None):
    ms = ctypes.c_float(0)
    st = stream if stream is not None else _current_stream_handle()
    _check_tools(tools_lib().ss_read_ceiling(tensor.data_ptr(), tensor.numel(), st, reps, ctypes.byref(ms)))
    return tensor.numel() / (ms.value * 1e-3) / 1e9


def device_info():
    name = ctypes.create_string_buffer(256)
    cus, mem = ctypes.c_int(0), _sz(0)
    _check(lib().ss_device_info(name, 256, ctypes.byref(cus), ctypes.byref(mem)))
    return {"name": name.value.decode(), "compute_units": cus.value, "total_mem": mem.value}


def selftest_dpp():
    """320 uint32 of the cross-lane self-test (ss_selftest_dpp; tests/test_gpu_parity.py::test_cross_lane_primitives)."""
    out = np.zeros(320, dtype=np.uint32)
    _check_tools(tools_lib().ss_selftest_dpp(out.ctypes.data))
    return out
