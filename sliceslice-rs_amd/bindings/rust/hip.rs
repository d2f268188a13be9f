// src/hip.rs - the backend module a maintainer of cloudflare/sliceslice-rs would add next to src/x86.rs,
// gated like the others in src/lib.rs:11-24 (`#[cfg(feature = "hip")] pub mod hip;`).
//
// SOURCE ONLY: this image has no rustc, so the file has never been compiled.  The `extern "C"` block is
// checked mechanically against include/sliceslice_hip.h (names, arity, pointer / integer widths) by
// tests/test_bindings_cpu.py; the C ABI itself is what the GPU tests exercise.
//
// Mirrors `x86::DynamicAvx2Searcher<N>` member for member (src/x86.rs:405-525): the searcher owns
// `needle: N` by value for any `N: Needle` - `&[u8]`, `[u8; K]`, `Box<[u8]>`, `Vec<u8>`, `Rc<[u8]>`,
// `Arc<[u8]>` (src/lib.rs:43-104) - while the library keeps its own host + device copy of the bytes, the
// way `DynamicAvx2Searcher` keeps a private `[u8; n]` for n in 2..=16 (src/x86.rs:476-490).
#![allow(non_camel_case_types, dead_code)]
use crate::Needle;
use std::os::raw::{c_char, c_float, c_int, c_uint, c_void};
// (the resident search service - an opt-in component outside the hot path, in a library of its own - is src/hip_service.rs)

#[repr(C)] pub struct ss_searcher { _private: [u8; 0] }
#[repr(C)] pub struct ss_comm { _private: [u8; 0] }
#[repr(C)] pub struct ss_comm_set { _private: [u8; 0] }
#[repr(C)] pub struct ss_batch_plan { _private: [u8; 0] }
/// `ss_tuning_state` (include/sliceslice_hip.h): every launch-tuning state a handle holds for one haystack.  No result depends on it.
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct ss_tuning_state {
    pub autotune: u32, pub census_state: u32, pub census_age: u32,
    pub tiles: u32, pub tiles3: u32, pub tiles2: u32, pub match_tiles: u32, pub lanes: u32,
    pub pair_lanes: u32, pub triple_lanes: u32, pub deep_lanes: u32,
    pub triple_state: u32, pub on_trial: u32, pub trials: u32, pub accepted: u32, pub settled: u32, pub proposal: u32,
    pub own: [u32; 3], pub in_force: [u32; 3],
    pub order_measured: u32, pub norder: u32, pub order: [u8; 16],
    pub histogram_state: u32,
    pub workgroups_per_cu: u32, pub grid: u32, pub kernel_mode: u32, pub last_found: u32,
}

pub const SS_OK: c_int = 0;
pub const SS_ERR_POSITION: c_int = 1;
pub const SS_ERR_ARGUMENT: c_int = 2;
pub const SS_ERR_NO_DEVICE: c_int = 3;
pub const SS_ERR_HIP: c_int = 4;
pub const SS_ERR_RCCL: c_int = 5;
pub const SS_ERR_NOMEM: c_int = 6;
pub const SS_ERR_PEER: c_int = 7;
pub const SS_NPOS: u64 = u64::MAX;
pub const SS_UNIQUE_ID_BYTES: usize = 128;
pub const SS_COMBINE_RCCL: c_int = 0;
pub const SS_COMBINE_HOST: c_int = 1;
pub const SS_BATCH_BAD_POSITION: c_int = -1;

#[link(name = "sliceslice_hip")]
extern "C" {
    // construction (src/x86.rs:454-493)
    pub fn ss_searcher_new(needle: *const u8, n: usize, out: *mut *mut ss_searcher) -> c_int;
    pub fn ss_searcher_with_position(needle: *const u8, n: usize, position: usize, out: *mut *mut ss_searcher) -> c_int;
    pub fn ss_searcher_free(s: *mut ss_searcher);
    pub fn ss_searcher_info(s: *const ss_searcher, needle_len: *mut usize, position: *mut usize) -> c_int;
    pub fn ss_searcher_filter3(s: *const ss_searcher, first: *mut usize, second: *mut usize, third: *mut usize) -> c_int;
    pub fn ss_searcher_set_filter3(s: *mut ss_searcher, first: usize, second: usize, third: usize) -> c_int;
    // position policy (the reference leaves `position` to its caller, src/x86.rs:252-255)
    pub fn ss_byte_histogram_device(d_haystack: *const c_void, len: usize, sample_bytes: usize, hip_stream: *mut c_void, hist: *mut u64) -> c_int;
    pub fn ss_choose_position(needle: *const u8, n: usize, hist: *const u64, position: *mut usize) -> c_int;
    pub fn ss_choose_filter_triple(needle: *const u8, n: usize, hist: *const u64, first: *mut usize, second: *mut usize, third: *mut usize) -> c_int;
    // search_in (src/x86.rs:498-525)
    pub fn ss_search_device(s: *const ss_searcher, d_haystack: *const c_void, len: usize, hip_stream: *mut c_void, found: *mut c_int) -> c_int;
    pub fn ss_search_device_async(s: *const ss_searcher, d_haystack: *const c_void, len: usize, hip_stream: *mut c_void, d_found: *mut c_int) -> c_int;
    pub fn ss_search_host(s: *const ss_searcher, haystack: *const u8, len: usize, found: *mut c_int) -> c_int;
    pub fn ss_search_file(s: *const ss_searcher, path: *const c_char, found: *mut c_int) -> c_int;
    // find: the Option<usize> shape of the bench competitors (bench/sse4-strstr/src/lib.rs:4-15)
    pub fn ss_find_device(s: *const ss_searcher, d_haystack: *const c_void, len: usize, hip_stream: *mut c_void, position: *mut u64) -> c_int;
    pub fn ss_find_host(s: *const ss_searcher, haystack: *const u8, len: usize, position: *mut u64) -> c_int;
    pub fn ss_find_device_async(s: *const ss_searcher, d_haystack: *const c_void, len: usize, base_offset: u64, hip_stream: *mut c_void, d_best: *mut u64) -> c_int;
    // many problems, one launch
    pub fn ss_search_batched(d_haystacks: *const c_void, d_hay_begin: *const u64, d_hay_end: *const u64, d_needles: *const c_void,
                             d_needle_begin: *const u64, d_needle_end: *const u64, d_position: *const u64, count: usize,
                             hip_stream: *mut c_void, d_found: *mut c_int) -> c_int;
    pub fn ss_find_batched(d_haystacks: *const c_void, d_hay_begin: *const u64, d_hay_end: *const u64, d_needles: *const c_void,
                           d_needle_begin: *const u64, d_needle_end: *const u64, count: usize, hip_stream: *mut c_void, d_position: *mut u64) -> c_int;
    pub fn ss_batch_plan_create(d_haystacks: *const c_void, d_hay_begin: *const u64, d_hay_end: *const u64, d_needles: *const c_void,
                                d_needle_begin: *const u64, d_needle_end: *const u64, d_position: *const u64, count: usize, find: c_int,
                                hip_stream: *mut c_void, out: *mut *mut ss_batch_plan) -> c_int;
    pub fn ss_batch_plan_run(plan: *const ss_batch_plan, hip_stream: *mut c_void, d_out: *mut c_void) -> c_int;
    pub fn ss_batch_plan_free(plan: *mut ss_batch_plan);
    pub fn ss_search_pairs(d_haystacks: *const c_void, d_hay_begin: *const u64, d_hay_end: *const u64, d_needles: *const c_void,
                           d_needle_begin: *const u64, d_needle_end: *const u64, d_position: *const u64, count: usize,
                           hip_stream: *mut c_void, d_found: *mut c_int) -> c_int;
    // kernel timing (what a roofline figure is computed from)
    pub fn ss_searcher_set_timing(s: *mut ss_searcher, enabled: c_int) -> c_int;
    pub fn ss_searcher_last_kernel_ms(s: *const ss_searcher, ms: *mut c_float) -> c_int;
    pub fn ss_searcher_last_launch(s: *const ss_searcher, workgroups_per_cu: *mut c_int, grid: *mut c_uint) -> c_int;
    // multi-GPU, one process per GPU
    pub fn ss_shard_range(len: usize, needle_len: usize, nranks: c_int, rank: c_int, begin: *mut usize, end: *mut usize) -> c_int;
    pub fn ss_comm_unique_id(id: *mut u8) -> c_int;
    pub fn ss_comm_init_rank(id: *const u8, nranks: c_int, rank: c_int, out: *mut *mut ss_comm) -> c_int;
    pub fn ss_comm_free(c: *mut ss_comm);
    pub fn ss_comm_count(c: *const ss_comm, nranks: *mut c_int) -> c_int;
    pub fn ss_comm_rccl_info(path: *mut c_char, path_cap: usize, version: *mut c_int) -> c_int;
    pub fn ss_search_sharded(s: *const ss_searcher, d_shard: *const c_void, shard_len: usize, c: *mut ss_comm, hip_stream: *mut c_void, found: *mut c_int) -> c_int;
    pub fn ss_find_sharded(s: *const ss_searcher, d_shard: *const c_void, shard_len: usize, shard_begin: u64, c: *mut ss_comm,
                           hip_stream: *mut c_void, position: *mut u64) -> c_int;
    // multi-GPU inside one process
    pub fn ss_comm_init_all(ndev: c_int, devs: *const c_int, out: *mut *mut ss_comm_set) -> c_int;
    pub fn ss_comm_set_free(set: *mut ss_comm_set);
    pub fn ss_comm_set_combine(set: *mut ss_comm_set, combine: c_int) -> c_int;
    pub fn ss_comm_set_issue(set: *mut ss_comm_set, issue: c_int) -> c_int;
    pub fn ss_comm_set_count(set: *const ss_comm_set, nranks: *mut c_int) -> c_int;
    pub fn ss_comm_set_last_kernel_ms(set: *mut ss_comm_set, ms: *mut c_float, count: c_int) -> c_int;
    pub fn ss_comm_set_last_issue_us(set: *const ss_comm_set, us: *mut c_float) -> c_int;
    pub fn ss_search_sharded_all(s: *const ss_searcher, d_shards: *const *const c_void, shard_lens: *const usize, set: *mut ss_comm_set, found: *mut c_int) -> c_int;
    pub fn ss_find_sharded_all(s: *const ss_searcher, d_shards: *const *const c_void, shard_lens: *const usize, shard_begins: *const u64,
                               set: *mut ss_comm_set, position: *mut u64) -> c_int;
    // launch tuning: the switch and the read-only report
    pub fn ss_set_autotune(enabled: c_int) -> c_int;
    pub fn ss_searcher_tuning_state(s: *const ss_searcher, d_haystack: *const c_void, len: usize, out: *mut ss_tuning_state) -> c_int;
    // diagnostics
    pub fn ss_last_error() -> *const c_char;
    pub fn ss_device_info(name: *mut c_char, name_cap: usize, compute_units: *mut c_int, total_mem: *mut usize) -> c_int;
}

/// Haystack already resident in device memory (caller-owned `hipMalloc` memory).
#[derive(Clone, Copy)]
pub struct DeviceSlice { pub ptr: *const c_void, pub len: usize }

/// GPU counterpart of `x86::DynamicAvx2Searcher<N>`; keeps `needle: N` by value like the reference.
pub struct DynamicHipSearcher<N: Needle> { handle: *mut ss_searcher, needle: N }

unsafe impl<N: Needle + Send> Send for DynamicHipSearcher<N> {}
unsafe impl<N: Needle + Sync> Sync for DynamicHipSearcher<N> {}   // ss_search_* is re-entrant per handle

pub(crate) fn check(rc: c_int) {
    if rc == SS_OK { return; }
    let msg = unsafe { std::ffi::CStr::from_ptr(ss_last_error()) }.to_string_lossy().into_owned();
    // contract violations panic exactly where the reference does (x86.rs:300, :473)
    if rc == SS_ERR_POSITION { panic!("{}", msg) } else { panic!("sliceslice_hip error {}: {}", rc, msg) }
}

impl<N: Needle> DynamicHipSearcher<N> {
    /// x86.rs:454-459.  `ss_searcher_new` (not `with_position(len - 1)`): the caller did not choose a
    /// position, so the library may pick both filter bytes; `position()` still reports `len - 1`.
    pub fn new(needle: N) -> Self {
        let b = needle.as_bytes();
        let mut handle = std::ptr::null_mut();
        check(unsafe { ss_searcher_new(b.as_ptr(), b.len(), &mut handle) });
        Self { handle, needle }
    }
    /// x86.rs:468-493; panics like the reference for `position >= len` (and `position != 0` for one byte).
    /// `needle[position]` is always one of the bytes the device filter tests; its partner is `needle[0]` up to
    /// position 15 and a byte at most 15 in front of `position` beyond (`filter()`), which cannot change a result.
    pub fn with_position(needle: N, position: usize) -> Self {
        let b = needle.as_bytes();
        let mut handle = std::ptr::null_mut();
        check(unsafe { ss_searcher_with_position(b.as_ptr(), b.len(), position, &mut handle) });
        Self { handle, needle }
    }
    #[inline]
    pub fn inlined_search_in(&self, haystack: &[u8]) -> bool {
        let mut found = 0;
        check(unsafe { ss_search_host(self.handle, haystack.as_ptr(), haystack.len(), &mut found) });
        found != 0
    }
    pub fn search_in(&self, haystack: &[u8]) -> bool { self.inlined_search_in(haystack) }
    /// The HBM-roofline path: haystack already on the device.
    pub fn search_in_device(&self, haystack: DeviceSlice, stream: *mut c_void) -> bool {
        let mut found = 0;
        check(unsafe { ss_search_device(self.handle, haystack.ptr, haystack.len, stream, &mut found) });
        found != 0
    }
    /// Leftmost occurrence - the `Option<usize>` of `find_subsequence` (tests/i386.rs:6-10).
    pub fn find(&self, haystack: &[u8]) -> Option<usize> {
        let mut pos = SS_NPOS;
        check(unsafe { ss_find_host(self.handle, haystack.as_ptr(), haystack.len(), &mut pos) });
        if pos == SS_NPOS { None } else { Some(pos as usize) }
    }
    pub fn find_device(&self, haystack: DeviceSlice, stream: *mut c_void) -> Option<usize> {
        let mut pos = SS_NPOS;
        check(unsafe { ss_find_device(self.handle, haystack.ptr, haystack.len, stream, &mut pos) });
        if pos == SS_NPOS { None } else { Some(pos as usize) }
    }
    pub fn position(&self) -> usize {
        let mut p = 0usize;
        check(unsafe { ss_searcher_info(self.handle, std::ptr::null_mut(), &mut p) });
        p
    }
    pub fn needle(&self) -> &N { &self.needle }
    pub fn handle(&self) -> *const ss_searcher { self.handle }
}

impl<N: Needle> Drop for DynamicHipSearcher<N> {
    fn drop(&mut self) { unsafe { ss_searcher_free(self.handle) } }
}

/// Counterpart of `x86::Avx2Searcher<N>` (src/x86.rs:266-382): needles of at least one byte; an empty needle panics
/// like the reference (`assert!(position < size)`, src/x86.rs:300; test `avx2_empty_needle`).
pub struct HipSearcher<N: Needle>(DynamicHipSearcher<N>);

impl<N: Needle> HipSearcher<N> {
    pub fn new(needle: N) -> Self {
        assert!(!needle.as_bytes().is_empty());
        Self(DynamicHipSearcher::new(needle))
    }
    pub fn with_position(needle: N, position: usize) -> Self {
        assert!(position < needle.as_bytes().len());
        Self(DynamicHipSearcher::with_position(needle, position))
    }
    #[inline]
    pub fn inlined_search_in(&self, haystack: &[u8]) -> bool { self.0.inlined_search_in(haystack) }
    pub fn search_in(&self, haystack: &[u8]) -> bool { self.0.search_in(haystack) }
}

/// Counterpart of `MemchrSearcher` (src/lib.rs:119-142).
pub struct MemchrHipSearcher(DynamicHipSearcher<[u8; 1]>);

impl MemchrHipSearcher {
    pub fn new(needle: u8) -> Self { Self(DynamicHipSearcher::new([needle])) }
    #[inline]
    pub fn inlined_search_in(&self, haystack: &[u8]) -> bool { self.0.inlined_search_in(haystack) }
    pub fn search_in(&self, haystack: &[u8]) -> bool { self.0.search_in(haystack) }
}

/// All GPUs of the node behind ONE `search_in`: the haystack is range-partitioned into one shard per
/// device (n-1 bytes of overlap, `ss_shard_range`), resident in that device's HBM; a search is one scan per
/// device plus one grouped all-reduce(MAX) of the found flag (`ss_search_sharded_all`).
pub struct NodeHaystack { pub shards: Vec<DeviceSlice>, pub begins: Vec<u64> }

pub struct NodeSearcher<N: Needle> { inner: DynamicHipSearcher<N>, set: *mut ss_comm_set, ndev: usize }

impl<N: Needle> NodeSearcher<N> {
    pub fn new(needle: N, ndev: usize) -> Self {
        // the searcher first: if its constructor panics there is no communicator set yet that could leak
        let inner = DynamicHipSearcher::new(needle);
        let mut set = std::ptr::null_mut();
        check(unsafe { ss_comm_init_all(ndev as c_int, std::ptr::null(), &mut set) });
        Self { inner, set, ndev }
    }
    /// Byte range of shard `g` of a haystack of `len` bytes.
    pub fn shard_range(&self, len: usize, g: usize) -> (usize, usize) {
        let (mut b, mut e) = (0usize, 0usize);
        let n = self.inner.needle().as_bytes().len();
        check(unsafe { ss_shard_range(len, n, self.ndev as c_int, g as c_int, &mut b, &mut e) });
        (b, e)
    }
    /// `ncclCommCount` of every communicator of the set (they must agree).
    pub fn rccl_ranks(&self) -> usize {
        let mut n = 0;
        check(unsafe { ss_comm_set_count(self.set, &mut n) });
        n as usize
    }
    pub fn search_in(&self, haystack: &NodeHaystack) -> bool {
        assert_eq!(haystack.shards.len(), self.ndev);
        let ptrs: Vec<*const c_void> = haystack.shards.iter().map(|s| s.ptr).collect();
        let lens: Vec<usize> = haystack.shards.iter().map(|s| s.len).collect();
        let mut found = 0;
        check(unsafe { ss_search_sharded_all(self.inner.handle(), ptrs.as_ptr(), lens.as_ptr(), self.set, &mut found) });
        found != 0
    }
    pub fn find(&self, haystack: &NodeHaystack) -> Option<usize> {
        // ss_find_sharded_all reads ndev entries of each array: a shorter Vec would be an out-of-bounds read from safe code
        assert_eq!(haystack.shards.len(), self.ndev);
        assert_eq!(haystack.begins.len(), self.ndev);
        let ptrs: Vec<*const c_void> = haystack.shards.iter().map(|s| s.ptr).collect();
        let lens: Vec<usize> = haystack.shards.iter().map(|s| s.len).collect();
        let mut pos = SS_NPOS;
        check(unsafe { ss_find_sharded_all(self.inner.handle(), ptrs.as_ptr(), lens.as_ptr(), haystack.begins.as_ptr(), self.set, &mut pos) });
        if pos == SS_NPOS { None } else { Some(pos as usize) }
    }
}

impl<N: Needle> Drop for NodeSearcher<N> {
    fn drop(&mut self) { unsafe { ss_comm_set_free(self.set) } }
}

// Hook-up to the crate's generic test suite (src/lib.rs:383-420; the x86 back end does the same at
// src/x86.rs:589-611): every KAT of src/lib.rs:422-544 runs for every `position`.
#[cfg(test)]
mod tests {
    use super::DynamicHipSearcher;
    use crate::tests::TestSearcher;

    impl TestSearcher for DynamicHipSearcher<&[u8]> {
        fn with_position(needle: &'static [u8], position: usize) -> Self { DynamicHipSearcher::with_position(needle, position) }
        fn search_in(&self, haystack: &[u8]) -> bool { DynamicHipSearcher::search_in(self, haystack) }
    }

    crate::generate_tests!(dynamic_hip_searcher, DynamicHipSearcher);

    impl TestSearcher for super::HipSearcher<&[u8]> {
        fn with_position(needle: &'static [u8], position: usize) -> Self { super::HipSearcher::with_position(needle, position) }
        fn search_in(&self, haystack: &[u8]) -> bool { super::HipSearcher::search_in(self, haystack) }
    }

    #[test]
    #[should_panic]
    fn hip_empty_needle() { let _ = super::HipSearcher::new(Box::<[u8]>::from(&b""[..])); }               // x86.rs:545-549

    #[test]
    #[should_panic]
    fn dynamic_hip_invalid_position() { let _ = DynamicHipSearcher::with_position(b"foo".as_ref(), 3); }   // x86.rs:533-537

    #[test]
    fn owns_any_needle_form() {                                       // src/lib.rs:43-104
        let hay = b"Lorem ipsum dolor sit amet";
        assert!(DynamicHipSearcher::new(Box::<[u8]>::from(&b"ipsum"[..])).search_in(hay));
        assert!(DynamicHipSearcher::new(b"ipsum".to_vec()).search_in(hay));
        assert!(DynamicHipSearcher::new(std::sync::Arc::<[u8]>::from(&b"ipsum"[..])).search_in(hay));
        assert!(DynamicHipSearcher::new(*b"ipsum").search_in(hay));
    }
}
