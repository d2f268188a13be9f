// src/hip.rs  (source only: this image has no rustc; the C ABI below is the tested contract)
use crate::Needle;
use std::os::raw::{c_int, c_void};

#[repr(C)] pub struct ss_searcher { _private: [u8; 0] }

#[link(name = "sliceslice_hip")]
extern "C" {
    fn ss_searcher_new(needle: *const u8, n: usize, out: *mut *mut ss_searcher) -> c_int;
    fn ss_searcher_with_position(needle: *const u8, n: usize, position: usize,
                                 out: *mut *mut ss_searcher) -> c_int;
    fn ss_searcher_free(s: *mut ss_searcher);
    fn ss_search_host(s: *const ss_searcher, hay: *const u8, len: usize, found: *mut c_int) -> c_int;
    fn ss_search_device(s: *const ss_searcher, d_hay: *const c_void, len: usize,
                        hip_stream: *mut c_void, found: *mut c_int) -> c_int;
    fn ss_last_error() -> *const std::os::raw::c_char;
}

const SS_ERR_POSITION: c_int = 1;

/// Haystack already resident in device memory (caller-owned `hipMalloc` memory).
#[derive(Clone, Copy)]
pub struct DeviceSlice { pub ptr: *const c_void, pub len: usize }

/// GPU counterpart of `x86::DynamicAvx2Searcher<N>`; keeps `needle: N` by value like the reference.
pub struct DynamicHipSearcher<N: Needle> { handle: *mut ss_searcher, needle: N }

unsafe impl<N: Needle + Send> Send for DynamicHipSearcher<N> {}
unsafe impl<N: Needle + Sync> Sync for DynamicHipSearcher<N> {}   // ss_search_* is re-entrant per handle

fn check(rc: c_int) {
    if rc == 0 { return; }
    let msg = unsafe { std::ffi::CStr::from_ptr(ss_last_error()) }.to_string_lossy().into_owned();
    // contract violations panic exactly where the reference does (x86.rs:300, :473)
    if rc == SS_ERR_POSITION { panic!("{}", msg) } else { panic!("sliceslice_hip error {}: {}", rc, msg) }
}

impl<N: Needle> DynamicHipSearcher<N> {
    /// `position` defaults to the last byte (x86.rs:454-459).
    pub fn new(needle: N) -> Self {
        let position = needle.as_bytes().len().wrapping_sub(1);
        Self::with_position(needle, position)
    }
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
    pub fn needle(&self) -> &N { &self.needle }
}

impl<N: Needle> Drop for DynamicHipSearcher<N> {
    fn drop(&mut self) { unsafe { ss_searcher_free(self.handle) } }
}

// and in the crate's generic test suite (src/lib.rs:383-420):
//   impl crate::tests::TestSearcher for DynamicHipSearcher<&[u8]> { ... }
//   crate::generate_tests!(dynamic_hip_searcher, DynamicHipSearcher);
