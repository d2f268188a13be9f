// src/hip_service.rs - the resident search service (include/sliceslice_hip_service.h): an OPT-IN component outside the hot path,
// gated by a feature of its own (`#[cfg(feature = "hip-service")] pub mod hip_service;`).  A crate built with that feature links
// libsliceslice_hip_service.so - the drop-in library's objects plus the service - INSTEAD of libsliceslice_hip.so (build.rs).
//
// SOURCE ONLY, like src/hip.rs: never compiled here (no rustc); the `extern "C"` block is checked mechanically against
// include/sliceslice_hip_service.h by tests/test_bindings_cpu.py.
#![allow(non_camel_case_types, dead_code)]
use crate::hip::{check, ss_searcher, DeviceSlice, DynamicHipSearcher};
use crate::Needle;
use std::os::raw::{c_int, c_void};

#[repr(C)] pub struct ss_service { _private: [u8; 0] }

extern "C" {
    pub fn ss_service_start(workgroups: c_int, lease_ms: f64, out: *mut *mut ss_service) -> c_int;
    pub fn ss_service_search(sv: *mut ss_service, s: *const ss_searcher, d_haystack: *const c_void, len: usize, found: *mut c_int) -> c_int;
    pub fn ss_service_bind(sv: *mut ss_service, d_haystack: *const c_void, len: usize) -> c_int;
    pub fn ss_service_stop(sv: *mut ss_service);
}

/// A resident search service on the current device (`ss_service_*`): a kernel that stays on the GPU and answers one
/// `search_in` at a time without a launch - 5 us per search instead of 8.5-9.5: the floor of the per-call shape (one PCIe
/// round trip).  The shape of the reference's own bench loop (bench/benches/i386.rs:246-256): build the searchers FIRST,
/// `bind` the text if it does not change between searches, then one `search_in` per needle.  The GPU's own answer to that
/// loop is ONE call for all needles: `ss_batch_plan_create` once (the searchers), `ss_batch_plan_run` per iteration.
pub struct SearchService { handle: *mut ss_service }

unsafe impl Send for SearchService {}
unsafe impl Sync for SearchService {}     // requests queue on a mutex inside the library

impl SearchService {
    /// `workgroups` = 0: 64; `lease_ms` = 0.0: 20 ms without a request, then the kernel leaves until the next one.
    pub fn start(workgroups: i32, lease_ms: f64) -> Self {
        let mut handle = std::ptr::null_mut();
        check(unsafe { ss_service_start(workgroups as c_int, lease_ms, &mut handle) });
        Self { handle }
    }
    /// The semantics of `DynamicHipSearcher::search_in_device` for a haystack that is COMPLETE in device memory.
    pub fn search_in<N: Needle>(&self, searcher: &DynamicHipSearcher<N>, haystack: DeviceSlice) -> bool {
        let mut found = 0;
        check(unsafe { ss_service_search(self.handle, searcher.handle(), haystack.ptr, haystack.len, &mut found) });
        found != 0
    }
    /// The caller vouches that `haystack` stays unchanged until `unbind` / the next `bind`.
    pub fn bind(&self, haystack: DeviceSlice) { check(unsafe { ss_service_bind(self.handle, haystack.ptr, haystack.len) }) }
    pub fn unbind(&self) { check(unsafe { ss_service_bind(self.handle, std::ptr::null(), 0) }) }
}

impl Drop for SearchService {
    fn drop(&mut self) { unsafe { ss_service_stop(self.handle) } }
}
