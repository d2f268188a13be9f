/* sliceslice_hip_service.h - the resident search service: an OPT-IN component outside SURVEY.md section 8, shipped in a library of its
 * own.
 *
 *   libsliceslice_hip.so          the drop-in library: every function of sliceslice_hip.h, no service code (no resident kernel,
 *                                 none of its mailbox / BAR / HDP-flush machinery)
 *   libsliceslice_hip_service.so  the same objects PLUS the service (sliceslice-rs_amd/csrc/ss_service.hip): every function of
 *                                 sliceslice_hip.h and the four below
 *
 * A process uses ONE of the two: a program that wants the service links libsliceslice_hip_service.so INSTEAD of
 * libsliceslice_hip.so (searchers belong to the library that made them - the two keep separate control-block pools and upload
 * tickets - so a handle from one must never be passed to the other).
 */
#ifndef SLICESLICE_HIP_SERVICE_H
#define SLICESLICE_HIP_SERVICE_H

#include "sliceslice_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- resident search service ------------------------------------------------------------------------------------
 * The reference answers a search of a small haystack in tens of nanoseconds (README.md:38: 10.5 M word-in-word searches in
 * 79 ms; bench/benches/i386.rs:246-256: 4,585 searches of an 857 kB text in 35 ms = 7.7 us each); a kernel launch alone
 * costs ss_search_device 8-10 us.  A search service is a small kernel that STAYS on the device and takes requests from a
 * 256-byte mailbox in DEVICE memory that the host writes through the PCIe BAR: no launch, no dispatch, no completion signal -
 * the host writes the request, every wave of the service polls the mailbox in its own memory and scans its share (same kernels'
 * code), the answer arrives in a pinned word the caller spins on: 5.0-5.7 us per search (launch: 8.5-9.5).  That is the floor
 * of the per-call shape - one PCIe round trip per search; the GPU's answer to the reference's
 * many-needles loop is the batched call.  Needs CPU-visible device memory (large BAR: every MI300-class part;
 * SS_ERR_NO_DEVICE otherwise).
 *   ss_service_start(workgroups, lease_ms, &sv)  on the current device; workgroups = 0: 64 (one per 4 compute units), at most
 *       one per compute unit; lease_ms = 0: 20 ms.  The kernel is resident only while requests keep coming: after
 *       `lease_ms` without one it leaves by itself (and is started again by the next request, at the price of one launch);
 *       under continuous traffic it leaves all the same once it has been resident for 16 leases (at least 250 ms), so that
 *       nothing that waits for the whole device - hipDeviceSynchronize, hipFree - waits longer than that.
 *   ss_service_search(sv, s, d_haystack, len, &found)  the semantics of ss_search_device, for searchers whose filter bytes
 *       lie within 16 bytes of each other (every constructor-built searcher; else SS_ERR_ARGUMENT).  The haystack must be
 *       COMPLETE in device memory: the service is not ordered behind work pending on any stream.  One request at a time per
 *       service (callers queue on a mutex); any haystack length is correct, a few MiB and less is what it is for.
 *   ss_service_bind(sv, d_haystack, len)  the caller vouches that [d_haystack, d_haystack + len) stays UNCHANGED until
 *       the next bind (len == 0: nothing bound) - the reference's bench shape: one text, thousands of needles.  A kernel that
 *       never ends sees no kernel boundary, so by default every request drops the caches' copy of whatever it is about to read
 *       (2 us of a request's 8); inside a bound range only the first request does, and so does any request whose searcher
 *       was uploaded to the device after the latest such acquire.  Writing to a bound range without re-binding: stale reads.
 *   ss_service_stop  asks the kernel to leave, waits for it and for every ss_service_search / _bind that has already
 *       entered, frees everything.  Calls that ARRIVE after ss_service_stop has been entered are the caller's bug (the
 *       handle is dead), exactly as with free(). */
typedef struct ss_service ss_service;
SS_API int ss_service_start(int workgroups, double lease_ms, ss_service **out);
SS_API int ss_service_search(ss_service *sv, const ss_searcher *s, const void *d_haystack, size_t len, int *found);
SS_API int ss_service_bind(ss_service *sv, const void *d_haystack, size_t len);
SS_API void ss_service_stop(ss_service *sv);

#ifdef __cplusplus
}
#endif
#endif /* SLICESLICE_HIP_SERVICE_H */
