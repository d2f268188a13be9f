/*
 * sliceslice_hip.h - C ABI of the MI355X (gfx950) substring searcher: the reference-facing entry points.
 *
 * This is the drop-in boundary for ONE hot path of cloudflare/sliceslice-rs:
 * `DynamicAvx2Searcher::{new, with_position, search_in}` -> bool.  The reference
 * has no FFI on this path (its public surface is the Rust API); the entry points
 * below are what a Rust `extern "C"` block, a cgo / ctypes stub or a C++ caller
 * binds in order to replace that path (INTEGRATION.md shows the bindings).  The
 * style - pointer + length pairs, integer status - follows the reference's only
 * FFI precedent, bench/sse4-strstr/src/wrapper.h:7
 *     size_t avx2_strstr_v2(const char* s, size_t n, const char* needle, size_t k);
 * Benchmark / tuning helpers and the test hooks are in sliceslice_hip_tuning.h.
 *
 * All citations are paths under /root/reference (sliceslice-rs @ 2024_08_07).
 *
 * Semantics (bit-identical to DynamicAvx2Searcher, src/x86.rs:498-519, 356-361,
 * src/lib.rs:130-136):
 *     n == 0            -> found = 1 (even for an empty haystack)
 *     n == 1            -> found = (needle[0] occurs in haystack); empty -> 0
 *     len <  n          -> found = 0
 *     otherwise         -> found = exists i in [0, len-n]: hay[i..i+n] == needle
 * `position` selects the second filter byte and never changes the result
 * (src/lib.rs:375-378).
 *
 * Errors: the reference panics on contract violations (src/x86.rs:300,473);
 * nothing unwinds across this ABI - every function returns an ss_status.
 *
 * Threading: what a searcher SEARCHES FOR (needle, position, filter bytes) is
 * immutable after construction (ss_searcher_set_filter3 aside, which is refused
 * while a search runs); any number of threads may call ss_search_* on one handle
 * concurrently (each call uses its own flag slot).  What a handle does carry as
 * mutable state, per device, is LAUNCH TUNING for scans of 256 MiB and more: a
 * candidate census of the haystacks it has been used on - candidate counts, and
 * per needle position how many of the sampled candidates match there - decides
 * between four, five and six workgroups per CU and one or two tiles per workgroup
 * (ss_searcher_last_launch reports the choice), orders the second level's schedule, and moves the first-phase
 * bytes THE LIBRARY owns (all three for ss_searcher_new, never a caller's) to
 * positions that let fewer candidates through - or, where none get through, that
 * cost the first phase least - FOR THAT HAYSTACK
 * (ss_searcher_filter3 keeps reporting the searcher's own).  No result depends on
 * any of it; for a given haystack and needle the choices are the same once the
 * handle has settled (a dozen scans at most).  See "launch tuning" below:
 * ss_searcher_tuning_state reports all of it, ss_set_autotune(0) switches it off.
 * A call synchronises only the stream it was given.
 *
 * No CPU fallback exists: every ss_search_* that has to look at haystack bytes
 * launches a HIP kernel, and fails with SS_ERR_NO_DEVICE / SS_ERR_HIP otherwise.
 *
 * Environment (read once per process; everything else is an argument):
 *     SLICESLICE_SPIN_WAIT=0      wait for the stream instead of spinning on the pinned answer word
 *     SLICESLICE_NO_BAR_WRITES=1  never write device memory from the CPU (control blocks go by hipMemcpy; no service, no relay)
 *     SLICESLICE_RCCL_LIB=<path>  the RCCL library to dlopen instead of librccl.so.1 / librccl.so
 *     SLICESLICE_SET_THREADS=0    communicator sets issue their per-device work from the calling thread (SS_ISSUE_SERIAL)
 *     SLICESLICE_AUTOTUNE=0       launch tuning off (the initial value of ss_set_autotune): static choices, no sampling kernels
 */
#ifndef SLICESLICE_HIP_H
#define SLICESLICE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_API __attribute__((visibility("default")))

typedef enum ss_status {
    SS_OK = 0,
    SS_ERR_POSITION = 1,   /* reference panic: `assert!(position < needle.size())` x86.rs:300,
                              `assert_eq!(position, 0)` for one-byte needles x86.rs:473 */
    SS_ERR_ARGUMENT = 2,   /* NULL where a pointer is required, bad offsets, ... */
    SS_ERR_NO_DEVICE = 3,  /* no gfx950-class HIP device visible */
    SS_ERR_HIP = 4,        /* a HIP runtime call failed; see ss_last_error() */
    SS_ERR_RCCL = 5,       /* an RCCL call failed / librccl not loadable */
    SS_ERR_NOMEM = 6,
    SS_ERR_PEER = 7        /* collective searches (ss_search_sharded / ss_find_sharded): ANOTHER rank failed the local
                              part of this search; every rank has left the collective, none has an answer */
} ss_status;

/* Opaque searcher: owns a host copy and a device copy of the needle, `position`,
 * and the filter bytes - the counterpart of `DynamicAvx2Searcher<N>` (src/x86.rs:405-442)
 * and of the pre-splatted `VectorHash` (src/lib.rs:165-176). */
typedef struct ss_searcher ss_searcher;

/* DynamicAvx2Searcher::new (src/x86.rs:454-459): position = n - 1 (wrapping for n == 0).
 * Also places the needle and the searcher's control block on the CURRENT device (other devices: on first use there): a block
 * of a per-device slab, written by the CPU through the PCIe BAR - about 3 us, no HIP runtime call once a slab exists. */
SS_API int ss_searcher_new(const uint8_t *needle, size_t n, ss_searcher **out);

/* DynamicAvx2Searcher::with_position (src/x86.rs:468-493).
 *   n == 0: any position accepted (N0);  n == 1: position must be 0;  n >= 2: position < n.
 * Violations return SS_ERR_POSITION (the reference panics).  The needle bytes are copied; the
 * caller keeps ownership of its buffer (the reference copies n in 2..=16 too, x86.rs:476-490). */
SS_API int ss_searcher_with_position(const uint8_t *needle, size_t n, size_t position, ss_searcher **out);

/* No search of `s` may be running.  The synchronous entry points have finished when they return; work left behind by
 * ss_search_device_async / ss_find_device_async is waited for here (hipDeviceSynchronize on the devices it used). */
SS_API void ss_searcher_free(ss_searcher *s);

/* needle().len() and the `position` the searcher was built with (`new`: n - 1).  Either pointer may be NULL. */
SS_API int ss_searcher_info(const ss_searcher *s, size_t *needle_len, size_t *position);

/* The needle bytes the filter tests: needle[first], needle[second] (first <= second < n) and, when second - first <= 15,
 * a third byte needle[third] with first < third <= first + 15 (third == second: none).  The reference tests needle[0] and
 * needle[position] (src/x86.rs:297-316) and proves with its own tests that the result does not depend on `position`
 * (src/lib.rs:375-378); every further byte is one more necessary condition of a match, so it cannot change a result either.
 *   ss_searcher_with_position always tests the caller's byte needle[position].  Up to position 15 its partner is the
 *     reference's needle[0], plus the rarest other byte of needle[1..15] as the third; from position 16 the partner is a
 *     byte at most 15 in front of `position`, so that one 16-byte load serves all three.
 *   ss_searcher_new - whose caller did not choose - picks all three by a static rarity ranking of the needle's
 *     bytes (first byte + the two rarest of the 15 bytes behind it, over the first 1024 needle bytes), so that
 *     text-like haystacks rarely pass the filter.  ss_searcher_info still reports position n-1.  On a haystack of 256 MiB or more
 *     the library samples the haystack's byte histogram in front of the second scan and filters THAT haystack with the needle's
 *     rarest bytes under it when that promises 16 x fewer candidates (non-Latin UTF-8 text, padding patterns: bytes the static
 *     ranking takes for rare); ss_searcher_filter3 keeps reporting the static choice.
 *   ss_searcher_set_filter3 sets the triple verbatim (third == second: a plain two-byte filter, e.g. the reference's own
 *     pair (0, n-1)).  A pair 16 or more apart has no third byte and runs on the cross-lane kernels; beyond 16 * 63 bytes
 *     apart the device filters with `first` and two bytes close behind it and tests the caller's `second` first thing when a
 *     candidate reaches memory.  SS_ERR_POSITION if out of range; SS_ERR_ARGUMENT while any search is in flight on the
 *     searcher (the triple is only rewritten when nothing can be reading it). */
SS_API int ss_searcher_filter3(const ss_searcher *s, size_t *first, size_t *second, size_t *third);
SS_API int ss_searcher_set_filter3(ss_searcher *s, size_t first, size_t second, size_t third);

/* Row f3: a `position` policy.  The reference leaves `position` to the caller and defaults to the last byte
 * (src/x86.rs:252-255, 285).  ss_byte_histogram_device counts byte values of a device haystack (every
 * ceil(len/sample_bytes)-th 16-byte chunk; sample_bytes = 0 -> all) into hist[256] (host memory).
 * ss_choose_position returns the index (>= 1) of the needle byte that is rarest under `hist` - ties to the later byte;
 * hist == NULL -> n-1, the reference default - for ss_searcher_with_position.  ss_choose_filter_triple is the same choice for
 * all three filter bytes (what ss_searcher_new picks; hist == NULL: its static ranking, else cost of a byte = log2(count + 1),
 * so sums compare products of frequencies) for ss_searcher_set_filter3.  Pure host functions.  The result of a search never
 * depends on the choice (src/lib.rs:375-378); only how often candidates are verified does. */
SS_API int ss_byte_histogram_device(const void *d_haystack, size_t len, size_t sample_bytes, void *hip_stream,
                                    uint64_t hist[256]);
SS_API int ss_choose_position(const uint8_t *needle, size_t n, const uint64_t hist[256], size_t *position);
SS_API int ss_choose_filter_triple(const uint8_t *needle, size_t n, const uint64_t hist[256], size_t *first, size_t *second,
                                   size_t *third);

/* DynamicAvx2Searcher::search_in (src/x86.rs:523-525) on a haystack ALREADY RESIDENT in device
 * memory (any alignment, any length up to the device's memory).  Enqueues on `hip_stream`
 * (a hipStream_t; NULL = the default stream) and returns when the ANSWER has arrived, writing 0/1 to *found.  For scans of
 * up to ~20 ms the answer is a pinned word the scan's last workgroup (or a one-lane kernel behind the scan) stores and the
 * caller spins on - the stream itself may still be retiring the command for a few microseconds when the call returns (every
 * 256th call does wait for the stream); a caller that needs the stream idle - to destroy it, to read its own hipEvent
 * timings - synchronises it itself.  Longer scans, and SLICESLICE_SPIN_WAIT=0, wait for the stream.  ss_find_device: the same. */
SS_API int ss_search_device(const ss_searcher *s, const void *d_haystack, size_t len, void *hip_stream, int *found);

/* Same scan without the host round trip: ORs the result into the caller's device flag
 * (`*d_found`, int32, caller-zeroed) and returns after enqueueing.  This is the building block of
 * a range-sharded multi-GPU search with the caller's own collective (torch.distributed: searcher.py). */
SS_API int ss_search_device_async(const ss_searcher *s, const void *d_haystack, size_t len, void *hip_stream, int *d_found);

/* ---- row f1: position-returning find ------------------------------------------- */
/* Offset of the LEFTMOST occurrence, or SS_NPOS - the `Option<usize>` shape every competitor in the
 * reference's benches has (bench/sse4-strstr/src/lib.rs:4-15 `avx2_strstr_v2 -> Option<usize>`,
 * bench/benches/i386.rs:191-205) and of the reference tests' oracle `find_subsequence`
 * (tests/i386.rs:6-10).  Not part of DynamicAvx2Searcher itself, which only answers bool.
 * n == 0 -> 0.  Same kernels as search_in with the flag replaced by an atomicMin'ed uint64; a wave
 * only skips work that lies to the right of the best match so far. */
#define SS_NPOS UINT64_MAX
SS_API int ss_find_device(const ss_searcher *s, const void *d_haystack, size_t len, void *hip_stream, uint64_t *position);
/* The same for a HOST haystack (chunked upload, n-1 byte carry, stops issuing chunks after the first
 * chunk that reports a match). */
SS_API int ss_find_host(const ss_searcher *s, const uint8_t *haystack, size_t len, uint64_t *position);
/* Enqueue-only form: atomicMin's base_offset + local offset into *d_best (uint64 in device memory,
 * caller-initialised to SS_NPOS).  With base_offset = the shard's begin, an all-reduce(MIN) over the
 * ranks' d_best gives the global leftmost match of a range-sharded haystack. */
SS_API int ss_find_device_async(const ss_searcher *s, const void *d_haystack, size_t len, uint64_t base_offset,
                                void *hip_stream, uint64_t *d_best);

/* Drop-in form of search_in(&[u8]) for a HOST haystack: uploads the bytes in double-buffered chunks of up
 * to 64 MiB (needle_len-1 bytes of carry between chunks) and scans them on the device.  PCIe-bound by
 * construction; never used for roofline numbers.  The device buffers and streams come from a per-device
 * set that lives for the rest of the process and is lent to one call at a time (a concurrent call on the
 * same device builds and frees a private set), so a small haystack costs tens of microseconds per call.
 * Slices of up to 64 KiB take a shorter road (ss_find_host too): the CPU copies them into a pinned, device-visible
 * buffer of the calling thread and the kernel reads them over PCIe - no upload command (10-12 us per call instead of 16-24). */
SS_API int ss_search_host(const ss_searcher *s, const uint8_t *haystack, size_t len, int *found);

/* Row f2: the host-file front end of examples/grep.rs:42-56 (open the file, one
 * search_in) as a pipeline: reader threads pread() 64 MiB chunks into pinned buffers while earlier
 * chunks are uploading and being scanned on their own streams; n-1 bytes are carried across chunk
 * edges.  Bound by the file read / PCIe, never by the scan.  Uses the same cached per-device staging set
 * (plus pinned host buffers) as ss_search_host. */
SS_API int ss_search_file(const ss_searcher *s, const char *path, int *found);

/* Batched search, one launch (BASELINE.json config 5): problem i searches the needle
 * d_needles[needle_begin[i] .. needle_end[i]) in the haystack d_haystacks[hay_begin[i] .. hay_end[i]).
 * Everything lives in device memory; the four range arrays hold `count` uint64 each.  CSR callers
 * pass (off, off + 1); ranges may alias, e.g. 4,585 needles against ONE haystack - the loop of
 * bench/benches/i386.rs:252-256 as a single launch.  position[i] follows the with_position rules
 * (NULL = the `new` default n_i - 1) and is always one of the bytes the device tests; its partner bytes are picked on
 * the device by the rule of ss_searcher_with_position (coarser ranking).  Writes `count` int32 flags to d_found (device), each
 * with the semantics of ss_search_device for its problem; a problem whose position breaks those rules - where the
 * reference panics while building the searcher, src/x86.rs:300,473 - gets SS_BATCH_BAD_POSITION instead
 * (a device array cannot be validated on the host without a read-back).  One workgroup (or more) per
 * problem: meant for haystacks of KiBs to GiBs.  Not capturable into a hipGraph (SS_ERR_ARGUMENT on a capturing stream: the
 * call keeps per-stream scratch that a later call may reallocate) - a graph takes an ss_batch_plan. */
#define SS_BATCH_BAD_POSITION (-1)
SS_API int ss_search_batched(const void *d_haystacks, const uint64_t *d_hay_begin, const uint64_t *d_hay_end,
                             const void *d_needles, const uint64_t *d_needle_begin, const uint64_t *d_needle_end,
                             const uint64_t *d_position, size_t count, void *hip_stream, int *d_found);

/* Row f1 for a whole batch: the LEFTMOST offset of needle i in haystack i (SS_NPOS: absent; the empty needle: 0) - the
 * `Option<usize>` shape of bench/sse4-strstr/src/lib.rs:4-15 for many problems in one call.  Same ranges, same plan kernel and
 * scan grid as ss_search_batched (the `new` position for every problem); d_position: `count` uint64 in device memory.
 * Both calls remember a few batches per device (blob, range array, count): in front of the SECOND call that names a batch a byte
 * histogram of its haystacks is sampled on the call's stream (at most 4 MiB read, never waited for), and later calls choose the
 * needle bytes they filter on by it (row f3 of SURVEY.md 8f; no result depends on the choice). */
SS_API int ss_find_batched(const void *d_haystacks, const uint64_t *d_hay_begin, const uint64_t *d_hay_end,
                           const void *d_needles, const uint64_t *d_needle_begin, const uint64_t *d_needle_end,
                           size_t count, void *hip_stream, uint64_t *d_position);

/* Plan once, search many: the reference builds its 4,585 searchers ONCE and times only the searches
 * (bench/benches/i386.rs:246-256).  ss_batch_plan_create does for a batch what the constructors do for one needle: it reads
 * the range arrays and the NEEDLE bytes (on `hip_stream`, and waits for it), samples a byte histogram of the haystacks (at most
 * 4 MiB read) so that the needle bytes the scan filters on are the ones that are rare IN THESE HAYSTACKS (row f3 of SURVEY.md 8f;
 * a caller's `position` is kept; no result depends on the choice), and keeps one descriptor per problem in memory of its own;
 * ss_batch_plan_run is then ONE kernel launch - the scan, which writes the outputs itself - for every plan: no plan kernel, no
 * scratch acquire, nothing allocated, nothing initialised by the host, capturable into a hipGraph and replayable.  (Problems scanned
 * by several workgroups keep their state in the plan; consecutive runs tell themselves apart by the identity the hardware gives
 * every launch - the AQL dispatch id and the queue - so no second kernel is needed to re-arm anything: batched_kernels.hpp.)
 * `find` != 0: the plan answers leftmost offsets (d_out = `count` uint64, as ss_find_batched), else flags (d_out = `count` int32, as
 * ss_search_batched).  The caller vouches that ranges, needle bytes and the haystacks' ADDRESSES are unchanged between create and
 * the last run; haystack CONTENTS may change freely.  (A plan of long problems holds two launch layouts and picks by how many
 * problems an earlier run found - a tally the runs leave in pinned memory; a run never waits for it, no result depends on it, and
 * the choice is made by the HOST when the launch is issued: a run captured into a hipGraph keeps the layout it was captured with.)
 * ONE run at a time per plan (the plan's state words are the runs' scratch): runs must be ordered one behind the other - the same
 * stream, or events.  d_out needs no initialisation.  No run may be in flight when the plan is freed. */
typedef struct ss_batch_plan ss_batch_plan;
SS_API int ss_batch_plan_create(const void *d_haystacks, const uint64_t *d_hay_begin, const uint64_t *d_hay_end,
                                const void *d_needles, const uint64_t *d_needle_begin, const uint64_t *d_needle_end,
                                const uint64_t *d_position, size_t count, int find, void *hip_stream, ss_batch_plan **out);
SS_API int ss_batch_plan_run(const ss_batch_plan *plan, void *hip_stream, void *d_out);
SS_API void ss_batch_plan_free(ss_batch_plan *plan);

/* Same contract as ss_search_batched, one LANE per problem: the reference's short-haystack workload
 * (bench/benches/i386.rs:118-129: 10,513,405 word-in-word searches of <= 24 bytes).  Use it when the
 * haystacks are tens of bytes; any length is correct but long haystacks belong to ss_search_batched. */
SS_API int ss_search_pairs(const void *d_haystacks, const uint64_t *d_hay_begin, const uint64_t *d_hay_end,
                           const void *d_needles, const uint64_t *d_needle_begin, const uint64_t *d_needle_end,
                           const uint64_t *d_position, size_t count, void *hip_stream, int *d_found);

/* Kernel timing: when enabled, every scan launched through `s` is bracketed by hipEvents ON THE LAUNCH STREAM;
 * ss_searcher_last_kernel_ms returns the elapsed time of the CALLING THREAD's most recent completed scan through s
 * (milliseconds) - what a roofline figure for the scan kernel is computed from (bench.py). */
SS_API int ss_searcher_set_timing(ss_searcher *s, int enabled);
SS_API int ss_searcher_last_kernel_ms(const ss_searcher *s, float *ms);
/* Launch shape of the latest scan enqueued through `s` on the current device: workgroups per CU (4, 5 or 6: see "Threading" above)
 * and workgroups in the grid.  Read-only diagnostics - what a benchmark line reports next to its kernel times. */
SS_API int ss_searcher_last_launch(const ss_searcher *s, int *workgroups_per_cu, unsigned *grid);

/* ---- multi-GPU: one process per GPU, native RCCL ------------------------------------------- */
/* Range partition used by every sharded caller: rank r of G scans bytes
 * [r*S, min(len, (r+1)*S + n-1)) with S = ceil(len/G): an overlap of n-1 bytes, so a match that
 * straddles a boundary is seen by exactly the left rank. */
SS_API int ss_shard_range(size_t len, size_t needle_len, int nranks, int rank, size_t *begin, size_t *end);

/* The found flag of a range-sharded search is combined by ONE ncclAllReduce(int32, ncclMax)
 * (OR over {0,1} == MAX; RCCL has no bitwise-OR op).  librccl is dlopen()ed on first use. */
typedef struct ss_comm ss_comm;
#define SS_UNIQUE_ID_BYTES 128
SS_API int ss_comm_unique_id(uint8_t id[SS_UNIQUE_ID_BYTES]);                       /* rank 0, then broadcast */
SS_API int ss_comm_init_rank(const uint8_t id[SS_UNIQUE_ID_BYTES], int nranks, int rank, ss_comm **out);
SS_API void ss_comm_free(ss_comm *c);
SS_API int ss_comm_count(const ss_comm *c, int *nranks);                            /* ncclCommCount: what RCCL itself sees */
/* Which collective library this process' communicators are made of: the file the resolved ncclAllReduce lives in (dladdr) and what
 * its ncclGetVersion says (e.g. 22606; 0: the library has none).  A process that has torch in it holds torch's bundled librccl next
 * to the system's; a benchmark line names the one in use (bench.py: config.librccl_path / librccl_version). */
SS_API int ss_comm_rccl_info(char *path, size_t path_cap, int *version);
/* Scan + all-reduce + read-back on one stream: the whole sharded search_in.  The communicator's flag is
 * never cleared - "found" is the call's epoch (every rank makes the same sequence of calls on a communicator,
 * so the epochs agree) - which saves the memset launch per search.  Collective: every rank must call it, and a rank whose
 * local part fails still takes part (it returns its own error, every other rank SS_ERR_PEER; the next search finds all
 * ranks in step).  One search at a time per communicator: a second concurrent call is refused with SS_ERR_ARGUMENT. */
SS_API int ss_search_sharded(const ss_searcher *s, const void *d_shard, size_t shard_len, ss_comm *c,
                             void *hip_stream, int *found);
/* The same for the leftmost offset: ss_find_device_async with base_offset = shard_begin, then ONE
 * ncclAllReduce(uint64, ncclMin); *position = SS_NPOS when no rank has a match. */
SS_API int ss_find_sharded(const ss_searcher *s, const void *d_shard, size_t shard_len, uint64_t shard_begin,
                           ss_comm *c, void *hip_stream, uint64_t *position);

/* ---- multi-GPU inside ONE process (ncclCommInitAll) --------------------------------------------------- */
/* The form a drop-in `search_in(&self, haystack) -> bool` (src/x86.rs:523) over all GPUs of a node needs: no
 * launcher, no rendezvous.  ss_comm_init_all creates one communicator, one stream, one flag and one pinned
 * mirror per device (devs == NULL: devices 0 .. ndev-1).  ss_search_sharded_all scans shard g (resident on
 * device g of the set, ranges from ss_shard_range) on device g's stream, combines the flags with the G
 * ncclAllReduce(int32, ncclMax) calls inside ONE ncclGroupStart/End, reads the result back from device 0,
 * drains every stream and restores the caller's current device.  ss_find_sharded_all does the same for the
 * leftmost offset (uint64, ncclMin; shard_begins[g] = global offset of shard g).
 * SS_COMBINE_HOST skips the collective: the host ORs the G pinned mirrors (possible only in this
 * single-process form; the difference between the two is the cost of the collective).
 * ss_search_sharded_all ends every device's scan at the first match on ANY device: the host relays the finding device's flag
 * into the others' through their PCIe BARs while it waits (needs CPU-visible device memory, else each device runs its scan
 * to the end).  One search at a time per set (the set's streams and flags are its scratch): a second concurrent call is
 * refused with SS_ERR_ARGUMENT.
 * Who issues the per-device work: SS_ISSUE_THREADS (the default for sets of two or more devices) - the set keeps one thread per
 * device, parked on that device; each enqueues its device's scan, its ncclAllReduce (RCCL's one-thread-per-communicator form, no
 * group) and the answer word, so the G chains start side by side.  SS_ISSUE_SERIAL - everything from the calling thread, the
 * all-reduces as one group (SLICESLICE_SET_THREADS=0 makes it the default).  ss_find_sharded_all is issued the same way.
 * Diagnostics (read-only; what a benchmark line reports): ss_comm_set_count = ncclCommCount of EVERY communicator of the set
 * (SS_ERR_RCCL if they disagree); ss_comm_set_last_kernel_ms = every device's scan-kernel time of the latest search (needs
 * ss_searcher_set_timing on its searcher; ms[count >= devices]); ss_comm_set_last_issue_us = host time the latest search spent
 * issuing {scans, collective, answer words / read-back, all of it} - per device maxima under SS_ISSUE_THREADS, sums otherwise. */
typedef struct ss_comm_set ss_comm_set;
#define SS_COMBINE_RCCL 0
#define SS_COMBINE_HOST 1
#define SS_ISSUE_THREADS 0
#define SS_ISSUE_SERIAL 1
SS_API int ss_comm_init_all(int ndev, const int *devs, ss_comm_set **out);
SS_API void ss_comm_set_free(ss_comm_set *set);
SS_API int ss_comm_set_combine(ss_comm_set *set, int combine);
SS_API int ss_comm_set_issue(ss_comm_set *set, int issue);
SS_API int ss_comm_set_count(const ss_comm_set *set, int *nranks);
SS_API int ss_comm_set_last_kernel_ms(ss_comm_set *set, float *ms, int count);
SS_API int ss_comm_set_last_issue_us(const ss_comm_set *set, float us[4]);
SS_API int ss_search_sharded_all(const ss_searcher *s, const void *const *d_shards, const size_t *shard_lens,
                                 ss_comm_set *set, int *found);
SS_API int ss_find_sharded_all(const ss_searcher *s, const void *const *d_shards, const size_t *shard_lens,
                               const uint64_t *shard_begins, ss_comm_set *set, uint64_t *position);

/* (The resident search service - ss_service_* - is NOT part of this library: it is an opt-in component outside the hot path,
 * built into a library of its own, libsliceslice_hip_service.so, and declared in sliceslice_hip_service.h.) */

/* ---- launch tuning: what it is, how to see it, how to switch it off ------------------------------------------------------------
 * No search RESULT depends on anything here (src/lib.rs:375-378: the reference asserts the same for every `position`).  What a
 * handle learns about the haystacks it meets only moves necessary conditions around and picks launch shapes:
 *   - per (searcher, device, haystack >= 256 MiB): a candidate census of 1,024 sampled 4 KiB tiles, taken by a small kernel in
 *     front of the SECOND scan of the pair (the first only leaves its name) and every 256th after it, on the scan's own stream, never waited for: workgroups per CU (four /
 *     five / six) and 16 KiB tiles per workgroup (one / two), the cross-lane kernels with or without a third byte, the THIRD first-phase byte where the library owns it and the near
 *     bytes that stand in for a far pair (by the measured number of candidates each position lets through, on trial against the
 *     next census), and the ORDER of the second level's schedule (the needle byte that kills most of the sampled candidates
 *     first);
 *   - per (device, haystack): a sampled byte histogram (searchers built by ss_searcher_new: the filter bytes themselves);
 *   - per (device, batch) and per plan: the rarity classes of the haystacks' sampled bytes; a plan's layout by what its previous
 *     run found;
 *   - per searcher: whether its latest synchronous search found the needle.
 * ss_set_autotune(0) (or SLICESLICE_AUTOTUNE=0 in the environment) switches ALL of it off, process-wide: the constructors' static
 * filter bytes, a needle-byte guess for workgroups per CU, the static rarity classes, one plan layout (plans made while it is off),
 * no sampling kernels - a call's cost then depends on its arguments alone.  Returns the previous setting.
 * ss_searcher_tuning_state reports, without launching or waiting for anything, every tuning state the handle holds for one
 * haystack on the current device. */
typedef struct ss_tuning_state {
    uint32_t autotune;              /* 1: on (the default) */
    uint32_t census_state;          /* 0: no census of this haystack, 3: seen once (the census is taken in front of the SECOND scan), 1: in flight, 2: counts are in */
    uint32_t census_age;            /* scans that have gone by these counts (taken again every 256) */
    uint32_t tiles, tiles3, tiles2, match_tiles, lanes;   /* sampled tiles; with a candidate of the triple / of the pair alone / with a
                                       prefix match of up to 64 bytes; candidate lanes */
    uint32_t pair_lanes, triple_lanes;                    /* sampled candidates the per-position match counts were taken from */
    uint32_t deep_lanes;            /* ... of them: candidates that only the compare in memory can tell from a match */
    uint32_t triple_state;          /* the filter bytes on this haystack: 0 still being looked at, 1 the searcher's own, 2 in_force[] differs */
    uint32_t on_trial, trials;      /* a proposal's census is in flight; proposals put on trial so far ... */
    uint32_t accepted, settled;     /* ... and how many of them replaced the bytes in force; 1: no byte is being looked at any more */
    uint32_t proposal;              /* the latest proposal: 1 from the haystack's histogram, 2 one byte moved by the census's match counts,
                                       3 the near form of a pair 16 or more apart, 4 a jump to the needle byte that kills most of the sampled
                                       candidates, 5 the compact form (all three bytes within eight) of a filter that meets no candidates */
    uint32_t own[3], in_force[3];   /* needle indices of the three first-phase bytes: the searcher's own / on this haystack */
    uint32_t order_measured, norder;/* the second level's schedule is ordered by the census (else by the static rarity table) */
    uint8_t order[16];              /* ... needle indices, first tested first (norder of them) */
    uint32_t histogram_state;       /* the device's sampled histogram of this haystack: 0 none, 1 in flight, 2 in */
    uint32_t workgroups_per_cu, grid, kernel_mode, last_found;   /* the handle's latest launch on this device (any haystack) */
} ss_tuning_state;
SS_API int ss_set_autotune(int enabled);
SS_API int ss_searcher_tuning_state(const ss_searcher *s, const void *d_haystack, size_t len, ss_tuning_state *out);

/* Diagnostics */
SS_API const char *ss_last_error(void);      /* thread-local, static storage */
SS_API int ss_device_info(char *name, size_t name_cap, int *compute_units, size_t *total_mem);

#ifdef __cplusplus
}
#endif
#endif /* SLICESLICE_HIP_H */
