/*
 * sliceslice_hip_tuning.h - NOT the drop-in boundary (that is sliceslice_hip.h): what the benchmark, the tuning tools and the
 * tests use around it.  Two groups:
 *
 *   1. Benchmark helpers, exported by libsliceslice_hip_tools.so (a separate small library: nothing here belongs in a
 *      product that replaces DynamicAvx2Searcher): the synthetic haystack generator of SURVEY.md 8d, the plain-read
 *      ceiling, the cross-lane self-test.
 *   2. Tuning knobs and test hooks, exported only by builds with -DSS_TEST_HOOKS (libsliceslice_hip_tuning.so and the
 *      sanitizer builds; the product library has neither the symbols nor the code behind them): kernel-variant and grid
 *      overrides, fault injection, epoch / counter setters, the pure filter-choice helper, service counters.  Hooks builds
 *      also read a few more environment variables (tuning: SLICESLICE_BATCH_WGS, _BATCH_MIN_TILES, _BATCH_OCC, _BATCH_STATIC_CLASSES, SLICESLICE_PLAN_ONE_LAYOUT;
 *      measurement: SLICESLICE_CROSS_EXIT=0, SLICESLICE_SERVICE_HDP_FLUSH=0, SLICESLICE_SERVICE_DEBUG).
 */
#ifndef SLICESLICE_HIP_TUNING_H
#define SLICESLICE_HIP_TUNING_H

#include "sliceslice_hip.h"
#include "sliceslice_hip_service.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- 1. libsliceslice_hip_tools.so ------------------------------------------------------------------------------- */

/* Synthetic haystack generator (not part of the reference):
 *   byte(i) = (splitmix64(splitmix64(seed) ^ (i >> 3)) >> (8 * (i & 7))) & 0xFF, then 0xFF -> 0x00,
 * with i = global_offset + k the GLOBAL byte index, so that range shards on different GPUs hold
 * slices of one logical haystack.  Device and host versions are bit-identical. */
SS_API int ss_fill_random_device(void *d_dst, uint64_t global_offset, size_t len, uint64_t seed, void *hip_stream);
SS_API int ss_fill_random_host(uint8_t *dst, uint64_t global_offset, size_t len, uint64_t seed);

/* Plain streaming read of `len` bytes (sum-reduced so it cannot be elided): the empirical
 * "achievable HBM read" ceiling printed next to the scan's GB/s.  ms = kernel time by hipEvents. */
SS_API int ss_read_ceiling(const void *d_src, size_t len, void *hip_stream, int reps, float *ms_per_rep);

/* Device self-test of the cross-lane primitives the scan relies on (DPP wave_shl:1, v_alignbyte):
 * fills out[0..320) (host memory); see tests/test_gpu_parity.py::test_cross_lane_primitives. */
SS_API int ss_selftest_dpp(uint32_t *out);

SS_API const char *ss_tools_last_error(void);

/* ---- 2. -DSS_TEST_HOOKS builds only ------------------------------------------------------------------------------ */

SS_API const char *ss_version(void);      /* names the build: "... tuning build: every kernel variant, test hooks" */

/* Kernel-variant override (the product library holds the kernels the constructors and ss_searcher_set_filter3 can select -
 * U = 4, non-temporal loads, the 8-byte phase for one-byte needles; the tuning build has every combination):
 * variant = 1000*LAYOUT + 100*MODE + 10*U + NT; LAYOUT 0 = automatic, 1 = 16 bytes per lane, 2 = 8-bytes-per-lane first
 * phase (position < 16 only); U in {4,8} pieces (KiB) per wave per tile; NT in {0,1} (plain / non-temporal loads);
 * MODE 0 = automatic, 2 = cross-lane position flags (filter pairs 16 or more apart only).  Two more decimal digits
 * on top are launch-shape experiments: + 10000*OCC (at most OCC workgroups per CU) + 100000*B (workgroup
 * size: 1 = 128, 2 = 256, 3 = 512 threads).  See DESIGN.md "Kernels". */
SS_API int ss_searcher_set_variant(ss_searcher *s, int variant);
/* Grid override: blocks > 0 = that many persistent workgroups (grid-stride over tiles); blocks < 0 =
 * -blocks tiles per short-lived workgroup; 0 = automatic. */
SS_API int ss_searcher_set_grid(ss_searcher *s, int blocks);

/* What ss_searcher_with_position would pick (pure host function): `second` == position always; position < 16: first == 0
 * (the reference's pair, x86.rs:297-316) plus the rarest other byte of needle[1..15] as `third`; position >= 16: `first` is a
 * byte at most 15 in front of `position` and `third` one of the 15 behind `first`.  SS_ERR_POSITION as the constructor. */
SS_API int ss_choose_filter_for_position(const uint8_t *needle, size_t n, size_t position, size_t *first, size_t *second,
                                         size_t *third);

/* Set the "found"-epoch counters (of every flag slot of `s` on the current device / of a
 * communicator or communicator set) so that a test can cross the 2^31 wrap. */
SS_API int ss_debug_set_epochs(ss_searcher *s, int value);
/* ... and the completion-word state of every slot of `s` on the current device, device counters and host copies alike:
 * the never-reset count of workgroups (the library starts a slot over before 2^31), the count of workgroups that found
 * the needle (wraps at 2^32) and the decreasing key of find()'s minimum (starts over at 0). */
SS_API int ss_debug_set_completion_state(ss_searcher *s, uint32_t workgroups, uint32_t found_workgroups, uint32_t find_key);
SS_API int ss_debug_set_comm_epoch(ss_comm *c, ss_comm_set *set, int value);
/* How many sharded searches of this process got their answer words only after the spin on them had run out of its budget (a
 * collective that took longer than the scan's estimate: the stream wait / drain-and-re-read path). */
SS_API uint64_t ss_debug_late_answers(void);
/* ... and make the next `count` scans launched through `s` fail before they reach the device (SS_ERR_HIP): how the tests
 * check that a rank-local failure leaves no other rank waiting in the collective of ss_search_sharded / ss_find_sharded. */
SS_API int ss_debug_fail_next_scans(ss_searcher *s, int count);
/* The candidate census of (`s`, d_haystack, len) on the current device, if its counts are in: counts[0] = wave-tiles sampled (0:
 * no census yet), [1] = tiles with a candidate of the device's three filter bytes, [2] = tiles with a candidate of the first two
 * alone, [3] = tiles in which a candidate's first 64 bytes equal the needle's, [4] = candidate lanes; and [5] = the kernel family of
 * the searcher's latest launch on the device (0 single stream, 2 / 3 cross-lane with / without the third byte); [6..8] = the three
 * filter bytes the device tests on THIS haystack (the searcher's own, or the triple chosen from the haystack's histogram); [9] = 0 not
 * decided / 1 the searcher's own / 2 the histogram's; [10] = histogram triples put on trial so far.  Launches nothing. */
SS_API int ss_debug_census(const ss_searcher *s, const void *d_haystack, size_t len, uint32_t counts[11]);
/* ... and its per-position match counters (aux_kernels.hpp, census_kernel): stats[k] = sampled PAIR candidates that match the needle
 * at position k (< 64), stats[64 + k] = sampled TRIPLE candidates that do, stats[128] / [129] = how many of each were sampled, stats[130] = the deep ones among the
 * triple candidates (match everywhere from the first filter byte on, yet no match).
 * *have = 0 while they are not in.  (What the product reports of it: ss_searcher_tuning_state.) */
SS_API int ss_debug_census_stats(const ss_searcher *s, const void *d_haystack, size_t len, uint32_t stats[131], int *have);

/* What the library remembers of an UNPLANNED batch (ss_search_batched / ss_find_batched with these haystacks, this range array and
 * this count) on the current device: *state = 0 unknown, 1 named once, 2 sampling in flight, 3 classes in - and then cls[0..256) =
 * the sixteen rarity classes (0 = rarest) its calls choose their filter bytes by. */
SS_API int ss_debug_batch_classes(const void *d_haystacks, const uint64_t *d_hay_begin, size_t count, uint32_t *state, uint8_t cls[256]);

/* What a plan's descriptor of `problem` filters on: out[0..2] = the indices in the needle of the three first-phase bytes (first <=
 * the other two), out[3] = the bytes themselves (first | second << 8 | third << 16 | one-byte needle << 24), out[4] = the slices
 * that scan the problem (0: answered without a scan; the indices are 0 then).  Copies 64 bytes from the device. */
SS_API int ss_debug_plan_filter(const ss_batch_plan *p, size_t problem, uint32_t out[5]);

/* A plan's layouts: out[0] = 1 if it holds two (plans of long problems), out[1] = slices per problem of the first (round robin where
 * more than eight), out[2] = of the second (contiguous runs), out[3] = problems found in the latest run whose tally has arrived,
 * out[4] = 1 if the next ss_batch_plan_run takes the second layout. */
SS_API int ss_debug_plan_layout(const ss_batch_plan *p, uint32_t out[5]);

/* A plan's ready-made cold part of `problem` (batched_kernels.hpp, BatchCold): out[0] = bytes in the second-level schedule, out[1] =
 * exact_len (bytes of the in-register compare | bytes in front of the first filter byte << 8; 0: the needle is too long for it),
 * out[2..5] = the schedule's indices (relative to the first filter byte), one byte each, low dword first; out[6..9] = the needle's
 * bytes at those indices; out[10..13] = the needle's dwords for the compare. */
SS_API int ss_debug_plan_cold(const ss_batch_plan *p, size_t problem, uint32_t out[14]);

/* requests served / kernel launches so far (a burst of requests shares one residency) / requests that skipped the acquire */
SS_API int ss_service_counters(ss_service *sv, uint64_t *requests, uint64_t *kernel_launches, uint64_t *settled);

#ifdef __cplusplus
}
#endif
#endif /* SLICESLICE_HIP_TUNING_H */
