// batched_kernels.hpp - K4: many (needle, haystack) problems in one grid (ss_search_batched / ss_find_batched / ss_batch_plan_*),
// and the one-lane-per-problem kernel for short haystacks (ss_search_pairs).  Included by ss_batched.hip only.
#pragma once
#include "scan_kernels.hpp"

namespace ss {

// ---- K4: batched, one grid for many (needle, haystack) problems ----------------------------------
// blockIdx.x = problem, blockIdx.y = slice of that problem's tiles: the workgroups of slice 0 of every
// problem are dispatched before any of slice 1, so when the needles are present early (the reference's
// i386 loop: every word occurs in the text) the later slices find the flag set on entry and leave - the
// sequential scan's early exit survives the slicing.  Per-problem flags, no cross-problem early exit.  The problem descriptor is built per workgroup from the range arrays
// (begin[i], end[i]) - CSR callers pass (off, off + 1); ranges may alias (many needles, one haystack).
struct BatchArgs {
    const uint8_t *haystacks;
    const uint64_t *hay_begin, *hay_end;
    const uint8_t *needles;
    const uint64_t *needle_begin, *needle_end;
    const uint64_t *position;   // may be null: n_i - 1
    int *found;                 // search: one int32 flag per problem
    uint64_t *best;             // find (ss_find_batched): one uint64 leftmost offset per problem (all ones = absent); else null
};
constexpr uint32_t kPlanSliceMajorMax = 8;   // launches with at most this many slices per problem use the slice-major layout (scan_batched_plan_kernel)
constexpr int kBadPosition = -1;   // SS_BATCH_BAD_POSITION: flag of a problem whose position breaks the with_position rules

// ---- K4, planned form: a one-lane-per-problem plan kernel + the scan grid ------------------------------------
// The kernel above rebuilds its problem descriptor in every workgroup: ranges -> needle bytes -> first haystack load is a
// chain of three dependent memory round trips (3-4 us under load) in front of every slice, which is why it only does well
// when a slice is long (4,096 x 1 MiB in ~10-tile slices: 0.88-0.90 of the HBM peak; 1,024 x 1 MiB in 8-tile slices: 0.73).
// Here the descriptors are built ONCE per problem by batch_plan_kernel (one lane per problem; it also writes the initial
// flag, so it replaces the memset launch), 64 bytes each, and a scan workgroup starts with ONE scalar load
// (s_load_dwordx16 of its problem's descriptor, issued together with the entry poll of the problem's flag) before its first
// haystack load - one round trip more than scan_kernel, whose descriptor travels in the kernel arguments.  With the start-up
// chain gone, slices can be short (kPlanMinTiles) and the grid generous: surplus slices leave after that one scalar load.
struct __attribute__((aligned(64))) BatchDesc {
    const uint8_t *base;       // 16-byte-aligned start of the filter stream: hay + anchor - mis
    uint64_t end;              // candidate offsets (0: nothing to scan - trivial problem, answered by the plan kernel)
    uint64_t nchunks_all;
    uint64_t n;                // needle length
    uint64_t needle_off;       // offset of the needle in the needle blob
    uint64_t anchor;           // index of the first filter byte in the needle
    uint64_t per;              // active slices of the problem << 32 | tiles per slice (both < 2^32: the grid is one-dimensional)
    uint32_t bytes;            // needle[anchor] | second byte << 8 | third byte << 16 | (one-byte needle) << 24
    uint32_t shifts;           // mis | r << 4 | Q << 6 | r3 << 8 | q3 << 10
};
static_assert(sizeof(BatchDesc) == 64, "one scalar load (s_load_dwordx16) per workgroup");
// The cold part of a problem - what verification needs - where it is written ahead of the scan (see ColdInPlan / ColdInCall below).
struct __attribute__((aligned(64))) BatchCold {
    uint64_t order_idx[2], order_val[2];       // as Problem::order_idx / order_val (build_refine_order)
    uint32_t tail16[4];                        // as Problem::tail16
    uint32_t norder, exact_len;
    // The problem's STATE while a scan runs lives here too - a line of its own per pair of problems, not one of 16 or 32 words of
    // an array: every wave polls its problem's word once per tile, the resident workgroups of a problem-major launch belong to a
    // few dozen consecutive problems, and with their words in ONE cache line every match (an atomic on that line) sent the polls of
    // all of them to memory - 1,024 x 1 MiB with every needle present ran 0.23-0.29 ms where the full scan takes 0.155.
    //   unplanned bool calls: pad[1] = the found flag the waves poll and raise (the caller's output is written behind it)
    //   unplanned find calls: pad[0..1] = one uint64, the leftmost offset so far (the caller's output is lowered behind it)
    //   plans:                not here - a plan's problems have a PlanState of their own (two words, one per run parity: below)
    uint32_t pad[2];
};
static_assert(sizeof(BatchCold) == 64, "one scalar load");

__host__ __device__ constexpr inline int rarity_class4(uint8_t b)
{
    const int r = byte_rarity_rank(b);
    return r < 64 ? 0 : (r < 128 ? 1 : (r < 192 ? 2 : 3));
}
// The four classes as two bit planes of 256 bits each (8 dwords per plane): no table in memory, no branches - the plan kernel
// fills its LDS table from these constants.
struct ClassPlanes {
    uint32_t lo[8], hi[8];
};
constexpr ClassPlanes make_class_planes()
{
    ClassPlanes p = {};
    for (int b = 0; b < 256; ++b) {
        const int c = rarity_class4((uint8_t)b);
        if (c & 1) p.lo[b >> 5] |= 1u << (b & 31);
        if (c & 2) p.hi[b >> 5] |= 1u << (b & 31);
    }
    return p;
}

// ---- the rarity classes of a PLAN come from its haystacks' own bytes (row f3 of SURVEY.md 8f for config 5) -------------------
// The static classes above are a corpus-free guess (letters common, everything outside text rare) - right for English text and
// binaries, exactly wrong where the "rare-looking" bytes are the haystacks' most frequent ones (UTF-8 text in a non-Latin script:
// every other byte is 0xD0 / 0xD1).  ss_batch_plan_create therefore has batch_sample_kernel take a byte histogram of
// kPlanSampleTiles pieces of 4 KiB - sample j reads problem (j * count / kPlanSampleTiles) (or j mod count when there are fewer
// problems than samples) at a pseudo-random offset of its haystack, so aliased ranges (many needles, one text) are sampled all
// over the text and not kPlanSampleTiles times at its start - and the workgroup that finishes last turns the counts into SIXTEEN
// classes: the number of whole bits in total / count, 15 = every second byte and more, 0 = never seen (or rarer than 1 in
// 32,768).  Per-wave LDS histograms, one global atomic per non-zero counter and workgroup.  The unplanned calls do the same without
// ever waiting: the sampling goes in front of the SECOND call that names the same haystacks, later calls use its classes once
// they are in (ss_batched.hip).  Results never depend on the classes (lib.rs:375-378).
constexpr uint32_t kPlanSampleTiles = 1024, kPlanSampleBytes = 4096, kPlanSampleBlocks = 64;
// 16 classes from a sampled histogram: whole bits of total / count, rarest = 0.
__device__ __forceinline__ uint8_t class_from_count(uint32_t cnt, uint32_t total)
{
    if (cnt == 0) return 0;
    const uint32_t ratio = total / cnt;                                            // >= 1
    const uint32_t bits = 31u - (uint32_t)__builtin_clz(ratio);
    return (uint8_t)(15u - (bits < 15u ? bits : 15u));
}
// The sampling's memory: 256 counters and the count of finished workgroups (all zero between launches: the workgroup that
// completes the count puts them back), and the 256 classes it leaves behind.
struct BatchClasses {
    uint32_t hist[256];
    uint32_t done, pad[15];
    uint8_t cls[256];
};
// `h_tag` (may be null): a pinned word that takes `tag` when the classes are in place - how the unplanned calls, which never wait,
// learn that a sampling launched in front of an earlier call has finished (ss_batched.hip).
__global__ void __launch_bounds__(kBlock) batch_sample_kernel(const uint8_t *haystacks, const uint64_t *hay_begin, const uint64_t *hay_end,
                                                               uint64_t count, BatchClasses *out, unsigned long long *h_tag,
                                                               unsigned long long tag)
{
    __shared__ uint32_t h[kWavesPerBlock][256];
    __shared__ uint32_t s_total;
    __shared__ int s_last;
    for (int k = threadIdx.x; k < kWavesPerBlock * 256; k += kBlock) (&h[0][0])[k] = 0;
    if (threadIdx.x == 0) s_total = 0;
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    constexpr uint32_t kPerWave = kPlanSampleTiles / (kPlanSampleBlocks * kWavesPerBlock);
    for (uint32_t t = 0; t < kPerWave; ++t) {
        const uint64_t j = ((uint64_t)blockIdx.x * kWavesPerBlock + wave) * kPerWave + t;
        if (count < kPlanSampleTiles && j >= count * ((kPlanSampleTiles + count - 1) / count)) break;
        const uint64_t prob = count >= kPlanSampleTiles ? j * count / kPlanSampleTiles : j % count;
        const uint64_t h0 = hay_begin[prob], h1 = hay_end[prob];
        if (h1 <= h0) continue;
        const uint64_t len = h1 - h0;
        uint64_t off = 0, take = len;
        if (len > kPlanSampleBytes) {
            const uint64_t frac = ((uint32_t)j * 2654435761u) >> 8;                 // 24 pseudo-random bits per sample
            off = (uint64_t)(((unsigned __int128)(len - kPlanSampleBytes) * frac) >> 24);
            take = kPlanSampleBytes;
        } else if (count < kPlanSampleTiles && j >= count) {
            continue;                                                               // a short haystack is read once
        }
        const uint8_t *p = haystacks + h0 + off + (uint64_t)lane * 64;
        const uint64_t mine = (uint64_t)lane * 64 < take ? take - (uint64_t)lane * 64 : 0;
        if (mine >= 64) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint32_t w = reinterpret_cast<const UnalignedU32 *>(p + 4 * q)->v;
                atomicAdd(&h[wave][w & 0xFF], 1u);
                atomicAdd(&h[wave][(w >> 8) & 0xFF], 1u);
                atomicAdd(&h[wave][(w >> 16) & 0xFF], 1u);
                atomicAdd(&h[wave][w >> 24], 1u);
            }
        } else {
            for (uint64_t q = 0; q < mine; ++q) atomicAdd(&h[wave][p[q]], 1u);
        }
    }
    __syncthreads();
    const uint32_t part = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
    if (part) __hip_atomic_fetch_add(&out->hist[threadIdx.x], part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(&out->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == kPlanSampleBlocks;
    __syncthreads();
    if (!s_last) return;
    // the workgroup that completes the count: counts -> classes, and everything back to zero for the next launch
    __threadfence();
    const uint32_t mine = __hip_atomic_exchange(&out->hist[threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t sum = mine;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) atomicAdd(&s_total, sum);
    __syncthreads();
    out->cls[threadIdx.x] = class_from_count(mine, s_total);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&out->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (h_tag) __hip_atomic_store(h_tag, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
constexpr uint32_t kClassNone = 255;            // above every class of either table

// What the plan kernel tells the host about a plan's problems (ss_batch_plan_create sizes the grid of the runs from it).
struct PlanStats {
    uint32_t max_slices;       // the most active slices any problem got
    uint32_t max_tiles;        // the longest scan, in tiles (saturating)
    uint64_t total_tiles;
};

// One problem's descriptor (and, for the unplanned calls, its initial output); returns its number of active slices.
__device__ __forceinline__ uint32_t plan_one(const BatchArgs &a, uint64_t prob, uint64_t h0, uint64_t h1, uint64_t n0, uint64_t n1, uint64_t given,
                                             BatchDesc *descs, uint32_t nslices, uint32_t min_tiles, int tile_pieces, const uint8_t *s_class,
                                             uint64_t *tiles_out, bool free_pair, BatchCold *colds)
{
    *tiles_out = 0;
    const uint64_t len = h1 - h0, n = n1 - n0;
    const uint64_t position = (a.position && n) ? given : n - 1;
    BatchDesc d;
    d.base = nullptr;
    d.end = d.nchunks_all = 0;
    d.n = n;
    d.needle_off = n0;
    d.anchor = 0;
    d.per = 0;                                      // no active slice
    d.bytes = d.shifts = 0;
    BatchCold lite;                                 // (unplanned calls) the needle's dwords where they are at hand: see ColdInCall
    lite.order_idx[0] = lite.order_idx[1] = lite.order_val[0] = lite.order_val[1] = 0;
    lite.tail16[0] = lite.tail16[1] = lite.tail16[2] = lite.tail16[3] = 0;
    lite.norder = lite.exact_len = 0;
    lite.pad[0] = lite.pad[1] = 0;
    int flag = 0;
    if (n == 0) {
        flag = 1;                                   // N0: found everywhere (x86.rs:500)
    } else if (n == 1 ? position != 0 : position >= n) {
        flag = kBadPosition;                        // the reference panics building this searcher (x86.rs:300, 473)
    } else if (len >= n) {
        const uint8_t *needle = a.needles + n0;
        uint64_t anchor = 0;
        if (position >= 16) {
            uint32_t cls[15];
#pragma unroll
            for (int k = 0; k < 15; ++k) cls[k] = needle[position - 15 + k];
#pragma unroll
            for (int k = 0; k < 15; ++k) cls[k] = s_class[cls[k]];
            uint32_t best_cls = kClassNone;
#pragma unroll
            for (int k = 0; k < 15; ++k) {          // later bytes win ties: the partner closest to `position`
                const bool better = cls[k] <= best_cls;
                best_cls = better ? cls[k] : best_cls;
                anchor = better ? position - 15 + k : anchor;
            }
        }
        uint32_t s2 = (uint32_t)(position - anchor);            // distance between the two filter bytes: 0 .. 15
        const uint32_t lim = n - anchor < 16 ? (uint32_t)(n - anchor) : 16u;
        uint32_t fb[16], cls[16];
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) fb[k] = needle[anchor + (k < lim ? k : 0u)];
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) cls[k] = s_class[fb[k]];
        if (free_pair && anchor == 0 && position == n - 1) {
            // nobody chose `position` (it is the default, the last byte) and the classes are the haystacks' own: the partner of
            // needle[0] is the rarest of the 15 bytes behind it, as ss_searcher_new would have it, the later one among equals
            uint32_t bc = kClassNone;
#pragma unroll
            for (uint32_t k = 1; k < 16; ++k) {
                const bool better = k < lim && cls[k] <= bc;
                bc = better ? cls[k] : bc;
                s2 = better ? k : s2;
            }
        }
        uint32_t p3 = s2, best_cls = kClassNone;
#pragma unroll
        for (uint32_t k = 1; k < 16; ++k) {         // the rarest of the 15 bytes behind the anchor, later ones winning ties
            const bool better = k < lim && k != s2 && cls[k] <= best_cls && n - anchor >= 3;
            best_cls = better ? cls[k] : best_cls;
            p3 = better ? k : p3;
        }
        if (p3 / 4 > s2 / 4) {                      // the kernels want the third byte's dword not behind the second's
            const uint32_t t = p3;
            p3 = s2;
            s2 = t;
        }
        uint32_t b2 = 0, b3 = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {         // fb[s2], fb[p3] without a dynamic index (scratch)
            b2 = k == s2 ? fb[k] : b2;
            b3 = k == p3 ? fb[k] : b3;
        }
        if (anchor == 0 && n >= 2 && n <= 16) {
            // the whole needle sits in fb[0 .. n): the dwords of the in-register compare, no byte in front of the first filter byte.
            // (Measured against the lazy form in one process, profiles/r05/ab_call_cold.jsonl: the reference's i386 loop 0.135 ms a
            // call instead of 0.145; every second needle present, 16,384 x 64 KiB 0.168 instead of 0.207, 65,536 x 16 KiB 0.336
            // instead of 0.492; without matches the same.  Round-robin launches - few problems, two dozen workgroups each - gain
            // nothing from it, and lost 4-8 % as long as their state words shared cache lines: see BatchCold.)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                lite.tail16[j] = (4u * j + 0 < n ? fb[4 * j] : 0u) | ((4u * j + 1 < n ? fb[4 * j + 1] : 0u) << 8) |
                                 ((4u * j + 2 < n ? fb[4 * j + 2] : 0u) << 16) | ((4u * j + 3 < n ? fb[4 * j + 3] : 0u) << 24);
            lite.exact_len = (uint32_t)n;
        }
        const uint8_t *hf = a.haystacks + h0 + anchor;
        const uint32_t mis = (uint32_t)((uintptr_t)hf & 15);
        d.base = hf - mis;
        d.end = len - n + 1;
        d.nchunks_all = (mis + len - anchor + 15) / 16;
        d.anchor = anchor;
        d.bytes = fb[0] | (b2 << 8) | (b3 << 16) | (n == 1 ? 1u << 24 : 0u);
        d.shifts = mis | ((s2 % 4) << 4) | ((s2 / 4) << 6) | ((p3 % 4) << 8) | ((p3 / 4) << 10);
        const uint64_t npieces = ((mis + d.end + 15) / 16 + 63) / 64;
        const uint64_t ntiles = (npieces + tile_pieces - 1) / tile_pieces;
        uint64_t eff = (ntiles + min_tiles - 1) / min_tiles;
        eff = eff < nslices ? (eff ? eff : 1) : nslices;
        d.per = (eff << 32) | ((ntiles + eff - 1) / eff);
        *tiles_out = ntiles;
    }
    if (colds) {                                           // (unplanned calls: ColdInCall; the state word idle)
        lite.pad[0] = lite.pad[1] = a.best ? ~0u : 0u;
        colds[prob] = lite;
    }
    if (d.per == 0) d.shifts = (uint32_t)flag;             // no scan: the answer travels in the descriptor too (plan runs)
    if (a.best) a.best[prob] = n == 0 ? 0ull : ~0ull;      // the empty needle matches at offset 0 of every haystack
    else if (a.found) a.found[prob] = flag;
    descs[prob] = d;
    return (uint32_t)(d.per >> 32);
}

// One LANE per problem.  `nslices` = slices per problem of the scan launch that follows, `min_tiles` = the shortest slice worth a
// workgroup.  Same rules as scan_batched_kernel: needle[position] is always a first-phase byte; its partner is needle[0]
// when position < 16, else the rarest (class) byte of the 15 in front of it, closest to `position` among equals; the third
// byte is the rarest of the 15 behind the anchor, the later one among equals; the two are ordered by dword (q3 <= Q).
// Written for LATENCY - the scan cannot start before this kernel has ended: the rarity classes come from a 256-entry table
// in LDS (byte_rarity_rank is a dozen branches), and the needle bytes of a step are fetched by unconditional loads
// (out-of-range slots re-read byte 0 of the window) that are all in flight together; a first cut with a predicated
// load-rank loop ran 8-12 us, one memory round trip per byte.
// `stats` (plans only, else null): see PlanStats.
__global__ void __launch_bounds__(kBlock) batch_plan_kernel(const BatchArgs a, uint64_t count, BatchDesc *descs,
                                                             uint32_t nslices, uint32_t min_tiles, int tile_pieces, PlanStats *stats,
                                                             const uint8_t *cls, BatchCold *colds)
{
    __shared__ uint8_t s_class[256];
    __shared__ uint32_t s_max, s_maxt;
    __shared__ unsigned long long s_sum;
    if (threadIdx.x == 0) {
        s_max = s_maxt = 0;
        s_sum = 0;
    }
    if (cls) {                                                                 // (uniform: a kernel argument) the classes of the
        s_class[threadIdx.x] = cls[threadIdx.x];                               // haystacks' own sampled histogram: batch_sample_kernel
    } else {
        constexpr ClassPlanes P = make_class_planes();                         // compile-time constants, selected by wave
        const uint32_t t = threadIdx.x, w = t >> 5;                            // kBlock == 256: one table entry per thread
        uint32_t lo = P.lo[0], hi = P.hi[0];
#pragma unroll
        for (uint32_t k = 1; k < 8; ++k) {
            lo = w == k ? P.lo[k] : lo;
            hi = w == k ? P.hi[k] : hi;
        }
        s_class[t] = (uint8_t)(((lo >> (t & 31)) & 1u) | (((hi >> (t & 31)) & 1u) << 1));
    }
    const uint64_t prob = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = prob < count;
    const uint64_t pi = live ? prob : 0;                                       // (every lane reaches the barrier)
    const uint64_t h0 = a.hay_begin[pi], h1 = a.hay_end[pi];
    const uint64_t n0 = a.needle_begin[pi], n1 = a.needle_end[pi];
    const uint64_t given = a.position ? a.position[pi] : 0;
    __syncthreads();
    uint64_t tiles = 0;
    const bool free_pair = cls != nullptr && a.position == nullptr;
    if (stats) {                                                               // (uniform: a kernel argument)
        uint32_t eff = 0;
        if (live) eff = plan_one(a, pi, h0, h1, n0, n1, given, descs, nslices, min_tiles, tile_pieces, s_class, &tiles, free_pair, colds);
        if (eff != 0) {
            atomicMax(&s_max, eff);
            atomicMax(&s_maxt, tiles > 0xffffffffull ? 0xffffffffu : (uint32_t)tiles);
            atomicAdd(&s_sum, (unsigned long long)tiles);
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_max != 0) {
            atomicMax(&stats->max_slices, s_max);
            atomicMax(&stats->max_tiles, s_maxt);
            atomicAdd(reinterpret_cast<unsigned long long *>(&stats->total_tiles), s_sum);
        }
        return;
    }
    if (live) (void)plan_one(a, pi, h0, h1, n0, n1, given, descs, nslices, min_tiles, tile_pieces, s_class, &tiles, free_pair, colds);
}

// The cold fields of a planned problem, re-read from its descriptor by the waves that need them (scan_tiles' ColdT).
struct ColdFields {
    const uint8_t *hay, *needle;
    uint64_t n, end;
    uint32_t norder, exact_len;
    uint64_t order_idx[2], order_val[2];
    uint32_t tail16[4];
    int *host_flag;
    uint32_t *tally;                              // (ColdInPlan, plans that hold two layouts) the run's count of found problems; else null
    uint64_t far_off;
    uint32_t ready;                               // (ColdInCall) the record holds the needle's dwords: nothing to build
    __device__ __forceinline__ const ColdFields *operator->() const { return this; }
};
// A PLAN carries the cold part ready-made: what a wave of the unplanned kernel builds when it first meets a candidate - the
// second-level schedule (up to 15 further needle bytes, rarest first) and, for needles that end within 16 bytes of the first filter
// byte, the needle's dwords for the in-register compare - costs it a dependent round trip to the needle bytes plus a few hundred
// operations, once per wave and WORKGROUP: nothing on random bytes, where next to no wave meets a candidate; where the needles ARE
// there (the reference's bench: every word occurs in the text) it sits on the path of every problem's answer - 65,536 problems of
// 16 KiB, every second needle present: 0.447 ms a run, 0.271 with the cold part ready-made - and on text full of near misses it is
// paid by every other workgroup.  batch_cold_kernel (one LANE per problem, once per plan) writes one 64-byte BatchCold per
// problem; a wave then needs one more load.
// The unplanned calls' form: the plan kernel of a call writes a record too, but only what costs it nothing - for a needle of up to
// 16 bytes whose first filter byte is needle[0] (every needle of that length unless the caller chose a position of 16 or more) the
// needle's dwords are already in its registers: tail16, exact_len (non-zero says: usable as it is), an empty schedule.  A wave that meets a candidate looks
// there first and builds the cold part itself (scan_tiles, BUILD_ORDER) only when the record says it must.
struct ColdInCall {
    static constexpr bool kHasOrder = false;
    static constexpr bool kMaybeOrder = true;
    const BatchDesc *dp;
    const BatchCold *cp;
    const uint8_t *needles;
    void *out_word;                                 // the caller's output of this problem - int flag or uint64 offset: the wave that finds writes it
    __device__ __forceinline__ ColdFields operator()() const
    {
        const BatchDesc *q = dp;
        const BatchCold *c = cp;
        __asm__ volatile("" : "+s"(q), "+s"(c));    // opaque: the loads stay in the cold path
        ColdFields f;
        f.hay = q->base + (q->shifts & 15) - q->anchor;
        f.needle = needles + q->needle_off;
        f.n = q->n;
        f.end = q->end;
        f.norder = 0;
        f.exact_len = c->exact_len;
        f.order_idx[0] = f.order_idx[1] = f.order_val[0] = f.order_val[1] = 0;
        f.tail16[0] = c->tail16[0]; f.tail16[1] = c->tail16[1]; f.tail16[2] = c->tail16[2]; f.tail16[3] = c->tail16[3];
        f.host_flag = static_cast<int *>(out_word);
        f.tally = nullptr;
        f.far_off = 0;
        f.ready = c->exact_len;                      // (the plan kernel sets it only where it left the dwords)
        return f;
    }
};
template <bool MULTI>
struct ColdInPlanT {
    static constexpr bool kHasOrder = true;
    static constexpr bool kMaybeOrder = false;
    static constexpr bool kSingleLaunchPlan = MULTI;  // scan_tiles: state word first, the caller's output behind it, the tally
    const BatchDesc *dp;
    const BatchCold *cp;
    const uint8_t *needles;
    void *out_word;                                 // problems scanned by several workgroups: the caller's output of this problem; else null
    uint32_t *tally;
    __device__ __forceinline__ ColdFields operator()() const
    {
        const BatchDesc *q = dp;
        const BatchCold *c = cp;
        __asm__ volatile("" : "+s"(q), "+s"(c));    // opaque: the loads stay in the cold path
        ColdFields f;
        f.hay = q->base + (q->shifts & 15) - q->anchor;
        f.needle = needles + q->needle_off;
        f.n = q->n;
        f.end = q->end;
        f.norder = c->norder;
        f.exact_len = c->exact_len;
        f.order_idx[0] = c->order_idx[0]; f.order_idx[1] = c->order_idx[1];
        f.order_val[0] = c->order_val[0]; f.order_val[1] = c->order_val[1];
        f.tail16[0] = c->tail16[0]; f.tail16[1] = c->tail16[1]; f.tail16[2] = c->tail16[2]; f.tail16[3] = c->tail16[3];
        f.host_flag = MULTI ? static_cast<int *>(out_word) : nullptr;
        f.tally = MULTI ? tally : nullptr;
        f.far_off = 0;
        f.ready = 1;
        return f;
    }
};

// One LANE per problem, once per plan, behind the plan kernel: the cold part of every problem that is scanned.  The schedule follows
// build_refine_order's rules - the bytes 16..31 behind the first filter byte first, rarest first, at most kFarFirst of them; then
// the bytes 1..15, rarest first; then what is left of the far ones; fifteen in all, the first-phase bytes left out - with the
// rarity classes the plan's filter bytes were chosen by (`cls`: the haystacks' own, else the static four).  Only the ORDER of the
// checks depends on the classes; the dwords of the exact compare are the needle's bytes.
__global__ void __launch_bounds__(kBlock) batch_cold_kernel(const BatchArgs a, const BatchDesc *__restrict__ descs, uint64_t count,
                                                             BatchCold *colds, const uint8_t *cls, int find)
{
    __shared__ uint8_t s_class[256];
    if (cls) {
        s_class[threadIdx.x] = cls[threadIdx.x];
    } else {
        s_class[threadIdx.x] = (uint8_t)rarity_class4((uint8_t)threadIdx.x);
    }
    __syncthreads();
    const uint64_t prob = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (prob >= count) return;
    const BatchDesc d = descs[prob];
    BatchCold c;
    c.order_idx[0] = c.order_idx[1] = c.order_val[0] = c.order_val[1] = 0;
    c.tail16[0] = c.tail16[1] = c.tail16[2] = c.tail16[3] = 0;
    c.norder = c.exact_len = 0;
    c.pad[0] = c.pad[1] = find ? ~0u : 0u;          // the plan's state word of this problem, idle
    if ((d.per >> 32) != 0 && ((d.bytes >> 24) & 1) == 0) {
        const uint8_t *needle = a.needles + d.needle_off + d.anchor;         // from the first filter byte on
        const uint64_t nrel = d.n - d.anchor;
        const uint32_t lim = nrel < (uint64_t)kRefineWindow ? (uint32_t)nrel : (uint32_t)kRefineWindow;
        const uint32_t pos2 = 4 * ((d.shifts >> 6) & 3) + ((d.shifts >> 4) & 3), pos3 = 4 * ((d.shifts >> 10) & 3) + ((d.shifts >> 8) & 3);
        uint32_t b[kRefineWindow], k[kRefineWindow];
#pragma unroll
        for (int K = 0; K < kRefineWindow; ++K) b[K] = needle[(uint32_t)K < lim ? K : 0];     // (all in flight together)
#pragma unroll
        for (int K = 0; K < kRefineWindow; ++K) k[K] = s_class[b[K]];
        uint32_t valid = 0;                         // bit K: a byte the schedule may use
#pragma unroll
        for (int K = 1; K < kRefineWindow; ++K)
            if ((uint32_t)K < lim && (uint32_t)K != pos2 && (uint32_t)K != pos3) valid |= 1u << K;
        uint64_t i0 = 0, i1 = 0, v0 = 0, v1 = 0;
        uint32_t m = 0, used = 0, far_used = 0;
        auto emit = [&](uint32_t K, uint32_t byte) {
            const uint32_t sh = 8 * (m & 7);
            if (m < 8) { i0 |= (uint64_t)K << sh; v0 |= (uint64_t)byte << sh; }
            else { i1 |= (uint64_t)K << sh; v1 |= (uint64_t)byte << sh; }
            ++m;
            used |= 1u << K;
        };
        // three passes, each class by class (rarest = 0 first), stable in K
#pragma unroll 1
        for (uint32_t cl = 0; cl < 16; ++cl) {
#pragma unroll
            for (int K = 16; K < kRefineWindow; ++K)
                if (((valid >> K) & 1u) && k[K] == cl && far_used < kFarFirst) { emit(K, b[K]); ++far_used; }
        }
#pragma unroll 1
        for (uint32_t cl = 0; cl < 16; ++cl) {
#pragma unroll
            for (int K = 1; K < 16; ++K)
                if (((valid >> K) & 1u) && k[K] == cl && m < 15) emit(K, b[K]);
        }
#pragma unroll 1
        for (uint32_t cl = 0; cl < 16; ++cl) {
#pragma unroll
            for (int K = 16; K < kRefineWindow; ++K)
                if (((valid >> K) & 1u) && !((used >> K) & 1u) && k[K] == cl && m < 15) emit(K, b[K]);
        }
        c.norder = m;
        c.order_idx[0] = i0; c.order_idx[1] = i1;
        c.order_val[0] = v0; c.order_val[1] = v1;
        if (nrel <= 16) {
            // the bytes from the first filter byte on, plus what sixteen leave room for of those in front of it (scan_tiles)
            const uint32_t behind = (uint32_t)nrel;
            const uint32_t back = (uint32_t)(d.anchor < 16 - behind ? d.anchor : 16 - behind);
            const uint32_t el = behind + back;
            c.exact_len = el | (back << 8);
            uint32_t t[16];
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) t[j] = (needle - back)[j < el ? j : 0];
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) t[j] = j < el ? t[j] : 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) c.tail16[j] = t[4 * j] | (t[4 * j + 1] << 8) | (t[4 * j + 2] << 16) | (t[4 * j + 3] << 24);
        }
    }
    colds[prob] = c;
}

// Grid: ONE dimension, nslices workgroups per problem; two ways of laying them out, chosen by the host from the slice count
// (the lengths live on the device; the count of problems is all the host knows):
//   * many problems, few slices each (nslices <= kPlanSliceMajorMax): SLICE-MAJOR, w = slice * count + problem, each slice a
//     contiguous run of the problem's tiles.  All slice-0 workgroups are dispatched before any slice-1 workgroup, so a needle
//     that is present early (the reference's i386 loop: every word occurs in the text, most of them in the first tiles) has
//     set its flag by the time the later slices of its problem start, and those leave at their entry poll - problem-major
//     layouts start all slices of a problem together and ran that loop at 0.21-0.45 ms instead of 0.15.
//   * few problems, many slices each: PROBLEM-MAJOR, w = problem * nslices + slice, and the active slices take the problem's
//     tiles ROUND ROBIN (slice s scans tiles s, s + eff, ...): the workgroups of a problem move through its haystack side
//     by side - consecutive addresses in flight, where slice-major puts 1,024 separate streams a haystack apart in flight
//     (1,024 x 1 MiB: 150 us instead of 162, kernel time) - and when one of them finds the needle the others are at the same
//     depth and stop at their next poll.
// FIND: the sink is the problem's uint64 (leftmost offset, atomicMin); a workgroup skips only what lies right of the best so far
// (scan_tiles does that tile by tile, so the slice-major entry poll is not needed).
// PLAN (ss_batch_plan_run: descriptors built once, searched many times - the reference builds its searchers once and times the
// searches, bench/benches/i386.rs:246-256): a run produces the caller's outputs itself - nothing is initialised by the host or by
// a kernel in front of the scan, so a run can be replayed from a hipGraph.  What shapes it: a workgroup of a batch lives for a
// few tiles, and ANY memory round trip at its end - a returning atomic, a load of the problem's flag - is time its slot on the CU
// stands idle.  Counting the workgroups of a problem out (so that the last one publishes) was built first and is gone again: with
// the counters of 32 problems in one 128-byte line the device-scope atomics queued line by line (1,024 x 1 MiB: 0.31 ms a run
// instead of 0.15); with a line per problem and a second level for problems of many workgroups the wait for ONE returning atomic
// per workgroup still cost a plan of few long problems 5-8 % (one haystack of 1 GiB: 164 us of kernel time instead of 152), and the
// workgroups that leave early because the needle has been found still had to count (the reference's 4,585-needle loop: 0.190 ms
// instead of 0.178).  So:
//   * a problem scanned by ONE workgroup (eff == 1): a match goes to a word in the workgroup's LDS (scan_tiles' wg_sink) and the
//     workgroup publishes the answer with one store - no state word, no atomic, no second kernel;
//   * a problem scanned by several: they work on the plan's own state word (flag / minimum, the others stop early) exactly as the
//     unplanned kernel works on the caller's output, and batch_publish_kernel - one lane per problem, launched behind the scan
//     only by plans that have such problems - copies the state word to the output and puts it back to idle.
// Problems without a scan (eff == 0: the empty needle, a bad position, a haystack shorter than the needle) are answered by
// their slice-0 workgroup from the descriptor.
// (-DSS_BATCH_WAVES_MAX=6 lets the register allocator aim at six waves per SIMD - tried with SLICESLICE_BATCH_OCC = 5 and 6 on
// every batch shape: no difference, so four it stays)
// ---- a plan run is ONE launch ------------------------------------------------------------------------------
// What the second launch (batch_publish_kernel) did for problems scanned by several workgroups - copy the state word to the caller's
// output, put it back to idle, tally - needs something that happens once per run and problem, in front of every finder.  Counting
// workgroups out was built in round 4 and measured at 5-8 % (a returning atomic at the end of every short-lived workgroup).  What
// works without any counting:
//   * every run has an IDENTITY that all its workgroups see and no neighbouring run shares: the AQL dispatch id of the launch (the
//     packet's index in its queue, 64 bits, from the hardware) and the id of the queue.  The plan's control word holds
//     (identity << 1 | parity) of the latest run.  A workgroup that reads its own identity there takes the parity as it is; one that
//     reads another identity - the previous run's - takes the OTHER parity; workgroup 0 stores (own identity, own parity).  Whichever
//     of the two a workgroup reads, it computes the same parity: consistent within a run, flipped between consecutive runs, no
//     atomics, nobody waits.  hipGraph replays get fresh dispatch ids like any launch.
//   * a problem has TWO state words, one per parity.  The waves of a run poll and raise word[parity]; the workgroup of slice 0 puts
//     word[parity ^ 1] back to idle - nobody looks at it in this run - for the run after.
//   * the caller's output: the slice-0 workgroup stores the idle value at its entry; a wave that finds the needle raises the state
//     word FIRST and writes the output behind it; the slice-0 workgroup re-reads the state word at its own end (behind a wait:
//     its store has been performed) and writes the output again if it is set.  Either a finder's state update is seen by that
//     re-read, or it came later - then so did its output store, behind the idle value.  FIND: the same with minima.
//   * the tally of found problems (plans with two layouts): the first finder of a problem adds one to tally[parity]; workgroup 0 of
//     the NEXT run stores the previous run's count to pinned memory and clears it.
struct __attribute__((aligned(64))) PlanCtl {
    unsigned long long run_word;                    // (identity of the latest run) << 1 | its parity
    uint32_t tally[2];                              // by parity: problems found (scanned by several workgroups) in that run
    uint32_t pad[12];
};
static_assert(sizeof(PlanCtl) == 64, "a line of its own");
struct __attribute__((aligned(128))) PlanState {
    // by parity, each word in a 64-byte half of its own: the word of THIS run is polled by every wave of the problem once per tile,
    // and the other one is written (re-armed) during the run - in the polled line that store cost round-robin plans 10 %
    // (profiles/r06/ab_plan_one_launch_parts.jsonl).  bool plans: 0 / 1; find plans: the leftmost offset so far (idle: all ones)
    struct __attribute__((aligned(64))) Half {
        uint64_t word;
        uint64_t pad[7];
    } half[2];
};
static_assert(sizeof(PlanState) == 128, "a problem's state: a cache line of its own, one half per run parity");
extern "C" __device__ unsigned long long ss_llvm_dispatch_id(void) __asm("llvm.amdgcn.dispatch.id");
__device__ __forceinline__ unsigned long long plan_run_identity()
{
    // 27 bits that tell the queue - the page number of its hsa_queue_t, an SGPR pair the hardware hands every wave: two live queues
    // never share it (the structures are pages apart; the bits compared cover 512 GiB of address space) - | 36 bits of the packet's
    // index in that queue.  NOTHING is read from the queue structure itself: it lives in host memory, and a first cut that loaded
    // hsa_queue_t::id from it paid a PCIe round trip at every workgroup's entry (1,024 x 1 MiB: 0.221 ms a run instead of 0.157;
    // profiles/r06/ab_plan_one_launch_queue_struct_read.jsonl).  (profiles/r06/dispatch_id_probe.json: one dispatch id per launch, a new
    // one per launch and per hipGraph replay; every stream its own queue structure.)
    const uint64_t queue = (uint64_t)(uintptr_t)__builtin_amdgcn_queue_ptr() >> 12;
    return ((queue & 0x7FFFFFFull) << 36) | (ss_llvm_dispatch_id() & ((1ull << 36) - 1ull));
}

#ifndef SS_BATCH_WAVES_MAX
#define SS_BATCH_WAVES_MAX 4
#endif
// MULTI (plans only): some problem of the plan is scanned by several workgroups - the run needs its parity, the state words, the
// slice-0 duties.  A plan whose every problem is scanned by ONE workgroup (16,384 x 64 KiB; the short cuts of config 5) launches the
// MULTI = false instantiation, which holds none of it: such problems publish from LDS, and the entry of their short-lived
// workgroups is what it was before (3-6 % on those shapes: profiles/r06/ab_plan_one_launch.jsonl).
template <int U, bool FIND = false, bool PLAN = false, bool MULTI = PLAN>
__global__ void __attribute__((amdgpu_waves_per_eu(4, SS_BATCH_WAVES_MAX))) __launch_bounds__(kBlock)
scan_batched_plan_kernel(const BatchArgs a, const BatchDesc *__restrict__ descs, uint32_t count, uint32_t nslices, BatchCold *colds,
                         PlanCtl *ctl, PlanState *states, unsigned long long *h_tally, uint32_t run)
{
    constexpr bool COUNTED = PLAN;                  // (the name the code below grew up with)
    __shared__ __attribute__((aligned(16))) uint8_t s_needle[kWavesPerBlock * kNeedleLds];
    // PLAN, eff == 1: a match word per WAVE (bool: the low int, 0 -> 1; FIND: minimum) - each wave sets its own word to idle before
    // it scans, so no barrier is needed in front of the scan; the one behind it settles all four for thread 0
    __shared__ unsigned long long s_wg[kWavesPerBlock];
    const uint32_t w = blockIdx.x;
    const bool slice_major = nslices <= kPlanSliceMajorMax;
    uint32_t prob, slice;
    if (slice_major) {
        slice = w / count;
        prob = w - slice * count;
    } else {
        prob = w / nslices;
        slice = w - prob * nslices;
    }
    // The problem's state word.  Unplanned calls: in its cold record (BatchCold: a line per pair of problems), not in an array of words.
    // A plan's run: word[parity] of the problem's PlanState - the parity comes from the plan's control word and this launch's identity
    // (see PlanCtl), requested here together with the descriptor: one scalar round trip.
    BatchCold *rec = colds + prob;
    const BatchDesc *dp = descs + prob;
    // slice-major, later slices: the problem's flag (one coherent load; a plan's run: both parities' words, chosen below) is requested
    // together with the descriptor (one scalar load, s_load_dwordx16) - one round trip decides whether and what to scan
    const bool peek = !FIND && slice_major && slice != 0;
    int seen0 = 0, seen1 = 0;
    if (MULTI) {
        if (peek) {
            seen0 = __hip_atomic_load(reinterpret_cast<int *>(&states[prob].half[0].word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seen1 = __hip_atomic_load(reinterpret_cast<int *>(&states[prob].half[1].word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (peek) {
        seen0 = __hip_atomic_load(reinterpret_cast<int *>(&rec->pad[1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    BatchDesc d = *dp;
    unsigned long long run_word = 0;
    // (issued behind the descriptor's load and waited for together with it: one scalar round trip)
    static_assert(PLAN || !MULTI, "only plans have runs");
    constexpr bool multi = MULTI;
    if (multi) __asm__ volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(run_word) : "s"(ctl) : "memory");
    if (COUNTED) {
        // The hot fields are pinned in scalar registers HERE, in front of the first store of the kernel (the LDS words'
        // initial values, the control word, the trivial problem's answer below): a load the compiler sinks behind a store cannot go through the scalar cache any more, so it became a
        // per-lane load and everything computed from it - tile bounds, loop control, addresses - per-lane arithmetic under exec
        // masks (101 VGPRs, and 311 us where the uncounted kernel takes 154 on 1,024 x 1 MiB).
        uint64_t base = reinterpret_cast<uint64_t>(d.base);
        __asm__ volatile("" : "+s"(base), "+s"(d.end), "+s"(d.nchunks_all), "+s"(d.per), "+s"(d.bytes), "+s"(d.shifts));
        d.base = reinterpret_cast<const uint8_t *>(base);
        if ((threadIdx.x & (kWave - 1)) == 0)
            __hip_atomic_store(&s_wg[threadIdx.x / kWave], FIND ? ~0ull : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    uint32_t parity = 0;
    if (multi) {
        const unsigned long long me = plan_run_identity();
        const bool mine = (run_word >> 1) == me;
        parity = mine ? (uint32_t)(run_word & 1ull) : (uint32_t)(run_word & 1ull) ^ 1u;
        if (blockIdx.x == 0 && threadIdx.x == 0 && !mine) {
            // the first workgroup of a run: the previous run's tally to the host (plans with two layouts), the run's identity and parity
            // into the control word.  Workgroups that still read the old word arrive at the same parity.
            if (h_tally) {
                const uint32_t total = __hip_atomic_exchange(&ctl->tally[parity ^ 1u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(h_tally, ((unsigned long long)run << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __hip_atomic_store(&ctl->run_word, (me << 1) | parity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // (MULTI = false: every problem publishes from LDS; the word the tiles poll is the idle one in the cold record, as in the calls)
    const int seen = MULTI ? (parity ? seen1 : seen0) : seen0;
    int *found = FIND ? nullptr : (MULTI ? reinterpret_cast<int *>(&states[prob].half[parity].word) : reinterpret_cast<int *>(&rec->pad[1]));
    void *sink = FIND ? (MULTI ? static_cast<void *>(&states[prob].half[parity].word) : static_cast<void *>(&rec->pad[0])) : static_cast<void *>(found);
    const uint32_t mis = d.shifts & 15;
    const uint64_t npieces = ((mis + d.end + 15) / 16 + 63) / 64;
    const uint64_t ntiles = (npieces + kWavesPerBlock * U - 1) / (kWavesPerBlock * U);
    const uint32_t eff = (uint32_t)(d.per >> 32), per = (uint32_t)d.per;
    if (slice >= eff) {                             // surplus slice, or a problem that needs no scan (eff == 0)
        if (COUNTED && eff == 0 && slice == 0 && threadIdx.x == 0) {
            if (FIND) a.best[prob] = d.n == 0 ? 0ull : ~0ull;
            else a.found[prob] = (int)d.shifts;     // the plan kernel's answer: 1 (empty needle), kBadPosition, or 0
        }
        return;
    }
    uint64_t t0, te, step;
    bool work = true;
    if (slice_major) {
        // (run = dispatch order on purpose.  Letting problem p's first-dispatched workgroup take run (p mod eff) - so that many
        // needles over ONE text do not all read the same place at the same time - was tried: the i386 loop 0.180 ms instead of
        // 0.134, its words are found in the first run; the other shapes the same.  tools/ab_batch_inproc.py --i386)
        t0 = (uint64_t)slice * per;
        te = t0 + per < ntiles ? t0 + per : ntiles;
        step = 1;
        work = __builtin_amdgcn_readfirstlane(seen) == 0;        // later slices of a needle that has been found: nothing to do
    } else {
        t0 = slice;
        te = ntiles;
        step = eff;
    }
    work = work && t0 < te;
    if (!work) return;                              // (a plan's single-workgroup problem always has work: t0 = 0 < te)
    const bool opener = MULTI && eff > 1 && slice == 0 && threadIdx.x == 0;      // (slice 0 always has work: its first tile is tile 0)
    if (opener) {
        // one lane per problem and run: the OTHER parity's state word back to idle for the run after this one (nobody looks at it in
        // this run), and the caller's output to its idle value - every finder writes the output BEHIND its update of the state word
        // that this lane re-reads at its own end (below)
        __hip_atomic_store(&states[prob].half[parity ^ 1u].word, FIND ? ~0ull : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (read-modify-write atomics: performed AT the device's coherence point, where the finders' atomics on the same words are)
        if (FIND) (void)__hip_atomic_exchange(a.best + prob, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else (void)__hip_atomic_exchange(a.found + prob, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    if (work) {
        Problem pr;                                 // hot fields only; the cold ones are re-read from the descriptor
        pr.base = d.base;
        pr.nchunks_all = d.nchunks_all;
        pr.npieces = npieces;
        pr.d = 0;
        pr.find_base = 0;
        pr.mis = mis;
        pr.r = (d.shifts >> 4) & 3;
        pr.n0x4 = 0x01010101u * (d.bytes & 0xFF);
        pr.nlx4 = 0x01010101u * ((d.bytes >> 8) & 0xFF);
        pr.n3x4 = 0x01010101u * ((d.bytes >> 16) & 0xFF);
        pr.r3 = (d.shifts >> 8) & 3;
        pr.q3 = (d.shifts >> 10) & 3;
        pr.epoch = 1;
        pr.flags = 0;
        pr.q = (d.shifts >> 6) & 3;
        // (a plan's problems come with their cold part ready-made; the unplanned kernel's waves build it when they need it)
        // (measured against plans whose waves build it like the unplanned kernel's, in one process - commit 21a0590 with
        // -DSS_NO_PLAN_COLD, profiles/r05/ab_plan_cold.jsonl: absent needles on random bytes the same to +-1 %, every second needle
        // present 16,384 x 64 KiB 0.154 ms instead of 0.183, 65,536 x 16 KiB 0.271 instead of 0.447)
        constexpr bool READY = PLAN;
        typename std::conditional<READY, ColdInPlanT<MULTI>, ColdInCall>::type cold;
        // (a plan's problem scanned by several workgroups: the finding wave writes the caller's output behind the state word)
        if constexpr (READY)
            cold = ColdInPlanT<MULTI>{dp, colds + prob, a.needles, MULTI && eff > 1 ? (FIND ? static_cast<void *>(a.best + prob) : static_cast<void *>(a.found + prob)) : nullptr,
                              MULTI && eff > 1 && h_tally ? &ctl->tally[parity] : nullptr};
        else cold = ColdInCall{dp, colds + prob, a.needles, FIND ? static_cast<void *>(a.best + prob) : static_cast<void *>(a.found + prob)};
        // single stream, non-temporal loads; the second byte's window is run-time data (kQDynamic)
        // (Measured and not adopted - commit 7511606 (-DSS_SIBLING_POLL), profiles/r05/ab_sibling_poll.jsonl: the waves of such a workgroup polling EACH
        // OTHER'S words between tiles, so that a match by one stops the other three.  It takes a barrier in front of the scan - a
        // wave must not read a sibling's word before its initial value is in place - and 10-15 more scalar registers at entry.)
        void *wg_sink = COUNTED && eff == 1 ? static_cast<void *>(&s_wg[threadIdx.x / kWave]) : nullptr;
        if ((d.bytes >> 24) & 1) scan_tiles<0, 0, true, U, 1, FIND, false, true>(pr, cold, s_needle, t0, step, te, sink, wg_sink);
        else scan_tiles<kQDynamic, 0, false, U, 1, FIND, false, true>(pr, cold, s_needle, t0, step, te, sink, wg_sink);
    }
    if (opener) {
        // ... the wave's own scan is over: the idle value has been PERFORMED by now (device-scope stores are acknowledged from the
        // device's coherence point, and the wait below has seen the acknowledgement), so a state word that is still idle here means
        // that whoever raises it later writes the output later too.  A WAIT, not a fence: an agent-scope fence writes back and
        // invalidates the XCD's whole L2 - with one such workgroup per problem and run that cost a slice-major plan a third of its
        // rate (profiles/r06/ab_plan_one_launch_with_fences.jsonl: 1,024 x 1 MiB 0.203 ms a run instead of 0.156)
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FIND) {
            const uint64_t v = __hip_atomic_load(&states[prob].half[parity].word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != ~0ull) __hip_atomic_fetch_min(a.best + prob, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const int v = __hip_atomic_load(reinterpret_cast<int *>(&states[prob].half[parity].word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != 0) (void)__hip_atomic_exchange(a.found + prob, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (COUNTED && eff == 1) {
        // the only workgroup of its problem: the answer is in the LDS word (a bare barrier settles it), one store publishes it
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long mine = FIND ? ~0ull : 0ull;
#pragma unroll
            for (int k = 0; k < kWavesPerBlock; ++k) {
                const unsigned long long v = __hip_atomic_load(&s_wg[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                mine = FIND ? (v < mine ? v : mine) : (mine | v);
            }
            if (FIND) a.best[prob] = mine;
            else a.found[prob] = mine != 0;
        }
    }
}

// ---- short-haystack pairs: one LANE per (needle, haystack) problem ---------------------------------
// The shape of the reference's short-haystack loop (bench/benches/i386.rs:118-129, tests/i386.rs:46-59:
// 10.5 M word-in-word searches of <= 24 bytes each): far too small for a workgroup per problem.  Each
// lane runs the same two-byte filter + compare sequentially over its few candidate offsets.
__global__ void __launch_bounds__(kBlock) scan_pairs_kernel(const BatchArgs a, uint64_t count)
{
    const uint64_t prob = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (prob >= count) return;
    const uint64_t h0 = a.hay_begin[prob], n0 = a.needle_begin[prob];
    const uint64_t len = a.hay_end[prob] - h0, n = a.needle_end[prob] - n0;
    int result = 0;
    uint64_t position = (a.position && n) ? a.position[prob] : n - 1;
    if (n == 0) {
        result = 1;
    } else if (n == 1 ? position != 0 : position >= n) {              // x86.rs:300, 473
        result = kBadPosition;
    } else if (len >= n) {
        const uint8_t *h = a.haystacks + h0, *nd = a.needles + n0;
        const uint8_t first = nd[0], last = nd[position];
        const uint64_t end = len - n + 1;
        for (uint64_t i = 0; i < end && !result; ++i) {
            if (h[i] != first || h[i + position] != last) continue;
            uint64_t k = 1;
            while (k < n && h[i + k] == nd[k]) ++k;
            result = k >= n;
        }
    }
    a.found[prob] = result;
}

}  // namespace ss
