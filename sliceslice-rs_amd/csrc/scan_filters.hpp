// scan_filters.hpp - the wave64 building blocks of the scan: the Problem descriptor, the byte-difference filters (first phase),
// the in-register second level, exact and in-memory verification, flag polling.  scan_kernels.hpp has the overview and puts
// them together (scan_tiles / scan_kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

namespace ss {

constexpr int kWave = 64;
constexpr int kBlock = 256;              // 4 waves: the batched / auxiliary kernels, and the scan's default
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxBlock = 512;           // scan_kernel takes its workgroup size from the launch (128 / 256 / 512)
constexpr int kMaxWavesPerBlock = kMaxBlock / kWave;
constexpr unsigned kPeekFromBlock = 1024;   // workgroups before this one start with the launch: nothing to see yet
#ifndef SS_BATCH_MIN_TILES
#define SS_BATCH_MIN_TILES 8
#endif
constexpr uint64_t kBatchMinTiles = SS_BATCH_MIN_TILES;   // batched kernel: tiles (16 KiB each) a slice should at least hold
constexpr int kFindOffsetBits = 40;      // completion-word find(): offsets below 2^40, the launch key above (see scan_kernel)
constexpr int kNeedleLds = 2048;         // needle bytes staged in LDS per wave; longer needles continue from global

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// One haystack/needle problem in "aligned coordinates": a = byte offset from `base` (16-B aligned).
//
// HOT fields first: what every wave needs before and while it streams.  The COLD fields behind them are needed only by a
// wave that has met a candidate (second-level schedule, verification, publishing) or by the last instructions of a
// workgroup (completion word).  scan_kernel receives the whole struct as its first kernel argument but reads the cold
// part straight from the kernarg segment, through a pointer the compiler cannot see through (ColdInKernarg), at the
// point of use: loaded at kernel entry like the hot fields they cost ~30 scalar registers that were spilled to vector
// lanes (v_writelane) in front of every short-lived workgroup's first load.
struct Problem {
    // ---- hot ----
    const uint8_t *base;      // hay + first - mis: the 16-byte-aligned start of the filter stream
    uint64_t nchunks_all;     // ceil((mis + len - first) / 16): chunks that contain a haystack byte
    uint64_t npieces;         // ceil(ceil((mis + end) / 16) / 64)
    uint64_t d;               // position / 16: chunk displacement of the second stream
    uint64_t find_base;       // FIND kernels: global offset of hay[0] (range shards), added to the match index
    uint32_t mis;             // 0..15
    uint32_t r;               // (position % 16) % 4: byte part of the shift
    uint32_t n0x4, nlx4;      // first and second filter byte, splatted over a dword
    // MODE 0 kernels test a THIRD needle byte in the first phase (position3 = 4*q3 + r3 < 16, relative to the first
    // filter byte like `position`; == position when the needle has no third byte to offer): text passes a two-byte
    // filter often enough that most tiles would enter the second phase, a three-byte filter hardly ever.
    uint32_t n3x4, q3, r3;
    int epoch;                // the value that means "found" in the flag (1 for caller-owned flags; pool slots
                              // use a fresh value per call, so a slot never has to be cleared)
    uint32_t flags;           // kProblemCounted: a completion word is in use (done_counter / host_done below)
    uint32_t q;               // (position % 16) / 4: the second byte's dword window - read by kernels instantiated with kQDynamic
                              // (batched, service); scan_kernel has it as a template argument
    // ---- cold ----
    const uint8_t *hay;       // the caller's pointer
    const uint8_t *needle;    // device copy of the needle
    uint64_t n;               // needle length (>= 1)
    uint64_t end;             // number of candidate offsets = len - n + 1   (>= 1)
    uint64_t order_idx[2];    // second-level filter: indices K of the extra needle bytes to test (relative to the first filter
    uint64_t order_val[2];    //   byte, rarest first, 1 byte each - entry t: word t/8, bits 8(t%8)..) and needle[K] in that order
    uint32_t norder;          //   how many (<= 15)
    // Exact in-register verification (the reference's const-length compare for SIZE = Some(1..=16), lib.rs:222-241):
    // when the needle ends at most 16 bytes behind the first filter byte, tail16 holds the L <= 16 needle bytes
    // needle[first - back .. n) (zero padded; back = as many of the bytes in front of the first filter byte as sixteen leave
    // room for - all of them for a needle of up to 16 bytes) and exact_len = L | back << 8; a candidate that survives the
    // second level is then compared against these four dwords in registers - no LDS staging, no re-read of the haystack
    // (exact_verify_piece).  exact_len == 0: the memory compare decides.
    uint32_t exact_len;
    uint32_t tail16[4];
    int *host_flag;           // optional pinned-host mirror of the found flag (saves the D2H copy); may be null
    // Completion word (small grids of ss_search_device / ss_find_device only; both null otherwise): every workgroup
    // counts itself out on *done_counter; the last one stores the answer to the pinned-host word *host_done - search:
    // (found-half of the counter) << 32 | epoch << 1 | found; find: the leftmost offset + 1, or all ones.  The host spins
    // on that word instead of waiting for the stream: one PCIe write instead of the completion-signal round trip.
    unsigned long long *done_counter;
    long long *host_done;
    // The counter is never reset: its low half counts workgroups out (the launch is complete when it reaches done_target),
    // its high half counts the workgroups that found the needle (found == the half has moved on from done_hi).  The host
    // keeps both halves per slot and starts over - behind a device synchronise - long before the low half could carry.
    uint32_t done_target, done_hi;
    // A caller-set filter pair too far apart for any kernel (ss_searcher_set_filter3: distance >= 16 * 63): the device filters
    // with the first byte and two partners close behind it, and the caller's far byte needle[far_off] is what a surviving
    // candidate at index i is tested for FIRST when it reaches memory (hay[i + far_off]).  0: none.
    uint64_t far_off;
};
constexpr uint32_t kProblemCounted = 1u;
constexpr int kQDynamic = -1;                     // scan_tiles<Q = kQDynamic, ...>: the window comes from Problem::q

// Where a wave finds the COLD fields of its Problem.
struct ColdInKernarg {        // scan_kernel: the Problem is the kernel's FIRST argument, i.e. offset 0 of the kernarg segment
    static constexpr bool kHasOrder = true;           // the second-level schedule and the needle's dwords come with the Problem
    static constexpr bool kMaybeOrder = false;
    typedef const Problem __attribute__((address_space(4))) *Ptr;
    __device__ __forceinline__ Ptr operator()() const
    {
        Ptr kp = (Ptr)__builtin_amdgcn_kernarg_segment_ptr();
        __asm__ volatile("" : "+s"(kp));     // opaque: the loads behind it stay where they are written
        return kp;
    }
};
struct ColdInRegisters {      // kernels that build their Problem themselves (batched)
    static constexpr bool kHasOrder = true;
    static constexpr bool kMaybeOrder = false;
    const Problem *p;
    __device__ __forceinline__ const Problem *operator()() const { return p; }
};

__device__ __forceinline__ uint32_t zero_byte_flags(uint32_t x) { return (x - 0x01010101u) & ~x; }

// lane l < 63 receives cur[l+1]; lane 63 keeps `last` (DPP wave_shl:1 without bound_ctrl leaves a lane
// that has no source lane untouched, i.e. equal to the `old` operand).
__device__ __forceinline__ uint32_t from_next_lane_or(uint32_t last, uint32_t cur)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)last, (int)cur, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

// lane l receives lane (l+1) mod 64: lane 63 gets lane 0 (DPP wave_rol:1).
__device__ __forceinline__ uint32_t rotate_from_next_lane(uint32_t v)
{
    // every lane has a source lane under wave_rol, so the `old` operand is never read: mov_dpp leaves it undefined and
    // saves the v_mov that update_dpp(0, ...) needs to materialise it (one VALU per moved dword)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x134 /* wave_rol:1 */, 0xf, 0xf, false);
}

// (Round 4 typed these loads as address_space(1) for a while, because the batched and service kernels - whose base pointer arrives
// through memory - get FLAT loads from the generic pointer.  It bought those kernels nothing measurable and cost the one-byte
// kernel 16 % (6.1 instead of 7.3 TB/s at 1 GiB: one of its eight loads lost its immediate offset and the schedule around it
// changed) and the 16-byte kernels 1-2 % at 1 GiB - profiles/r04/ab_global_cast.jsonl.  Generic pointers it is.)
template <bool NT>
__device__ __forceinline__ u32x4 load_chunk(const uint8_t *base, uint64_t chunk)
{
    const u32x4 *p = reinterpret_cast<const u32x4 *>(base) + chunk;
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// Full comparison of the needle with hay[i .. i+n), four bytes per step (unaligned global dword
// loads are legal on gfx950; the LDS/global needle side is dword-aligned by construction).
// Lane-private (divergent) on purpose: on random data almost every candidate dies in the first dword.
struct __attribute__((packed, aligned(1))) UnalignedU32 {
    uint32_t v;
};

__device__ __forceinline__ bool verify_candidate(const uint8_t *hay, const uint8_t *needle, uint64_t n, const uint8_t *s_needle,
                                                 uint64_t i)
{
    const uint8_t *h = hay + i;
    const uint64_t n_lds = n < (uint64_t)kNeedleLds ? n : (uint64_t)kNeedleLds;
    uint64_t k = 0;
    // sixteen bytes per step: the four haystack dwords are loaded together (one memory round trip per 16 bytes
    // instead of one per 4 - what a true match, whose every byte has to be looked at, is bound by)
    for (; k + 16 <= n_lds; k += 16) {
        const uint32_t a0 = reinterpret_cast<const UnalignedU32 *>(h + k)->v, a1 = reinterpret_cast<const UnalignedU32 *>(h + k + 4)->v;
        const uint32_t a2 = reinterpret_cast<const UnalignedU32 *>(h + k + 8)->v, a3 = reinterpret_cast<const UnalignedU32 *>(h + k + 12)->v;
        const u32x4 nd = *reinterpret_cast<const u32x4 *>(s_needle + k);
        if (((a0 ^ nd.x) | (a1 ^ nd.y) | (a2 ^ nd.z) | (a3 ^ nd.w)) != 0) return false;
    }
    for (; k + 4 <= n_lds; k += 4)
        if (reinterpret_cast<const UnalignedU32 *>(h + k)->v != *reinterpret_cast<const uint32_t *>(s_needle + k))
            return false;
    for (; k < n_lds; ++k)
        if (h[k] != s_needle[k]) return false;
    for (; k + 4 <= n; k += 4)   // needles longer than the LDS slice continue from the global copy
        if (reinterpret_cast<const UnalignedU32 *>(h + k)->v != reinterpret_cast<const UnalignedU32 *>(needle + k)->v)
            return false;
    for (; k < n; ++k)
        if (h[k] != needle[k]) return false;
    return true;
}

// h[0 .. count) == nd[0 .. count), both in global memory, for the few candidates the exact in-register compare hands over.
// Never a byte-by-byte loop - that is one dependent memory round trip per byte, half a microsecond each, which a text full of
// true matches paid in some wave of nearly every search: a dword per round trip, the last dword of a range OVERLAPPING the one
// before it so that no load reaches past either range (at most four round trips for up to sixteen bytes).
// (hb and nd are wave-uniform pointers, `off` the lane's 32-bit offset from hb: scalar base + vector offset addressing, one
// address register per lane instead of a 64-bit pointer per load - this sits inside kernels that live on 80 vector registers)
__device__ __forceinline__ bool same_bytes(const uint8_t *hb, uint32_t off, const uint8_t *nd, uint32_t count)
{
    auto u32 = [](const uint8_t *p, uint32_t o) { return reinterpret_cast<const UnalignedU32 *>(p + o)->v; };
    // 4 <= len <= 16 bytes from `at` on: the first and the last dword (all of a range of up to 8 bytes), then the two in
    // between; one load per side in flight - two pairs at once cost the kernels two vector registers they do not have
    auto group = [&](uint32_t at, uint32_t len) {
        const uint32_t o3 = at + len - 4;
        if (u32(hb, off + at) != u32(nd, at)) return false;
        if (u32(hb, off + o3) != u32(nd, o3)) return false;
        if (len <= 8) return true;
        const uint32_t o1 = at + 4, o2 = at + len - 8;
        if (u32(hb, off + o1) != u32(nd, o1)) return false;
        return u32(hb, off + o2) == u32(nd, o2);
    };
    if (count < 4) {                                    // 0 .. 3 bytes: first, middle, last
        if (count == 0) return true;
        const uint32_t mid = count >> 1, last = count - 1;
        return (uint32_t)((hb[off] ^ nd[0]) | (hb[off + mid] ^ nd[mid]) | (hb[off + last] ^ nd[last])) == 0;
    }
    for (uint32_t k = 0; k + 16 < count; k += 16)
        if (!group(k, 16)) return false;
    const uint32_t base = count > 16 ? count - 16 : 0;  // the last 4 .. 16 bytes (overlapping the group in front of them)
    return group(base, count - base);
}

// The filters work on raw byte DIFFERENCES: x ^ splat(b) has a zero byte exactly where the haystack byte
// equals b.  Differences of two needle bytes are combined with OR after one of them has been moved down
// the byte stream (cross-lane move + v_alignbyte), and a single zero-byte test then flags the offsets at
// which both bytes match - one test per dword instead of one per dword and needle byte plus an AND.

// Position-byte differences of one chunk (4 dwords).
__device__ __forceinline__ void position_diffs(const u32x4 &B, uint32_t nlx4, uint32_t w[4])
{
    w[0] = B.x ^ nlx4;
    w[1] = B.y ^ nlx4;
    w[2] = B.z ^ nlx4;
    w[3] = B.w ^ nlx4;
}

// Filter one piece.  A = this lane's chunk of the first-byte stream; w = position-byte differences of this
// lane's chunk of the position-byte stream; wl = what lane 63 must see as "the next lane's" differences
// (lane 0 of the next piece / the halo chunk; only lane 63's value is used).  Returns per-dword
// candidate flags (bit 7 of each candidate byte; the other bits are garbage).
template <int Q, bool ONE_BYTE>
__device__ __forceinline__ void filter_piece(const u32x4 &A, const uint32_t w[4], const uint32_t wl[4],
                                             const Problem &pr, uint32_t g[4])
{
    const uint32_t d0 = A.x ^ pr.n0x4, d1 = A.y ^ pr.n0x4, d2 = A.z ^ pr.n0x4, d3 = A.w ^ pr.n0x4;
    if (ONE_BYTE) {
        g[0] = zero_byte_flags(d0); g[1] = zero_byte_flags(d1); g[2] = zero_byte_flags(d2); g[3] = zero_byte_flags(d3);
        return;
    }
    // 8-dword window {this lane's chunk, next lane's chunk}; dwords Q .. Q+4 are needed.
    uint32_t x[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x[j] = w[j];
        x[4 + j] = (j <= Q) ? from_next_lane_or(wl[j], w[j]) : 0u;
    }
    g[0] = zero_byte_flags(d0 | __builtin_amdgcn_alignbyte(x[Q + 1], x[Q + 0], pr.r));
    g[1] = zero_byte_flags(d1 | __builtin_amdgcn_alignbyte(x[Q + 2], x[Q + 1], pr.r));
    g[2] = zero_byte_flags(d2 | __builtin_amdgcn_alignbyte(x[Q + 3], x[Q + 2], pr.r));
    g[3] = zero_byte_flags(d3 | __builtin_amdgcn_alignbyte(x[Q + 4], x[Q + 3], pr.r));
}

// Three-byte filter of one piece (MODE 0: both extra bytes within 15 bytes of the first).  The RAW dwords of the next
// lane's chunk are moved once (DPP commutes with the xor), then every filter byte costs five xors, four
// v_alignbyte and four ors, and ONE zero-byte test per dword decides all three bytes.  A = this lane's chunk;
// NX = what lane 63 must see as "the next lane's chunk" (lane 0 of the next piece, already rotated into lane 63, or
// the halo chunk); only dwords 0 .. max(Q, Q3) of it are used.
template <int Q, int Q3>
__device__ __forceinline__ void filter_piece3(const u32x4 &A, const uint32_t NX[4], const Problem &pr, uint32_t g[4])
{
    constexpr int QM = Q > Q3 ? Q : Q3;
    uint32_t x[8];
    x[0] = A.x; x[1] = A.y; x[2] = A.z; x[3] = A.w;
    x[4] = from_next_lane_or(NX[0], A.x);
    x[5] = QM >= 1 ? from_next_lane_or(NX[1], A.y) : 0u;
    x[6] = QM >= 2 ? from_next_lane_or(NX[2], A.z) : 0u;
    x[7] = QM >= 3 ? from_next_lane_or(NX[3], A.w) : 0u;
    uint32_t y[5], z[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        y[k] = x[Q + k] ^ pr.nlx4;
        z[k] = x[Q3 + k] ^ pr.n3x4;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        g[j] = zero_byte_flags((x[j] ^ pr.n0x4) | __builtin_amdgcn_alignbyte(y[j + 1], y[j], pr.r) |
                               __builtin_amdgcn_alignbyte(z[j + 1], z[j], pr.r3));
}

// ---- second-level filter ------------------------------------------------------------------------------
// Run only by waves that have candidates: AND the candidate flags with the flags of needle[K] at byte
// offset K, for up to 15 further needle bytes, still entirely in registers.  Text-like haystacks pass
// the two-byte filter at percent rates; every extra byte cuts that by the byte's frequency before any
// candidate touches memory.  The bytes are tried rarest-first (a static, corpus-free rarity guess:
// build_refine_order) and the wave stops as soon as no lane has a candidate left.

// Smaller = expected to be rarer in typical haystacks (text, logs, source, binaries).  Only the ORDER
// of the checks depends on this; the result of a search never does.
__host__ __device__ constexpr inline int byte_rarity_rank(uint8_t b)
{
    if (b == ' ') return 255;
    if (b >= 'a' && b <= 'z') {
        // 250 - 4 * (place in "etaoinshrdlcumwfgypbvkjxqz", most to least frequent English letters)
        const uint8_t kLetter[26] = {/*a*/ 242, /*b*/ 174, /*c*/ 206, /*d*/ 214, /*e*/ 250, /*f*/ 190, /*g*/ 186,
                                         /*h*/ 222, /*i*/ 234, /*j*/ 162, /*k*/ 166, /*l*/ 210, /*m*/ 198, /*n*/ 230,
                                         /*o*/ 238, /*p*/ 178, /*q*/ 154, /*r*/ 218, /*s*/ 226, /*t*/ 246, /*u*/ 202,
                                         /*v*/ 170, /*w*/ 194, /*x*/ 158, /*y*/ 182, /*z*/ 150};
        return kLetter[b - 'a'];
    }
    if (b == 0) return 200;                                    // zero padding is common in binaries
    if (b == '\n' || b == '\r' || b == '\t') return 140;
    if (b >= '0' && b <= '9') return 120;
    if (b == '.' || b == ',' || b == '-' || b == '_' || b == '/' || b == ':' || b == '"' || b == '=') return 110;
    if (b >= 'A' && b <= 'Z') return 100;
    if (b >= 0x21 && b <= 0x7E) return 60;                     // other printable punctuation
    if (b == 0xFF) return 50;
    return 20;                                                 // control bytes, 0x80..0xFE
}

// The second level's schedule: up to 15 of the indices 1 .. min(n,32)-1 (relative to the first filter byte) other than the
// first-phase bytes, packed one byte each.  Bytes 16..31 - the next lane's chunk, one more cross-lane hop - come FIRST, rarest
// first (at most kFarFirst of them), then bytes 1..15 rarest first: a candidate that has passed three rare bytes on text is
// usually an occurrence of a stock phrase around those bytes, and what tells the needle from the phrase is more likely to
// sit in the NEXT words than between the filter bytes.  Only the order (and which 15 of up to 29 bytes are tried before the
// compare) depends on this; the result of a search never does.
constexpr int kRefineWindow = 32;
constexpr uint32_t kFarFirst = 10;
#ifndef SS_REFINE_BYTES_PER_BALLOT
#define SS_REFINE_BYTES_PER_BALLOT 1
#endif
#ifndef SS_EXACT_REFINE_STEPS
#define SS_EXACT_REFINE_STEPS 2
#endif
constexpr uint32_t kExactRefineSteps = SS_EXACT_REFINE_STEPS;     // schedule bytes in front of the exact in-register compare
constexpr uint32_t kExactSparseLanes = 24;                        // ... none at all with this few candidate lanes in a tile
constexpr uint32_t kRefineBytesPerBallot = SS_REFINE_BYTES_PER_BALLOT;   // schedule bytes applied between two wave ballots

__host__ __device__ inline uint32_t build_refine_order(const uint8_t *needle, uint64_t n, uint64_t position,
                                                       uint64_t idx[2], uint64_t val[2], uint64_t position3 = ~0ull)
{
    uint8_t ks[2][kRefineWindow];            // [0] = far (K >= 16), [1] = near; each sorted by rarity rank
    int rk[2][kRefineWindow];
    uint32_t cnt[2] = {0, 0};
    const int lim = n < (uint64_t)kRefineWindow ? (int)n : kRefineWindow;
    for (int K = 1; K < lim; ++K) {
        if ((uint64_t)K == position || (uint64_t)K == position3) continue;   // already tested by the first-level filter
        const int g = K >= 16 ? 0 : 1;
        const int r = byte_rarity_rank(needle[K]);
        int at = (int)cnt[g];
        while (at > 0 && rk[g][at - 1] > r) {                  // insertion sort, stable
            rk[g][at] = rk[g][at - 1];
            ks[g][at] = ks[g][at - 1];
            --at;
        }
        rk[g][at] = r;
        ks[g][at] = (uint8_t)K;
        ++cnt[g];
    }
    idx[0] = idx[1] = val[0] = val[1] = 0;
    uint32_t m = 0;
    auto emit = [&](uint8_t K) {
        idx[m >> 3] |= (uint64_t)K << (8 * (m & 7));
        val[m >> 3] |= (uint64_t)needle[K] << (8 * (m & 7));
        ++m;
    };
    uint32_t far_used = 0;
    for (; far_used < cnt[0] && far_used < kFarFirst; ++far_used) emit(ks[0][far_used]);
    for (uint32_t t = 0; t < cnt[1] && m < 15; ++t) emit(ks[1][t]);
    for (; far_used < cnt[0] && m < 15; ++far_used) emit(ks[0][far_used]);
    return m;
}

// Device form for kernels that build the problem descriptor themselves (batched): lane K ranks
// needle[K] (K < 32); far bytes first, then near ones, four rarity classes each, emitted from wave ballots.  Coarser than
// the host sort, which only changes the order of the checks.
__device__ __forceinline__ uint32_t build_refine_order_wave(const uint8_t *needle, uint64_t n, uint64_t position,
                                                            int lane, uint64_t idx[2], uint64_t val[2], uint64_t position3 = ~0ull)
{
    const int lim = n < (uint64_t)kRefineWindow ? (int)n : kRefineWindow;
    const bool valid = lane >= 1 && lane < lim && (uint64_t)lane != position && (uint64_t)lane != position3;
    const uint32_t b = valid ? needle[lane] : 0u;
    const int r = byte_rarity_rank((uint8_t)b);
    const int cls = !valid ? -1 : (r < 64 ? 0 : (r < 128 ? 1 : (r < 192 ? 2 : 3)));
    uint64_t i0 = 0, i1 = 0, v0 = 0, v1 = 0;
    uint32_t m = 0;
    auto take = [&](uint32_t mask, uint32_t cap) {
        while (mask && m < cap) {
            const int K = __ffs((int)mask) - 1;
            mask &= mask - 1;
            const uint64_t v = (uint32_t)__builtin_amdgcn_readlane((int)b, K) & 0xFF;
            const uint32_t sh = 8 * (m & 7);
            if (m < 8) { i0 |= (uint64_t)K << sh; v0 |= v << sh; }
            else { i1 |= (uint64_t)K << sh; v1 |= v << sh; }
            ++m;
        }
    };
#pragma unroll 1
    for (int c = 0; c < 4; ++c) take((uint32_t)__ballot(cls == c) & 0xFFFF0000u, kFarFirst);     // bytes 16..31
#pragma unroll 1
    for (int c = 0; c < 4; ++c) take((uint32_t)__ballot(cls == c) & 0x0000FFFFu, 15u);            // bytes 1..15
    idx[0] = i0; idx[1] = i1; val[0] = v0; val[1] = v1;
    return m;
}

// Lane 63's next lane is lane 0 of the following piece: `N` is that piece's register (kind 1: lane 0
// holds the chunk -> wave_rol), or the halo chunk already sitting in lane 63 (kind 0).
struct NextPiece {
    u32x4 N;
    int kind;     // wave-uniform
};

__device__ __forceinline__ uint32_t next_lane_diffs(uint32_t own, uint32_t nword, uint32_t nkx4, int kind)
{
    const uint32_t f = nword ^ nkx4;
    return from_next_lane_or(kind == 1 ? rotate_from_next_lane(f) : f, own);       // `old` operand: what lane 63 will see
}

// The value two lanes ahead in the concatenation {this piece, next piece}: a second wave_shl:1 on top of next_lane_diffs.
// Lane 62 receives what lane 63 got in the first hop; lane 63 needs lane 1 of the next piece, which exists only when that
// piece is a register of this wave (kind 1) - otherwise it is unknown and passes ("matches"; the compare settles it).
__device__ __forceinline__ uint32_t next2_lane_diffs(uint32_t hop1, uint32_t nword, uint32_t nkx4, int kind)
{
    uint32_t last = 0u;
    if (kind == 1) last = rotate_from_next_lane(rotate_from_next_lane(nword ^ nkx4));
    return from_next_lane_or(last, hop1);
}

// One needle byte at offset K = 4*QK + rk (1..31): the differences to needle[K], moved down by K bytes, clear
// the candidate flags where they are not zero.  QK is a template parameter so that only the window dwords
// QK .. QK+4 of {own chunk, next lane's, the lane after's} are built (no run-time selects); rk is a run-time byte shift.
template <int QK>
__device__ __forceinline__ void refine_flags_q(const u32x4 &A, const NextPiece &np, uint32_t nkx4, uint32_t rk, uint32_t g[4])
{
    static_assert(QK >= 0 && QK <= 7, "second-level bytes lie within 32 bytes of the first filter byte");
    constexpr auto need = [](int i) { return i >= QK && i <= QK + 4; };
    const uint32_t own[4] = {A.x ^ nkx4, A.y ^ nkx4, A.z ^ nkx4, A.w ^ nkx4};
    const uint32_t nw[4] = {np.N.x, np.N.y, np.N.z, np.N.w};
    uint32_t e[12];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        e[j] = own[j];
        e[4 + j] = e[8 + j] = 0u;
        if (need(4 + j) || need(8 + j)) e[4 + j] = next_lane_diffs(own[j], nw[j], nkx4, np.kind);
        if (need(8 + j)) e[8 + j] = next2_lane_diffs(e[4 + j], nw[j], nkx4, np.kind);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] &= zero_byte_flags(__builtin_amdgcn_alignbyte(e[j + QK + 1], e[j + QK], rk));
}

// The second-level filter's schedule (Problem::norder / order_idx / order_val), as the wave holds it.
struct RefineOrder {
    uint32_t n;
    uint64_t idx[2], val[2];
};

// Second-level filter for a whole tile (U pieces of one wave): one needle byte at a time, rarest first, applied
// to all U pieces before the next wave ballot - the scalar bookkeeping (schedule entry, window switch, ballot)
// is paid once per tile and byte instead of once per piece and byte, and the U independent pieces hide each
// other's DPP / VALU latencies.  On text nearly every piece of a tile holds candidates, so nothing is wasted;
// on random bytes the extra pieces cost ~2 VALU per KiB on average.  Returns false when no lane of the wave
// has a candidate left in any piece.
template <int U, int MODE>
__device__ __forceinline__ bool refine_tile(const u32x4 (&A)[U], const u32x4 &H, const RefineOrder &ro, uint32_t (&G)[U][4],
                                            uint32_t max_steps)
{
    auto any_left = [&]() {
        uint32_t o = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) o |= G[u][0] | G[u][1] | G[u][2] | G[u][3];
        return __ballot((o & 0x80808080u) != 0) != 0;
    };
    auto apply = [&](auto qk_c, uint32_t nkx4, uint32_t rk) {
        constexpr int QK = decltype(qk_c)::value;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            NextPiece np;
            np.N = u + 1 < U ? A[u + 1] : H;
            np.kind = u + 1 < U ? 1 : (MODE == 0 ? 0 : 1);
            refine_flags_q<QK>(A[u], np, nkx4, rk, G[u]);
        }
    };
    bool any = any_left();
    // kRefineBytesPerBallot schedule bytes between two wave ballots.  ONE is the measured optimum (in one process on one
    // buffer, profiles/r03/ab_refine_bytes_per_ballot.jsonl): with two, the reference's pair (0, n-1) on text - every tile
    // dense with chance hits - ran at 5.4 TB/s instead of 6.1-6.8, with three at 4.6: the first byte clears most tiles, and
    // the dozen VALU operations per piece of a second one cost more than the ballot -> compare -> branch chain they save.
    const uint32_t steps = ro.n < max_steps ? ro.n : max_steps;
    uint32_t t = 0;
#pragma unroll 1
    while (t < steps && any) {
#pragma unroll 1
        for (uint32_t k = 0; k < kRefineBytesPerBallot && t < steps; ++k, ++t) {
            const uint32_t sh = 8 * (t & 7);
            const int K = (int)(((t < 8 ? ro.idx[0] : ro.idx[1]) >> sh) & 0xFF);
            const uint32_t v = (uint32_t)(((t < 8 ? ro.val[0] : ro.val[1]) >> sh) & 0xFF);
            const uint32_t nkx4 = 0x01010101u * v, rk = (uint32_t)(K & 3);
            switch (K >> 2) {                            // wave-uniform
            case 0: apply(std::integral_constant<int, 0>{}, nkx4, rk); break;
            case 1: apply(std::integral_constant<int, 1>{}, nkx4, rk); break;
            case 2: apply(std::integral_constant<int, 2>{}, nkx4, rk); break;
            case 3: apply(std::integral_constant<int, 3>{}, nkx4, rk); break;
            case 4: apply(std::integral_constant<int, 4>{}, nkx4, rk); break;
            case 5: apply(std::integral_constant<int, 5>{}, nkx4, rk); break;
            case 6: apply(std::integral_constant<int, 6>{}, nkx4, rk); break;
            default: apply(std::integral_constant<int, 7>{}, nkx4, rk); break;
            }
        }
        any = any_left();
    }
    return any;
}

// Per-piece form of the same filter: tiles in which at most two pieces hold candidates (the usual case with three filter
// bytes).  (Round 1 kept the MODE 2 kernels on this form for every tile - the tile-wide one cost them a wave of occupancy;
// since the cold fields left the registers both fit, and tile-wide is worth 4.5-4.9 -> 6.1-6.2 TB/s for the reference's
// pair on text: profiles/r03/ab_refine_bytes_per_ballot.jsonl, `m2pp` = per piece.)
// Returns false when no lane of the wave has a candidate left in this piece.
__device__ __forceinline__ bool refine_piece(const u32x4 &A, const NextPiece &np, const RefineOrder &ro, uint32_t g[4])
{
    bool any = __ballot(((g[0] | g[1] | g[2] | g[3]) & 0x80808080u) != 0) != 0;
    uint32_t t = 0;
#pragma unroll 1
    while (t < ro.n && any) {
#pragma unroll 1
        for (uint32_t k = 0; k < kRefineBytesPerBallot && t < ro.n; ++k, ++t) {     // two bytes per ballot: see refine_tile
            const uint32_t sh = 8 * (t & 7);
            const int K = (int)(((t < 8 ? ro.idx[0] : ro.idx[1]) >> sh) & 0xFF);
            const uint32_t v = (uint32_t)(((t < 8 ? ro.val[0] : ro.val[1]) >> sh) & 0xFF);
            const uint32_t nkx4 = 0x01010101u * v, rk = (uint32_t)(K & 3);
            switch (K >> 2) {                            // wave-uniform
            case 0: refine_flags_q<0>(A, np, nkx4, rk, g); break;
            case 1: refine_flags_q<1>(A, np, nkx4, rk, g); break;
            case 2: refine_flags_q<2>(A, np, nkx4, rk, g); break;
            case 3: refine_flags_q<3>(A, np, nkx4, rk, g); break;
            case 4: refine_flags_q<4>(A, np, nkx4, rk, g); break;
            case 5: refine_flags_q<5>(A, np, nkx4, rk, g); break;
            case 6: refine_flags_q<6>(A, np, nkx4, rk, g); break;
            default: refine_flags_q<7>(A, np, nkx4, rk, g); break;
            }
        }
        any = __ballot(((g[0] | g[1] | g[2] | g[3]) & 0x80808080u) != 0) != 0;
    }
    return any;
}

// Candidate verification for one lane's flags; returns true when the needle was found.  The four flag
// dwords are walked by a run-time loop so that the compare code exists once per call site.
// What the verification needs of a Problem's cold part, as the wave holds it once it has met a candidate.
struct VerifyArgs {
    const uint8_t *hay, *needle;
    uint64_t n, end;
};

template <bool ONE_BYTE>
__device__ __forceinline__ bool verify_flags(const uint32_t g[4], uint64_t chunk, const Problem &pr, const VerifyArgs &va,
                                             const uint8_t *s_needle, uint64_t &where, uint64_t far_off = 0)
{
    bool hit = false;
    // all 16 flags of the lane in one word: flag of byte 4j+t at bit 8t+j (bit 7 of byte t of g[j] >> (7-j))
    uint32_t m = ((g[0] & 0x80808080u) >> 7) | ((g[1] & 0x80808080u) >> 6) | ((g[2] & 0x80808080u) >> 5) |
                 ((g[3] & 0x80808080u) >> 4);
    // address order = j major, t minor: take dword 0's flags first (bits 0, 8, 16, 24), then dword 1's ...
#pragma unroll 1
    for (int j = 0; j < 4 && !hit; ++j) {
        uint32_t mj = (m >> j) & 0x01010101u;
        while (mj != 0 && !hit) {
            const int bit = __ffs((int)mj) - 1;         // lowest flagged byte first (tzcnt, lib.rs:221)
            mj &= mj - 1;                               // clear lowest set bit        (lib.rs:247)
            const uint64_t a = chunk * 16 + (uint64_t)(j * 4 + (bit >> 3));
            const uint64_t i = a - pr.mis;              // wraps for bytes in front of the haystack
            if (i < va.end) {
                if (ONE_BYTE) hit = va.hay[i] == (uint8_t)pr.n0x4;
                else if (far_off != 0 && va.hay[i + far_off] != va.needle[far_off]) hit = false;   // the caller's far filter byte
                else hit = verify_candidate(va.hay, va.needle, va.n, s_needle, i);
                where = i;                              // lowest match of this lane when hit
            }
        }
    }
    return hit;
}

// movemask of one flag dword: bit 7 of byte t -> bit t
__device__ __forceinline__ uint32_t flag_nibble(uint32_t g)
{
    return ((((g >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
}

// Exact verification of one lane's surviving flags WITHOUT touching memory (MODE 0 kernels, needles that end at most 16
// bytes behind the first filter byte).  `exact` = L | back << 8: the compare covers the L <= 16 needle bytes needle[first - back
// .. first - back + L) held in cmp16 - the bytes from the first filter byte on plus as many of the `back` bytes IN FRONT of it
// (filters chosen by rarity may start inside the needle) as sixteen allow; a needle of up to 16 bytes is covered whole.
// A lane holds the 32 stream bytes of its own chunk and the next lane's (raw dwords, one DPP hop - lane 63 takes lane 0 of the
// wave's next piece or the halo chunk).  A candidate at byte b of a chunk needs the bytes from b - back on: with b >= back
// they lie in that window; a candidate with b < back starts in the PREVIOUS lane's chunk, so its FLAG moves to that lane (one
// more DPP hop, of a 16-bit flag word), whose window holds all of it.  Only lane 0 has nobody in front of it: its first `back`
// flags are settled in memory (same_bytes; one candidate in ~170 on average.  Comparing them against lane 63's chunk of the
// wave's previous piece, held in scalar registers, was tried: five more vector registers at the kernels' peak, i.e. a wave of
// occupancy).  The window is brought to the candidate's byte offset with v_alignbyte and compared with cmp16 under a length
// mask.  Flags are walked lowest first (lib.rs:220-247) - a lane's own before those handed to it, which lie further right -
// so `where_off` is the lane's leftmost match and lanes stay in address order.  Needle bytes further in front than `back`
// (needles of more than 16 bytes) are compared in memory, for exact survivors only.
__device__ __forceinline__ bool exact_verify_piece(const u32x4 &A, const NextPiece &np, const uint32_t g[4], uint64_t chunk_wave,
                                                   int lane, const Problem &pr, const VerifyArgs &va, const uint32_t cmp16[4],
                                                   uint32_t exact, uint32_t &where_off)
{
    // index of the needle's first byte for a candidate at stream byte t of this lane's window: ubase (wave-uniform; wraps for
    // chunks in front of the haystack) + 16 * lane + t
    const uint64_t ubase = chunk_wave * 16 - pr.mis;
    const uint8_t *hb = va.hay + ubase;
    const uint32_t exact_len = exact & 0xFFu, back = (exact >> 8) & 0xFFu;          // wave-uniform
    // (named scalars, not an array: a select between array ELEMENTS becomes a select between addresses, and the window
    // ends up in scratch memory behind a dynamic index)
    auto hop = [&](uint32_t nword, uint32_t own) {
        return from_next_lane_or(np.kind == 1 ? rotate_from_next_lane(nword) : nword, own);
    };
    const uint32_t w0 = A.x, w1 = A.y, w2 = A.z, w3 = A.w;
    const uint32_t w4 = hop(np.N.x, w0), w5 = hop(np.N.y, w1), w6 = hop(np.N.z, w2), w7 = hop(np.N.w, w3);
    uint32_t M[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rem = (int)exact_len - 4 * j;
        M[j] = rem >= 4 ? ~0u : (rem <= 0 ? 0u : (1u << (8 * rem)) - 1u);
    }
    // bit t: a candidate whose first filter byte is stream byte t of {own chunk, next lane's chunk}
    uint32_t flags = flag_nibble(g[0]) | (flag_nibble(g[1]) << 4) | (flag_nibble(g[2]) << 8) | (flag_nibble(g[3]) << 12);
    if (back != 0) {
        const uint32_t low = flags & ((1u << back) - 1u);
        flags = (lane == 0 ? flags : flags & ~low) | (from_next_lane_or(0u, low) << 16);
    }
    const uint64_t anchor = (uint64_t)((pr.base + pr.mis) - va.hay);     // index of the first filter byte in the needle
    const uint32_t front = (uint32_t)(anchor - back);                    // needle bytes in front of the register window
    bool hit = false;
    while (flags != 0 && !hit) {
        const int t = __ffs((int)flags) - 1;            // lowest flagged byte first (tzcnt, lib.rs:221)
        flags &= flags - 1;                             // clear lowest set bit        (lib.rs:247)
        const uint32_t off = 16u * (uint32_t)lane + (uint32_t)t;
        const uint64_t i = ubase + off;                 // wraps for bytes in front of the haystack
        if (i >= va.end) continue;
        uint32_t in_memory = front;                     // needle bytes this candidate still has to match in memory
        if ((uint32_t)t < back) {
            // lane 0: the bytes in front of this candidate lie in a chunk the wave may not hold - the whole needle, in memory
            in_memory = (uint32_t)va.n;
        } else {
            const int start = t - (int)back;            // byte offset of needle[first - back] in the window: 0 .. 15
            const int q = start >> 2;
            const uint32_t r = (uint32_t)(start & 3);
            auto pick = [&](uint32_t a, uint32_t b1, uint32_t c, uint32_t d) {
                const uint32_t lo = q & 1 ? b1 : a, hi = q & 1 ? d : c;
                return q & 2 ? hi : lo;
            };
            const uint32_t sw[5] = {pick(w0, w1, w2, w3), pick(w1, w2, w3, w4), pick(w2, w3, w4, w5), pick(w3, w4, w5, w6),
                                    pick(w4, w5, w6, w7)};
            uint32_t diff = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) diff |= (__builtin_amdgcn_alignbyte(sw[j + 1], sw[j], r) ^ cmp16[j]) & M[j];
            if (diff != 0) continue;
        }
        hit = in_memory == 0 || same_bytes(hb, off, va.needle, in_memory);
        where_off = off;                                // lowest match of this lane when hit: index ubase + off
    }
    return hit;
}

// tells the compiler that a 64-bit value is wave-uniform (SGPR pair)
__device__ __forceinline__ uint64_t uniform64(uint64_t x)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
}

__device__ __forceinline__ int poll_found(const int *found, int epoch)
{
    return __builtin_amdgcn_readfirstlane(
               __hip_atomic_load(found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == epoch;
}

// Entry peek of a workgroup at the flag THROUGH THE SCALAR CACHE: a hit costs tens of cycles instead of an
// L2 round trip, so even one-tile workgroups can afford it before they issue their loads.  The scalar cache
// is not coherent - a stale "not found" only means the workgroup does its tile as usual.  Staleness is
// bounded: every tile also polls coherently (free, behind its data loads), and a wave that sees the flag set
// there invalidates its CU's scalar cache on the way out (forget_scalar_cache), so the workgroups that
// follow on that CU leave at the peek.  A peek HIT is always confirmed with a coherent load before the workgroup
// leaves (scan_kernel), so correctness never rests on the dispatch-time invalidation of the scalar cache
// (which tests/test_gpu_parity.py::test_caller_owned_flags_are_not_seen_stale observes on the current ROCm).
__device__ __forceinline__ int scalar_peek(const int *p)
{
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

__device__ __forceinline__ uint64_t scalar_peek64(const uint64_t *p)
{
    uint64_t v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

__device__ __forceinline__ void forget_scalar_cache() { __builtin_amdgcn_s_dcache_inv(); }
// ... which only matters to grids large enough to peek (kPeekFromBlock).  Completion-word launches are small grids, and their
// last instructions read the cold half of the Problem back through that very cache: invalidating it there puts a memory
// round trip on the path a match's latency is made of.
__device__ __forceinline__ void forget_scalar_cache_unless(bool small_grid)
{
    if (!small_grid) __builtin_amdgcn_s_dcache_inv();
}

__device__ __forceinline__ void publish_found(int *found, int epoch = 1)
{
    __hip_atomic_store(found, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// FIND kernels keep the leftmost match offset in one uint64 (all-ones = none yet), lowered by atomicMin.

// Per-wave lazy staging of the needle into the wave's private LDS slice (no workgroup barrier: the DS
// operations of one wave execute in order).
__device__ __forceinline__ void stage_needle_wave(uint8_t *s_needle, const uint8_t *needle, uint64_t n, int lane)
{
    const uint32_t m = n < (uint64_t)kNeedleLds ? (uint32_t)n : (uint32_t)kNeedleLds;
    for (uint32_t k = lane; k < m; k += kWave) s_needle[k] = needle[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// scan_tiles (below): scans tiles tile0, tile0+tile_step, ... (< tile_end) of one problem with the calling
// workgroup.  A tile is kWavesPerBlock*U consecutive pieces; wave w owns pieces tile*4U + w*U + u, u < U.
// NTMODE: 0 = plain loads; 1 = non-temporal loads.
// ---- 8-bytes-per-lane first phase (L8) -------------------------------------------------------------------
// Plain streaming reads run ~2 % faster when a wave instruction covers 512 contiguous bytes (dwordx2 per
// lane) than 1 KiB (dwordx4) - profiles/r01/readbench_8gib.txt.  The L8 kernels therefore run the two-byte
// filter on *half-pieces*: 64 lanes x 8 bytes, two dwords per lane; a candidate's position byte lies up to
// two lanes ahead.  Only tiles in which some candidate survives are transposed (ds_bpermute) into the
// 16-bytes-per-lane layout and handed to the second phase unchanged; a wave that keeps meeting candidates
// (text) stays in the 16-byte layout for its following tiles.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ u32x2 load_half(const uint8_t *base, uint64_t half_chunk)
{
    const u32x2 *p = reinterpret_cast<const u32x2 *>(base) + half_chunk;
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// Two-byte filter of one half-piece on raw byte differences.  a = this lane's 8 bytes; t = a ^ needle[position]
// (zero bytes where the position byte matches); tn = the same of the NEXT half-piece, whose lanes 0 and 1 are
// what lanes 62/63 see one and two lanes ahead (position = 4*Q + r < 16 reaches at most 15 + 7 bytes on).
// The position-byte differences are brought `position` bytes down the stream (ds_bpermute for the lanes
// ahead - the LDS crossbar, not the VALU - and v_alignbyte for the byte part) and OR-ed onto the
// first-byte differences: a byte of the result is zero exactly where both filter bytes match, so ONE
// zero-byte test per dword replaces two tests and an AND.  Returns acc | flags (bit 7 of candidate bytes).
template <int Q, bool ONE_BYTE>
__device__ __forceinline__ uint32_t filter_half(const u32x2 &a, const u32x2 &t, const u32x2 &tn, const Problem &pr,
                                                int lane, uint32_t acc)
{
    const uint32_t d0 = a.x ^ pr.n0x4, d1 = a.y ^ pr.n0x4;
    if (ONE_BYTE) return acc | zero_byte_flags(d0) | zero_byte_flags(d1);
    // dword stream relative to this lane: x[0..1] this lane, x[2..3] next lane, x[4..5] the lane after;
    // stream dwords Q .. Q+2 are used
    uint32_t x[6] = {t.x, t.y, 0, 0, 0, 0};
    const int i1 = ((lane + 1) & (kWave - 1)) << 2, i2 = ((lane + 2) & (kWave - 1)) << 2;
    if (Q <= 2) x[2] = (uint32_t)__builtin_amdgcn_ds_bpermute(i1, (int)(lane < 1 ? tn.x : t.x));
    if (Q >= 1) x[3] = (uint32_t)__builtin_amdgcn_ds_bpermute(i1, (int)(lane < 1 ? tn.y : t.y));
    if (Q >= 2) x[4] = (uint32_t)__builtin_amdgcn_ds_bpermute(i2, (int)(lane < 2 ? tn.x : t.x));
    if (Q >= 3) x[5] = (uint32_t)__builtin_amdgcn_ds_bpermute(i2, (int)(lane < 2 ? tn.y : t.y));
    const uint32_t c0 = d0 | __builtin_amdgcn_alignbyte(x[Q + 1], x[Q], pr.r);
    const uint32_t c1 = d1 | __builtin_amdgcn_alignbyte(x[Q + 2], x[Q + 1], pr.r);
    return acc | zero_byte_flags(c0) | zero_byte_flags(c1);
}

// THREE-byte filter of one half-piece when both further bytes lie within three bytes of the first (Q == 0 and q3 == 0: the window
// is this lane's two dwords and the FIRST dword of the next lane - one DPP hop, where a farther byte would need the lane after
// next as well: two hops per value, which costs more than the 8-byte loads bring).  a = this lane's 8 bytes, nx = dword 0 of the
// next lane (lane 63: of lane 0 of the next half-piece / the halo).  Returns acc | flags.
__device__ __forceinline__ uint32_t filter_half3_near(const u32x2 &a, uint32_t nx, const Problem &pr, uint32_t acc)
{
    const uint32_t y0 = a.x ^ pr.nlx4, y1 = a.y ^ pr.nlx4, y2 = nx ^ pr.nlx4;
    const uint32_t z0 = a.x ^ pr.n3x4, z1 = a.y ^ pr.n3x4, z2 = nx ^ pr.n3x4;
    const uint32_t c0 = (a.x ^ pr.n0x4) | __builtin_amdgcn_alignbyte(y1, y0, pr.r) | __builtin_amdgcn_alignbyte(z1, z0, pr.r3);
    const uint32_t c1 = (a.y ^ pr.n0x4) | __builtin_amdgcn_alignbyte(y2, y1, pr.r) | __builtin_amdgcn_alignbyte(z2, z1, pr.r3);
    return acc | zero_byte_flags(c0) | zero_byte_flags(c1);
}

// two half-pieces (lo = bytes 0..511, hi = bytes 512..1023 of a piece, 8 bytes per lane) -> the piece in
// the 16-bytes-per-lane layout: lane l takes the two half-chunks 2*(l%32), 2*(l%32)+1 of half l/32.
__device__ __forceinline__ u32x4 transpose_halves(const u32x2 &lo, const u32x2 &hi, int lane)
{
    const int i0 = ((lane & 31) * 2) << 2, i1 = i0 + 4;
    const bool up = lane >= 32;
    u32x4 A;
    const uint32_t ax = (uint32_t)__builtin_amdgcn_ds_bpermute(i0, (int)lo.x), bx = (uint32_t)__builtin_amdgcn_ds_bpermute(i0, (int)hi.x);
    const uint32_t ay = (uint32_t)__builtin_amdgcn_ds_bpermute(i0, (int)lo.y), by = (uint32_t)__builtin_amdgcn_ds_bpermute(i0, (int)hi.y);
    const uint32_t az = (uint32_t)__builtin_amdgcn_ds_bpermute(i1, (int)lo.x), bz = (uint32_t)__builtin_amdgcn_ds_bpermute(i1, (int)hi.x);
    const uint32_t aw = (uint32_t)__builtin_amdgcn_ds_bpermute(i1, (int)lo.y), bw = (uint32_t)__builtin_amdgcn_ds_bpermute(i1, (int)hi.y);
    A.x = up ? bx : ax;
    A.y = up ? by : ay;
    A.z = up ? bz : az;
    A.w = up ? bw : aw;
    return A;
}

// lane l receives `cur` of lane l+k when l+k < 64, else `nxt` of lane l+k-64 (0 <= k <= 64):
// the value k lanes further along the concatenation {cur, nxt} of two consecutive pieces.
__device__ __forceinline__ uint32_t from_lane_ahead(uint32_t cur, uint32_t nxt, int lane, int k)
{
    const int idx = ((lane + k) & (kWave - 1)) << 2;
    const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)cur);
    const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)nxt);
    return lane + k < kWave ? a : b;
}

}  // namespace ss
