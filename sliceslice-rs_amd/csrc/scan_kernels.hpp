// scan_kernels.hpp - device side of the MI355X (gfx950, wave64) substring scan.
//
// What the reference does per 32 candidate offsets with two AVX2 loads, two vpcmpeqb, a vpand and a
// vpmovmskb (vector_search_in_chunk, /root/reference/src/lib.rs:199-251; __m256i ops
// src/x86.rs:202-235) is re-expressed here for a 64-lane wavefront:
//
//   * a *piece* is 64 consecutive 16-byte chunks (1 KiB, 1 KiB-aligned relative to the 16-B-aligned
//     base) of the haystack, one chunk per lane, fetched by ONE coalesced global_load_dwordx4 per lane.
//     Every load is an aligned chunk that contains at least one in-range byte, so no load can cross
//     into an unmapped page (the guarantee the reference gets from its overlapped tail chunk,
//     lib.rs:276-284).  Pieces do not overlap: HBM traffic == haystack bytes (+16 B per wave-tile).
//   * the filter works on byte DIFFERENCES, 4 bytes per VALU op: A ^ splat(b) has a zero byte where the haystack byte
//     equals needle byte b.  The differences of the further filter bytes are moved down the stream by their distance
//     from the first filter byte and OR-ed onto the first byte's differences, and ONE zero-byte test
//     z(x) = (x - 0x01010101) & ~x  per dword flags the offsets at which ALL of them match (bit 7 of every zero
//     byte of x is set; it can also flag a 0x01 byte sitting above a zero byte - a false POSITIVE only, and
//     candidates are verified, so the boolean is unaffected).
//   * distance of the second filter byte = 16*d + 4*Q + r.  d == 0 (MODE 0: what both constructors always pick):
//     THREE filter bytes.  The raw dwords of the next lane's chunk are moved once (DPP
//     wave_shl:1; the xor commutes with the move), Q selects the second byte's dword window at compile time, the
//     third byte's window is wave-uniform run-time data, the byte parts are v_alignbyte_b32.
//   * lane 63's neighbour is lane 0 of the NEXT piece.  A wave owns U consecutive pieces, so that is a
//     register of the same wave (one DPP wave_rol:1 feeds it in as the `old` operand of the wave_shl);
//     after the wave's last piece it is a single 16-byte halo chunk loaded by lane 63 alone.
//   * d > 0 (a pair 16 or more apart, ss_searcher_set_filter only), two filter bytes: MODE 2 keeps ONE non-temporal load stream and fetches the
//     position-byte differences from the lane that owns chunk c+d with ds_bpermute (d <= 62); MODE 1 (larger d)
//     issues a second, plain load stream at +d chunks.
//   * a tile (U pieces per wave) is filtered in one straight-line phase; `__ballot(any flag)` is the wave's
//     movemask: zero -> next tile.  Otherwise a second-level filter clears the flags where one of the remaining
//     bytes of the 32 behind the first filter byte differs, rarest byte first, still in registers, with a ballot
//     after each byte - on the one or two pieces that hold candidates, or tile-wide when most do.  A wave that STILL
//     has a candidate stages the needle in LDS, walks its flags lowest-first (`__ffs`, clear lowest set bit -
//     lib.rs:220-247) and compares 16 bytes per step; the first equal candidate sets the found flag (lib.rs:242-244).
//   * workgroups are short-lived (one or two tiles each): the hardware dispatcher hands out tiles in address
//     order.  Every workgroup but the first few peeks at the found flag through the scalar cache before it
//     loads anything, and every tile polls it coherently behind its data loads, so a hit stops the grid
//     early - the reference's early `return true`.  FIND kernels keep the leftmost match offset instead
//     (atomicMin) and skip only work that lies to the right of it.
//   * L8 kernels run the first phase on 8 bytes per lane (dwordx2 loads) and transpose only tiles with
//     candidates into the 16-byte layout (used for one-byte needles).
//
// Nothing here depends on block->XCD placement; all inter-workgroup traffic is one relaxed agent-scope int
// (or uint64 minimum), read with relaxed agent-scope loads and scalar-cache peeks.
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
    uint32_t pad_;
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
};
constexpr uint32_t kProblemCounted = 1u;

// Where a wave finds the COLD fields of its Problem.
struct ColdInKernarg {        // scan_kernel: the Problem is the kernel's FIRST argument, i.e. offset 0 of the kernarg segment
    typedef const Problem __attribute__((address_space(4))) *Ptr;
    __device__ __forceinline__ Ptr operator()() const
    {
        Ptr kp = (Ptr)__builtin_amdgcn_kernarg_segment_ptr();
        __asm__ volatile("" : "+s"(kp));     // opaque: the loads behind it stay where they are written
        return kp;
    }
};
struct ColdInRegisters {      // kernels that build their Problem themselves (batched)
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
// holds the chunk -> wave_rol), or the halo chunk already sitting in lane 63 (kind 0), or unknown
// (kind 2: lane 63 passes conservatively and is settled by the memory compare).
struct NextPiece {
    u32x4 N;
    int kind;     // wave-uniform
};

__device__ __forceinline__ uint32_t next_lane_diffs(uint32_t own, uint32_t nword, uint32_t nkx4, int kind)
{
    uint32_t last = 0u;                                           // what lane 63 will see: "matches" when unknown
    if (kind != 2) {
        const uint32_t f = nword ^ nkx4;
        last = kind == 1 ? rotate_from_next_lane(f) : f;
    }
    return from_next_lane_or(last, own);
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
            np.kind = u + 1 < U ? 1 : (MODE == 0 ? 0 : (MODE == 2 ? 1 : 2));
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
                                             const uint8_t *s_needle, uint64_t &where)
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
// NTMODE: 0 = plain loads; 1 = non-temporal loads (first-byte stream; the position-byte stream of MODE 1
// stays plain so that its re-read hits).
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

// MODE selects where the position-byte flags of a candidate come from (position = 16*d + 4*Q + r):
//   0  d == 0: same chunk / next lane (DPP) - every needle of <= 16 bytes with the default position;
//   1  d  > 0: a second load stream at +d chunks (plain loads; the re-read hits in L1/L2);
//   2  0 < d < 64, small: ONE (non-temporal) load stream; the flags computed by the lane that owns chunk
//      c+d are fetched across lanes with ds_bpermute (the wave loads d+1 halo chunks after its last piece).
// FIND = false: `sink` is the int found flag (0 -> 1).  FIND = true: `sink` is the uint64 leftmost-match
// offset (row f1 of SURVEY.md 8f: the `Option<usize>` shape of tests/i386.rs:6-10 and
// bench/sse4-strstr/src/lib.rs:4-15); a wave only skips work that lies to the RIGHT of the best so far.
// LAZY_ORDER (batched kernel): the descriptor arrives without the second-level schedule; a wave builds it
// when it first meets a candidate, next to staging the needle - not on every workgroup's way in.
// `pr` is read for its HOT fields only; `cold()` yields a pointer through which the cold ones are read where they are needed.
template <int Q, int MODE, bool ONE_BYTE, int U, int NTMODE, bool FIND = false, bool L8 = false, bool LAZY_ORDER = false,
          typename ColdT = ColdInRegisters>
__device__ __forceinline__ void scan_tiles(const Problem &pr, ColdT cold, uint8_t *s_needle_block, uint64_t tile0,
                                           uint64_t tile_step, uint64_t tile_end, void *sink, int *wg_found = nullptr)
{
    static_assert(!L8 || (MODE == 0 && !FIND), "the 8-byte layout covers the single-stream bool kernels");
    constexpr bool TWO = MODE == 1;
    constexpr bool SHIFTED = MODE == 2;
    int *found = static_cast<int *>(sink);
    uint64_t *best = static_cast<uint64_t *>(sink);
    constexpr bool NTA = NTMODE >= 1;
    constexpr bool NTB = TWO ? NTMODE >= 2 : NTMODE >= 1;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);   // wave-uniform -> SGPR
    const int wpb = (int)(blockDim.x / kWave);                              // waves per workgroup (launch-time)
    uint8_t *s_needle = s_needle_block + wave * kNeedleLds;
    bool staged = false, ordered = false;
    const bool small_grid = (pr.flags & kProblemCounted) != 0;             // completion-word launch: nobody peeks
#ifdef SS_TWO_BYTE_PHASE1       // A/B builds only (sliceslice_rs_amd._build.build_ab + tools/ab_inproc.py): the round-1 two-byte first phase
    constexpr bool THREE = false;
#else
    constexpr bool THREE = MODE == 0 && !ONE_BYTE;                          // three-byte first phase
#endif
    // The cold part of the problem, fetched by a wave when it first meets a candidate (`ordered`): what the verification
    // needs, the second-level schedule, and the needle's dwords for the exact in-register verification
    // (exact_verify_piece: the single-stream multi-byte kernels, needles that end at most 16 bytes behind the first filter byte).
    RefineOrder ro = {0, {0, 0}, {0, 0}};
    VerifyArgs va = {nullptr, nullptr, 0, 0};
    constexpr bool EXACT_OK = MODE == 0 && !ONE_BYTE;
    uint32_t tail16[4] = {0, 0, 0, 0};
    uint32_t exact_len = 0u;
    bool dense = false;                                                     // L8: the previous tile had candidates
    const int d = (int)pr.d;                                                // SHIFTED: 1 <= d <= 62

    for (uint64_t tile = tile0; tile < tile_end; tile += tile_step) {
        u32x4 A[U], B[U], H = {0, 0, 0, 0};
        const uint64_t chunk0 = (tile * (uint64_t)(wpb * U) + (uint64_t)wave * U) * 64;   // wave-uniform
        // FIND polls first (oldest load, so waiting for it does not drain the data loads behind it); the value
        // is only made wave-uniform (readfirstlane = the wait) after the tile's data loads have been issued
        const uint64_t best_raw = FIND ? __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        uint64_t best_now = 0;
        // last chunk this wave touches: the halo chunk (MODE 0/1) or the d+1 halo chunks (MODE 2)
        const uint64_t halo = chunk0 + 64 * U + pr.d;
        const bool full = halo < pr.nchunks_all;
        bool have16 = false;
        if (L8 && !dense) {
            // ---- 8 bytes per lane: 2U half-pieces + a 16-byte halo in lanes 0 and 1 ----
            u32x2 Hh[2 * U], halo8 = {0, 0};
            const uint64_t half0 = chunk0 * 2;
            if (full) {
#pragma unroll
                for (int h = 0; h < 2 * U; ++h) Hh[h] = load_half<NTA>(pr.base, half0 + 64 * h + lane);
                if (!ONE_BYTE && lane < 2) halo8 = load_half<false>(pr.base, half0 + 128 * U + lane);
            } else {
#pragma unroll
                for (int h = 0; h < 2 * U; ++h) {
                    const uint64_t hc = half0 + 64 * h + lane;
                    Hh[h] = u32x2{0, 0};
                    if ((hc >> 1) < pr.nchunks_all) Hh[h] = load_half<NTA>(pr.base, hc);
                }
                if (!ONE_BYTE && lane < 2 && halo < pr.nchunks_all) halo8 = load_half<false>(pr.base, half0 + 128 * U + lane);
            }
            const int stop8 = poll_found(found, pr.epoch);
            uint32_t any8 = 0;
            u32x2 tc = {Hh[0].x ^ pr.nlx4, Hh[0].y ^ pr.nlx4}, tn = {0, 0};
#pragma unroll
            for (int h = 0; h < 2 * U; ++h) {
                if (!ONE_BYTE) {
                    const u32x2 nx = h + 1 < 2 * U ? Hh[h + 1] : halo8;
                    tn = u32x2{nx.x ^ pr.nlx4, nx.y ^ pr.nlx4};
                }
                any8 = filter_half<Q, ONE_BYTE>(Hh[h], tc, tn, pr, lane, any8);
                tc = tn;
            }
            if (stop8) {                                      // somebody has already found the needle
                forget_scalar_cache_unless(small_grid);
                return;
            }
            if (__ballot((any8 & 0x80808080u) != 0) == 0) continue;   // nothing in this tile: the common case
            // candidates: bring the tile into the 16-bytes-per-lane layout for the second phase
#pragma unroll
            for (int u = 0; u < U; ++u) A[u] = transpose_halves(Hh[2 * u], Hh[2 * u + 1], lane);
            if (!ONE_BYTE && lane == kWave - 1 && halo < pr.nchunks_all) H = load_chunk<false>(pr.base, halo);
            have16 = true;
        }
        // Loads + phase 1 are instantiated once per load shape (in place after the L8 transposition /
        // unconditional / predicated tail) so that the common, unconditional copy gets exact s_waitcnt
        // vmcnt(k) values: with the shapes merged at a control-flow join the compiler waited for ALL loads
        // of the tile before the first flag computation.
        uint32_t G[U][4];
        uint32_t any_tile = 0;
        int stop = 0;
        auto load_and_filter = [&](auto loaded_c, auto full_c, auto q3_c) {
            constexpr bool LOADED = decltype(loaded_c)::value, FULL = decltype(full_c)::value;
            constexpr int Q3 = decltype(q3_c)::value;
            if constexpr (!LOADED && FULL) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    A[u] = load_chunk<NTA>(pr.base, chunk0 + 64 * u + lane);
                    if (TWO) B[u] = load_chunk<NTB>(pr.base, chunk0 + 64 * u + lane + pr.d);
                }
                if (SHIFTED) {
                    if (lane <= d) H = load_chunk<false>(pr.base, chunk0 + 64 * U + lane);
                } else if (!ONE_BYTE && lane == kWave - 1) {
                    H = load_chunk<false>(pr.base, halo);
                }
            } else if constexpr (!LOADED) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint64_t c = chunk0 + 64 * u + lane;
                    A[u] = u32x4{0, 0, 0, 0};
                    if (c < pr.nchunks_all) A[u] = load_chunk<NTA>(pr.base, c);
                    if (TWO) {
                        B[u] = u32x4{0, 0, 0, 0};
                        if (c + pr.d < pr.nchunks_all) B[u] = load_chunk<NTB>(pr.base, c + pr.d);
                    }
                }
                if (SHIFTED) {
                    if (lane <= d && chunk0 + 64 * U + lane < pr.nchunks_all) H = load_chunk<false>(pr.base, chunk0 + 64 * U + lane);
                } else if (!ONE_BYTE && lane == kWave - 1 && halo < pr.nchunks_all) {
                    H = load_chunk<false>(pr.base, halo);
                }
            }
            // the poll is issued behind the data loads and consumed after them
            stop = FIND ? 0 : poll_found(found, pr.epoch);
            if (FIND) best_now = uniform64(best_raw);

            // ---- phase 1: the two-byte filter for all U pieces, straight-line (loads are consumed in order) ----
            uint32_t wcur[4] = {0, 0, 0, 0}, wnext[4] = {0, 0, 0, 0}, wlast[4] = {0, 0, 0, 0};
            if (!ONE_BYTE && !THREE) position_diffs(TWO ? B[0] : A[0], pr.nlx4, wcur);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t *g = G[u];
                if (THREE) {
                    // lane 63's next lane: lane 0 of the next piece (raw dwords, rotated in), or the halo chunk
                    constexpr int QM = Q > Q3 ? Q : Q3;
                    uint32_t nx[4] = {0, 0, 0, 0};
                    if (u + 1 < U) {
                        nx[0] = rotate_from_next_lane(A[u + 1].x);
                        if (QM >= 1) nx[1] = rotate_from_next_lane(A[u + 1].y);
                        if (QM >= 2) nx[2] = rotate_from_next_lane(A[u + 1].z);
                        if (QM >= 3) nx[3] = rotate_from_next_lane(A[u + 1].w);
                    } else {
                        nx[0] = H.x; nx[1] = H.y; nx[2] = H.z; nx[3] = H.w;
                    }
                    filter_piece3<Q, Q3>(A[u], nx, pr, g);
                } else if (SHIFTED) {
                    // flags of the following piece (or of the halo chunks), then the 8-dword window by lane distance
                    position_diffs(u + 1 < U ? A[u + 1] : H, pr.nlx4, wnext);
                    uint32_t x[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        x[j] = (j >= Q) ? from_lane_ahead(wcur[j], wnext[j], lane, d) : 0u;
                        x[4 + j] = (j <= Q) ? from_lane_ahead(wcur[j], wnext[j], lane, d + 1) : 0u;
                    }
                    g[0] = zero_byte_flags((A[u].x ^ pr.n0x4) | __builtin_amdgcn_alignbyte(x[Q + 1], x[Q + 0], pr.r));
                    g[1] = zero_byte_flags((A[u].y ^ pr.n0x4) | __builtin_amdgcn_alignbyte(x[Q + 2], x[Q + 1], pr.r));
                    g[2] = zero_byte_flags((A[u].z ^ pr.n0x4) | __builtin_amdgcn_alignbyte(x[Q + 3], x[Q + 2], pr.r));
                    g[3] = zero_byte_flags((A[u].w ^ pr.n0x4) | __builtin_amdgcn_alignbyte(x[Q + 4], x[Q + 3], pr.r));
                } else {
                    if (!ONE_BYTE) {
                        // lane 63's "next lane": lane 0 of the next piece, or the halo chunk after the last piece
                        if (u + 1 < U) {
                            position_diffs(TWO ? B[u + 1] : A[u + 1], pr.nlx4, wnext);
#pragma unroll
                            for (int j = 0; j < 4; ++j) wlast[j] = (j <= Q) ? rotate_from_next_lane(wnext[j]) : 0u;
                        } else {
                            position_diffs(H, pr.nlx4, wlast);
                        }
                    }
                    filter_piece<Q, ONE_BYTE>(A[u], wcur, wlast, pr, g);
                }
                any_tile |= g[0] | g[1] | g[2] | g[3];
                if (!ONE_BYTE && !THREE && (SHIFTED || u + 1 < U)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) wcur[j] = wnext[j];
                }
            }

        };
        // the third byte's dword window is wave-uniform run-time data: one copy of the phase per window
        auto run_phase1 = [&](auto loaded_c, auto full_c) {
            if constexpr (THREE) {
                // Whoever builds the Problem orders the two further bytes so that q3 <= Q (they are interchangeable): the
                // copies with Q3 > Q are never taken.  They stay instantiated all the same: with them pruned the register
                // allocator needed 146 VGPRs instead of 121 for Q < 3 (build() records every kernel's registers in csrc/kernel_resources.json; tests/test_bindings_cpu.py keeps an eye on it).
                switch (pr.q3) {
                case 0: load_and_filter(loaded_c, full_c, std::integral_constant<int, 0>{}); break;
                case 1: load_and_filter(loaded_c, full_c, std::integral_constant<int, 1>{}); break;
                case 2: load_and_filter(loaded_c, full_c, std::integral_constant<int, 2>{}); break;
                default: load_and_filter(loaded_c, full_c, std::integral_constant<int, 3>{}); break;
                }
            } else {
                load_and_filter(loaded_c, full_c, std::integral_constant<int, 0>{});
            }
        };
        if (have16) run_phase1(std::true_type{}, std::true_type{});
        else if (full) run_phase1(std::false_type{}, std::true_type{});
        else run_phase1(std::false_type{}, std::false_type{});
        if (FIND) {
            const uint64_t first = chunk0 * 16 > pr.mis ? chunk0 * 16 - pr.mis : 0;   // lowest index this wave can report
            if (best_now <= pr.find_base + first) {                                     // all of it lies right of a match
                forget_scalar_cache_unless(small_grid);
                return;
            }
        }

        // ---- phase 2 (rare on random bytes): the wave's "movemask != 0" ---------------------------------
        const bool cand_tile = __ballot((any_tile & 0x80808080u) != 0) != 0;
        if (L8) dense = cand_tile;      // stay in the 16-byte layout while tiles keep producing candidates
        if (stop) {                     // somebody has already found the needle: no point in verifying more
            forget_scalar_cache_unless(small_grid);
            return;
        }
        if (cand_tile) {
            // Kernels whose Problem sits in the kernarg segment re-read the cold fields for EVERY tile with candidates (scalar
            // cache hits, the lines were touched at entry) instead of carrying ~25 scalar registers from tile to tile: carried,
            // they pushed as many loop invariants out to vector lanes in front of every workgroup's first load.  Kernels that
            // have to BUILD the schedule (LAZY_ORDER) do it once per wave.
            if (!LAZY_ORDER || !ordered) {
                const auto c = cold();
                va.hay = reinterpret_cast<const uint8_t *>(uniform64((uint64_t)(uintptr_t)c->hay));
                va.needle = reinterpret_cast<const uint8_t *>(uniform64((uint64_t)(uintptr_t)c->needle));
                va.n = uniform64(c->n);
                va.end = uniform64(c->end);
                if (!ONE_BYTE && !LAZY_ORDER) {
                    ro.n = c->norder;
                    ro.idx[0] = c->order_idx[0]; ro.idx[1] = c->order_idx[1];
                    ro.val[0] = c->order_val[0]; ro.val[1] = c->order_val[1];
                    if (EXACT_OK) {
                        exact_len = c->exact_len;
#pragma unroll
                        for (int j = 0; j < 4; ++j) tail16[j] = c->tail16[j];
                    }
                }
                if (!ONE_BYTE && LAZY_ORDER) {
                    // the descriptor came without the schedule (and without the needle's dwords): built here, by the waves
                    // that need them, not on every workgroup's way in
                    const uint64_t position = pr.d * 16 + 4 * Q + pr.r;
                    const uint64_t position3 = THREE ? (uint64_t)(4 * pr.q3 + pr.r3) : ~0ull;
                    const uint64_t anchor = (uint64_t)((pr.base + pr.mis) - va.hay);      // index of the first filter byte
                    ro.n = (uint32_t)__builtin_amdgcn_readfirstlane(
                        (int)build_refine_order_wave(va.needle + anchor, va.n - anchor, position, lane, ro.idx, ro.val, position3));
                    for (int t = 0; t < 2; ++t) {
                        ro.idx[t] = uniform64(ro.idx[t]);
                        ro.val[t] = uniform64(ro.val[t]);
                    }
                    if (EXACT_OK && va.n - anchor <= 16) {
                        // the bytes from the first filter byte on, plus what sixteen leave room for of those in front of it
                        const uint32_t behind = (uint32_t)(va.n - anchor);
                        const uint32_t back = (uint32_t)(anchor < 16 - behind ? anchor : 16 - behind);
                        const uint32_t el = behind + back;
                        exact_len = el | (back << 8);
                        const uint32_t nbv = (uint32_t)lane < el ? (uint32_t)va.needle[anchor - back + lane] : 0u;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            tail16[j] = ((uint32_t)__builtin_amdgcn_readlane((int)nbv, 4 * j) & 0xFF) |
                                        (((uint32_t)__builtin_amdgcn_readlane((int)nbv, 4 * j + 1) & 0xFF) << 8) |
                                        (((uint32_t)__builtin_amdgcn_readlane((int)nbv, 4 * j + 2) & 0xFF) << 16) |
                                        (((uint32_t)__builtin_amdgcn_readlane((int)nbv, 4 * j + 3) & 0xFF) << 24);
                    }
                }
                ordered = true;
            }
            // The needle is staged into LDS only by a wave that still has a candidate AFTER the in-register
            // second-level filter (next to nobody, on random bytes and on text alike): staging costs a pass over
            // min(n, 2 KiB) needle bytes, which at 2^-16 candidates per offset and short-lived workgroups made a
            // 2000-byte needle 13 % slower than a 16-byte one.
            auto stage_once = [&]() {
                if (!staged && !ONE_BYTE) {
                    stage_needle_wave(s_needle, va.needle, va.n, lane);
                    staged = true;
                }
            };
            // second-level filter in registers (wave-uniform), up to the first 16 needle bytes, tile-wide
#ifndef SS_MODE2_TILE_WIDE
#define SS_MODE2_TILE_WIDE 1
#endif
            constexpr bool TILE_WIDE = MODE != 2 || SS_MODE2_TILE_WIDE != 0;
            // Which pieces of the tile hold candidates?  With the three-byte first phase a tile that gets here
            // usually holds ONE (text: a frequent phrase that shares the filter bytes); the second-level filter
            // then runs on that piece alone instead of on all U - a quarter of the work.  Tiles dense with
            // candidates (a caller-chosen pair of common bytes) keep the tile-wide form, whose scalar bookkeeping is
            // paid once per needle byte instead of once per piece and byte.
            uint32_t pm = (1u << U) - 1;
            bool per_piece = !TILE_WIDE;
            uint32_t cand_lanes = 0;                    // lanes of the tile that hold a candidate
            if (!ONE_BYTE && TILE_WIDE) {
                pm = 0;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint64_t bl = __ballot(((G[u][0] | G[u][1] | G[u][2] | G[u][3]) & 0x80808080u) != 0);
                    if (bl != 0) pm |= 1u << u;
                    cand_lanes += (uint32_t)__builtin_popcountll(bl);
                }
#ifdef SS_NO_SPARSE_REFINE       // A/B builds only
                per_piece = false;
#else
                per_piece = __builtin_popcount(pm) <= 2;
#endif
                if (!per_piece) {
                    // With the needle's dwords at hand (exact mode) the byte-wise schedule only has to thin out CHANCE hits - two
                    // bytes do that - because the exact compare settles whatever is left, many candidates per lane or few; a true
                    // match survives every step of the schedule, and a text full of them (the reference's bench: words of the
                    // manual searched in the manual) paid for all of its up to 15 ballot rounds in every tile: 5.7 us per search
                    // for a rare word, 10-11 us for 'instruction' (profiles/r03/service_experiments.md).
                    // ... and with few candidate lanes in the tile (every piece of this text holds a true match or two) not
                    // even those: the compare costs a lane ~40 operations per candidate, a schedule byte ~24 per PIECE.
                    const bool exact = EXACT_OK && exact_len != 0;
                    const uint32_t max_steps = exact ? (cand_lanes <= kExactSparseLanes ? 0u : kExactRefineSteps) : 15u;
                    if (max_steps != 0 && !refine_tile<U, MODE>(A, H, ro, G, max_steps)) continue;
                }
            }
            bool hit = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t *g = G[u];
                if (!ONE_BYTE && per_piece) {
                    if (((pm >> u) & 1u) == 0) continue;
                    NextPiece np;
                    np.N = u + 1 < U ? A[u + 1] : H;
                    // lane 63's next lane: lane 0 of the next piece (rotated in); after the last piece the halo chunk
                    // sitting in lane 63 (MODE 0), lanes 0..d of H (MODE 2), or unknown (MODE 1: settled by the compare)
                    np.kind = u + 1 < U ? 1 : (MODE == 0 ? 0 : (MODE == 2 ? 1 : 2));
                    // With the needle's dwords at hand the exact compare below settles a lane's candidates in ~50 VALU
                    // operations, all lanes at once - about what TWO steps of the byte-wise schedule cost - and a true match
                    // would sit through every one of its up to 13 steps first (a microsecond of ballots and branches).
                    if (!(EXACT_OK && exact_len != 0) && !refine_piece(A[u], np, ro, g)) continue;
                } else if (__ballot(((g[0] | g[1] | g[2] | g[3]) & 0x80808080u) != 0) == 0) {
                    continue;
                }
                uint64_t where = 0;
                bool h;
                if (EXACT_OK && exact_len != 0) {               // wave-uniform: the needle's dwords are at hand
                    NextPiece np;
                    np.N = u + 1 < U ? A[u + 1] : H;
                    np.kind = u + 1 < U ? 1 : 0;                // MODE 0: the halo chunk sits in lane 63
                    uint32_t where_off = 0;
                    h = exact_verify_piece(A[u], np, g, chunk0 + 64 * u, lane, pr, va, tail16, exact_len, where_off);
                    if (FIND) where = (chunk0 + 64 * u) * 16 - pr.mis + where_off;
                } else {
                    stage_once();
                    h = verify_flags<ONE_BYTE>(g, chunk0 + 64 * u + lane, pr, va, s_needle, where);
                }
                hit |= h;
                // search_in: the first piece with a match settles the wave (a text full of matches holds one in every piece)
                if (!FIND && __ballot(h) != 0) break;
                if (FIND) {
                    const uint64_t m = __ballot(h);
                    if (m != 0) {                       // lanes are in address order: lowest lane = leftmost
                        const int src = __ffsll((unsigned long long)m) - 1;
                        const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)where, src);
                        const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(where >> 32), src);
                        const uint64_t mine = pr.find_base + (((uint64_t)hi << 32) | lo);
                        // only a wave that can actually lower the minimum touches it (matches everywhere
                        // would otherwise serialise one atomic per wave on a single address)
                        if (lane == 0 && mine < __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                            __hip_atomic_fetch_min(best, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        forget_scalar_cache_unless(small_grid);
                        return;                         // the wave's later pieces and tiles are further right
                    }
                }
            }
            if (!FIND) {
                const uint64_t hits = __ballot(hit);
                if (hits != 0) {
                    // ONE lane of the wave publishes, and only the wave that flips the device flag writes the
                    // pinned-host mirror: a needle that occurs everywhere would otherwise have every wave of
                    // the grid queue a system-scope store to the same host address (measured: 14 ms for a
                    // one-byte needle over 1 GiB instead of 0.02 ms).
                    if (wg_found != nullptr) {
                        // Completion-word launches (grids of at most 256 workgroups, all of them resident from the start):
                        // the answer travels in the workgroup count (scan_kernel's epilogue) and there is nobody left to
                        // stop early, so the device flag is not even written - a global store in front of the count-out
                        // atomic of the same wave is a memory round trip on the path a match's latency is made of.
                        if (lane == __ffsll((unsigned long long)hits) - 1)
                            __hip_atomic_store(wg_found, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else if (lane == __ffsll((unsigned long long)hits) - 1 &&
                               __hip_atomic_load(found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != pr.epoch) {
                        const int old = __hip_atomic_exchange(found, pr.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        int *host_flag = cold()->host_flag;
                        if (old != pr.epoch && host_flag)
                            __hip_atomic_store(host_flag, pr.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    forget_scalar_cache_unless(small_grid);
                    return;
                }
            }
        }
    }
}

// ---- K1/K2/K3: one needle, one haystack ---------------------------------------------------------
// gridDim.x workgroups; workgroup b scans tiles [b*tiles_per_block, (b+1)*tiles_per_block) when
// tiles_per_block > 0 (contiguous runs, short-lived workgroups), or b, b+grid, ... when it is 0.
// Four waves per SIMD (<= 128 VGPRs) is what the shipped U = 4 kernels need: at three they run at 6.3 instead of 7.4 TB/s
// (profiles/r02/ab_filter_triples.jsonl).  The allocator lands on 121-123 by itself; asking for it with
// amdgpu_waves_per_eu(4, 4) makes it fill all 128 and spill two registers in the cross-lane kernels, so the build records
// every kernel's registers and occupancy instead (csrc/kernel_resources.json, checked by tests/test_bindings_cpu.py).
// -DSS_WAVES_PER_EU=5 asks for <= 96 VGPRs (occupancy experiments).
#ifdef SS_WAVES_PER_EU
#define SS_SCAN_OCCUPANCY __attribute__((amdgpu_waves_per_eu(SS_WAVES_PER_EU, SS_WAVES_PER_EU)))
#else
#define SS_SCAN_OCCUPANCY
#endif
template <int Q, int MODE, bool ONE_BYTE, int U, int NTMODE, bool FIND = false, bool L8 = false>
__global__ void SS_SCAN_OCCUPANCY __launch_bounds__(kMaxBlock) scan_kernel(const Problem pr, void *found, uint64_t tiles_per_block)
{
    // one 2 KiB slice per wave; the launch passes (waves per workgroup) * kNeedleLds bytes of dynamic LDS
    extern __shared__ __attribute__((aligned(16))) uint8_t s_needle[];
    // Early exit survives short-lived workgroups through the entry peek (scalar cache; see scalar_peek): once
    // a match is known the rest of the grid drains without touching memory, so the peek comes before anything
    // else.  The first workgroups of a grid start before anything can have been found and skip it.
    // Pieces per tile = waves per workgroup (2, 4 or 8) * U: a power of two, so no division anywhere.
    static_assert((U & (U - 1)) == 0, "U is a power of two");
    // The cold half of the Problem is read from the kernarg segment where it is needed (ColdInKernarg).  Its two cache lines
    // are TOUCHED here, next to the hot loads, so that a wave that meets a candidate finds them in the scalar cache instead of
    // paying a memory round trip on the path a match's latency is made of (a 1 KiB haystack with the needle at 0: 11.3 us
    // per call against 8.1 for an absent needle before this).  Two throw-away registers until the first wait below.
    static_assert(sizeof(Problem) <= 0x100 && offsetof(Problem, hay) < 0x80, "the cold fields live in the lines at 0x80 and 0xc0");
    uint32_t touch0, touch1;
    {
        ColdInKernarg::Ptr kp = (ColdInKernarg::Ptr)__builtin_amdgcn_kernarg_segment_ptr();
        __asm__ volatile("s_load_dword %0, %2, 0x80\n\ts_load_dword %1, %2, 0xc0" : "=&s"(touch0), "=&s"(touch1) : "s"(kp));
    }
    const unsigned tile_shift = (unsigned)__builtin_ctz(blockDim.x / kWave) + (unsigned)__builtin_ctz(U);
    uint64_t t0 = tiles_per_block ? (uint64_t)blockIdx.x * tiles_per_block : blockIdx.x;
    // completion-word launches of the bool kernels: "a wave of this workgroup has found the needle"
    __shared__ int s_wg_found;
    const bool counted = !FIND && (pr.flags & kProblemCounted) != 0;   // wave-uniform (kernel argument)
    if (counted) {
        if (threadIdx.x == 0) __hip_atomic_store(&s_wg_found, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
    }
    // (the hot fields have been waited for by now - `counted` is one - and scalar loads are waited for together)
    __asm__ volatile("s_waitcnt lgkmcnt(0)" : : "s"(touch0), "s"(touch1));
    bool skip = false;
    if (blockIdx.x >= kPeekFromBlock) {
        // a peek hit is confirmed with one coherent load before the workgroup leaves: the scalar cache is not
        // coherent, and a caller-owned sink (re-armed by the caller, e.g. on every hipGraph replay) has no
        // epoch that would make a line cached by an earlier launch harmless
        if (FIND) {
            const uint64_t first_chunk = (t0 << tile_shift) * 64;
            const uint64_t first = first_chunk * 16 > pr.mis ? first_chunk * 16 - pr.mis : 0;
            skip = scalar_peek64(static_cast<const uint64_t *>(found)) <= pr.find_base + first &&
                   uniform64(__hip_atomic_load(static_cast<const uint64_t *>(found), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) <=
                       pr.find_base + first;
        } else {
            skip = scalar_peek(static_cast<const int *>(found)) == pr.epoch && poll_found(static_cast<const int *>(found), pr.epoch);
        }
    }
    if (!skip) {
        const uint64_t ntiles = (pr.npieces + ((uint64_t)1 << tile_shift) - 1) >> tile_shift;
        // one call site (one copy of the code): contiguous run, or grid-stride when tiles_per_block == 0
        uint64_t step = gridDim.x, t1 = ntiles;
        if (tiles_per_block) {
            step = 1;
            t1 = t0 + tiles_per_block < ntiles ? t0 + tiles_per_block : ntiles;
        }
        scan_tiles<Q, MODE, ONE_BYTE, U, NTMODE, FIND, L8, false, ColdInKernarg>(pr, ColdInKernarg{}, s_needle, t0, step, t1, found,
                                                                                 counted ? &s_wg_found : nullptr);
    }
    if (counted) {
        // Completion word of the bool kernels.  Every workgroup counts itself out with ONE relaxed 64-bit atomic add that
        // also carries "found here" in the high half, so the workgroup that brings the low half to done_target knows the
        // answer from the sum: no flag to read back, nothing to order, nothing to reset.  The barrier is a bare s_barrier
        // behind an lgkmcnt(0) wait: only the LDS word has to be settled, not the global store of the early-exit flag.
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);                         // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __asm__ volatile("" ::: "memory");
        if (threadIdx.x == 0) {
            const auto c = ColdInKernarg{}();
            const unsigned long long f = (unsigned long long)__hip_atomic_load(&s_wg_found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned long long mine = 1ull + (f << 32);
            const unsigned long long total = __hip_atomic_fetch_add(c->done_counter, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + mine;
            if ((uint32_t)total == c->done_target) {
                const uint32_t hi = (uint32_t)(total >> 32);
                const long long word = (long long)(((unsigned long long)hi << 32) | ((unsigned long long)(uint32_t)pr.epoch << 1) |
                                                   (hi != c->done_hi ? 1ull : 0ull));
                __hip_atomic_store(c->host_done, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    } else if (FIND && (pr.flags & kProblemCounted) != 0) {
        // Completion word of find(): the word is the answer itself - leftmost offset + 1, or all ones for "absent" (the
        // host zeroes it before the launch).  A wave's atomicMin has no return value, and the barrier alone does not wait
        // for it (gfx950 lowers __syncthreads() to `s_waitcnt lgkmcnt(0); s_barrier` - no vmcnt): every wave therefore
        // drains its own vector-memory queue first.  vmcnt also counts no-return atomics on gfx9-class parts, and a
        // device-scope atomic is acknowledged only once it has been performed beyond the XCD's L2, so after the wait the
        // minimum is where the count-out atomic of thread 0 - and the reader behind it - will look for it.
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const auto c = ColdInKernarg{}();
            const unsigned long long total = __hip_atomic_fetch_add(c->done_counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
            if ((uint32_t)total == c->done_target) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                // The slot is not re-armed either: find_base carries a per-launch key in the bits above kFindOffsetBits
                // that is SMALLER for every later launch on the slot, so whatever an earlier launch left behind loses
                // every atomicMin and reads as "absent" here.
                const uint64_t v = __hip_atomic_load(static_cast<const uint64_t *>(found), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint64_t off = v - pr.find_base;
                const bool hit = v >= pr.find_base && off < (1ull << kFindOffsetBits);
                __hip_atomic_store(c->host_done, hit ? (long long)(off + 1) : -1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

#ifdef SS_MISC_KERNELS   // only the API translation unit (sliceslice_hip.hip) compiles what follows

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
constexpr int kBadPosition = -1;   // SS_BATCH_BAD_POSITION: flag of a problem whose position breaks the with_position rules

// waves_per_eu(4, 4): without it the allocator ends at 129 VGPRs - one over the 128 that four waves per SIMD allow - and the
// kernel runs three workgroups per CU instead of four.
template <int U>
__global__ void __attribute__((amdgpu_waves_per_eu(4, 4))) __launch_bounds__(kBlock) scan_batched_kernel(const BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_needle[kWavesPerBlock * kNeedleLds];
    const uint64_t prob = blockIdx.x;
    const uint32_t slice = blockIdx.y, nslices = gridDim.y;
    if (slice != 0 && poll_found(a.found + prob, 1)) return;
    const uint64_t h0 = a.hay_begin[prob], h1 = a.hay_end[prob];
    const uint64_t n0 = a.needle_begin[prob], n1 = a.needle_end[prob];
    const uint64_t len = h1 - h0, n = n1 - n0;
    int *found = a.found + prob;
    if (n == 0) {                                   // N0: found everywhere (x86.rs:500)
        if (slice == 0 && threadIdx.x == 0) publish_found(found);
        return;
    }
    uint64_t position = a.position ? a.position[prob] : n - 1;
    if (n == 1 ? position != 0 : position >= n) {   // the reference panics building this searcher (x86.rs:300, 473)
        if (slice == 0 && threadIdx.x == 0) publish_found(found, kBadPosition);
        return;
    }
    if (len < n) return;                            // flag stays 0

    // A `position` 16 or more behind needle[0] keeps its byte in the filter but gets a partner at most 15 in front of it
    // instead of needle[0] (same rule as the host's choose_anchor, coarser ranking): lane K ranks
    // needle[position - 15 + K]; the rarest class wins, the byte closest to `position` within it.  The filter then
    // works in the coordinates of hay + anchor, and every problem runs in the single-stream kernel.
    const int lane = threadIdx.x & (kWave - 1);
    const uint8_t *needle = a.needles + uniform64(n0);
    position = uniform64(position);
    uint64_t anchor = 0;
    if (position >= 16) {
        const bool valid = lane < 15;
        const int rk = valid ? byte_rarity_rank(needle[position - 15 + lane]) : 0;
        const int cls = !valid ? -1 : (rk < 64 ? 0 : (rk < 128 ? 1 : (rk < 192 ? 2 : 3)));
        uint32_t pick = 1;
#pragma unroll
        for (int c = 3; c >= 0; --c) {
            const uint32_t m = (uint32_t)__ballot(cls == c) & 0x7FFFu;
            if (m) pick = m;                                  // ends up as the lowest non-empty class
        }
        anchor = position - 15 + (31u - (uint32_t)__builtin_clz(pick));
    }

    // every field below is wave-uniform; uniform64 tells the compiler so (SGPRs, no scratch)
    Problem pr;
    pr.hay = a.haystacks + uniform64(h0);
    const uint8_t *hf = pr.hay + anchor;
    pr.mis = (uint32_t)((uintptr_t)hf & 15);
    pr.base = hf - pr.mis;
    pr.n = uniform64(n);
    pr.end = uniform64(len - n + 1);
    pr.nchunks_all = (pr.mis + uniform64(len) - anchor + 15) / 16;
    pr.npieces = ((pr.mis + pr.end + 15) / 16 + 63) / 64;
    // contiguous run of tiles per slice (same launch shape as the single-problem kernel); surplus slices
    // of a short haystack leave before anything else that depends on the needle is loaded
    const uint64_t ntiles = (pr.npieces + kWavesPerBlock * U - 1) / (kWavesPerBlock * U);
    // The host sizes the grid from the problem count alone (the lengths live here); a workgroup that got less than
    // kBatchMinTiles tiles would spend more time on its start-up chain (ranges -> needle bytes -> first haystack load) than
    // on the scan, so short haystacks are cut into fewer, longer slices and the slices left over leave right here.
    uint64_t eff = (ntiles + kBatchMinTiles - 1) / kBatchMinTiles;
    eff = eff < nslices ? (eff ? eff : 1) : nslices;
    const uint64_t per = (ntiles + eff - 1) / eff;
    const uint64_t t0 = (uint64_t)slice * per;
    const uint64_t te = t0 + per < ntiles ? t0 + per : ntiles;
    if (t0 >= te) return;

    pr.needle = needle;
    uint32_t s = (uint32_t)(position - anchor);                // distance between the two filter bytes: 0 .. 15
    pr.d = 0;
    // ONE load serves all three filter bytes and the ranking: lane K holds needle[anchor + K], K < 16 (every further
    // dependent load is a round trip a short-lived workgroup spends before its first haystack byte is requested)
    const int lim = n - anchor < 16 ? (int)(n - anchor) : 16;
    const uint32_t nb = lane < lim ? (uint32_t)needle[anchor + lane] : 0u;
    pr.n0x4 = 0x01010101u * (uint32_t)__builtin_amdgcn_readlane((int)nb, 0);
    // third first-phase byte: the rarest of the 15 bytes behind the anchor other than needle[position],
    // later bytes winning ties; four rarity classes are tried in turn
    uint32_t p3 = s;
    if (n - anchor >= 3) {
        const bool valid = lane >= 1 && lane < lim && (uint32_t)lane != s;
        const int rk = valid ? byte_rarity_rank((uint8_t)nb) : 0;
        const int cls = !valid ? -1 : (rk < 64 ? 0 : (rk < 128 ? 1 : (rk < 192 ? 2 : 3)));
        uint32_t pick = 0;
#pragma unroll
        for (int c = 3; c >= 0; --c) {
            const uint32_t m = (uint32_t)__ballot(cls == c) & 0xFFFFu;
            if (m) pick = m;                                  // ends up as the lowest non-empty class
        }
        if (pick) p3 = 31u - (uint32_t)__builtin_clz(pick);
    }
    if (p3 / 4 > s / 4) {                           // the kernels want the third byte's dword not behind the second's
        const uint32_t t = p3;
        p3 = s;
        s = t;
    }
    pr.r = s % 4;
    pr.nlx4 = 0x01010101u * (uint32_t)__builtin_amdgcn_readlane((int)nb, (int)s);
    pr.q3 = p3 / 4;
    pr.r3 = p3 % 4;
    pr.n3x4 = 0x01010101u * (uint32_t)__builtin_amdgcn_readlane((int)nb, (int)p3);
    // the second-level schedule is built lazily by the waves that need it (scan_tiles<..., LAZY_ORDER>)
    pr.norder = 0;
    pr.order_idx[0] = pr.order_idx[1] = pr.order_val[0] = pr.order_val[1] = 0;
    pr.find_base = 0;
    pr.host_flag = nullptr;
    pr.epoch = 1;
    pr.done_counter = nullptr;
    pr.host_done = nullptr;
    pr.done_target = pr.done_hi = 0;
    pr.flags = 0;
    pr.exact_len = 0;                               // ... as are the needle's dwords for the exact verification
    pr.tail16[0] = pr.tail16[1] = pr.tail16[2] = pr.tail16[3] = 0;

    const ColdInRegisters cold = {&pr};
    if (n == 1) {
        scan_tiles<0, 0, true, U, 1, false, false, true>(pr, cold, s_needle, t0, 1, te, found);
        return;
    }
    switch (s / 4) {                                // single stream, non-temporal loads
    case 0: scan_tiles<0, 0, false, U, 1, false, false, true>(pr, cold, s_needle, t0, 1, te, found); break;
    case 1: scan_tiles<1, 0, false, U, 1, false, false, true>(pr, cold, s_needle, t0, 1, te, found); break;
    case 2: scan_tiles<2, 0, false, U, 1, false, false, true>(pr, cold, s_needle, t0, 1, te, found); break;
    default: scan_tiles<3, 0, false, U, 1, false, false, true>(pr, cold, s_needle, t0, 1, te, found); break;
    }
}

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

// One LANE per problem.  `nslices` = slices per problem of the scan launch that follows, `min_tiles` = the shortest slice worth a
// workgroup.  Same rules as scan_batched_kernel: needle[position] is always a first-phase byte; its partner is needle[0]
// when position < 16, else the rarest (class) byte of the 15 in front of it, closest to `position` among equals; the third
// byte is the rarest of the 15 behind the anchor, the later one among equals; the two are ordered by dword (q3 <= Q).
// Written for LATENCY - the scan cannot start before this kernel has ended: the rarity classes come from a 256-entry table
// in LDS (byte_rarity_rank is a dozen branches), and the needle bytes of a step are fetched by unconditional loads
// (out-of-range slots re-read byte 0 of the window) that are all in flight together; a first cut with a predicated
// load-rank loop ran 8-12 us, one memory round trip per byte.
__global__ void __launch_bounds__(kBlock) batch_plan_kernel(const BatchArgs a, uint64_t count, BatchDesc *descs,
                                                             uint32_t nslices, uint32_t min_tiles, int tile_pieces)
{
    __shared__ uint8_t s_class[256];
    {
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
    if (!live) return;
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
            uint32_t best_cls = 4;
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
        uint32_t p3 = s2, best_cls = 4;
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
    }
    if (a.best) a.best[prob] = n == 0 ? 0ull : ~0ull;      // the empty needle matches at offset 0 of every haystack
    else a.found[prob] = flag;
    descs[prob] = d;
}

// The cold fields of a planned problem, re-read from its descriptor by the waves that need them (scan_tiles' ColdT).
struct ColdFields {
    const uint8_t *hay, *needle;
    uint64_t n, end;
    uint32_t norder, exact_len;
    uint64_t order_idx[2], order_val[2];
    uint32_t tail16[4];
    int *host_flag;
    __device__ __forceinline__ const ColdFields *operator->() const { return this; }
};
struct ColdInDesc {
    const BatchDesc *dp;
    const uint8_t *needles;
    __device__ __forceinline__ ColdFields operator()() const
    {
        const BatchDesc *q = dp;
        __asm__ volatile("" : "+s"(q));             // opaque: the loads stay in the cold path
        ColdFields f;
        f.hay = q->base + (q->shifts & 15) - q->anchor;
        f.needle = needles + q->needle_off;
        f.n = q->n;
        f.end = q->end;
        f.norder = f.exact_len = 0;                 // LAZY_ORDER: built by the wave
        f.order_idx[0] = f.order_idx[1] = f.order_val[0] = f.order_val[1] = 0;
        f.tail16[0] = f.tail16[1] = f.tail16[2] = f.tail16[3] = 0;
        f.host_flag = nullptr;
        return f;
    }
};

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
constexpr uint32_t kPlanSliceMajorMax = 8;
// FIND: the sink is the problem's uint64 (leftmost offset, atomicMin); a workgroup skips only what lies right of the best so far
// (scan_tiles does that tile by tile, so the slice-major entry poll is not needed).
template <int U, bool FIND = false>
__global__ void __attribute__((amdgpu_waves_per_eu(4, 4))) __launch_bounds__(kBlock)
scan_batched_plan_kernel(const BatchArgs a, const BatchDesc *__restrict__ descs, uint32_t count, uint32_t nslices)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_needle[kWavesPerBlock * kNeedleLds];
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
    int *found = FIND ? nullptr : a.found + prob;
    void *sink = FIND ? static_cast<void *>(a.best + prob) : static_cast<void *>(found);
    const BatchDesc *dp = descs + prob;
    // slice-major, later slices: the problem's flag (one coherent load) is requested together with the descriptor (one scalar
    // load, s_load_dwordx16) - one round trip decides whether and what to scan
    const int seen = !FIND && slice_major && slice != 0 ? __hip_atomic_load(found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const BatchDesc d = *dp;
    const uint32_t mis = d.shifts & 15;
    const uint64_t npieces = ((mis + d.end + 15) / 16 + 63) / 64;
    const uint64_t ntiles = (npieces + kWavesPerBlock * U - 1) / (kWavesPerBlock * U);
    const uint32_t eff = (uint32_t)(d.per >> 32), per = (uint32_t)d.per;
    if (slice >= eff) return;                       // surplus slice, or a problem the plan kernel has answered (eff == 0)
    uint64_t t0, te, step;
    if (slice_major) {
        t0 = (uint64_t)slice * per;
        te = t0 + per < ntiles ? t0 + per : ntiles;
        step = 1;
        if (__builtin_amdgcn_readfirstlane(seen) != 0) return;   // later slices of a needle that has been found
    } else {
        t0 = slice;
        te = ntiles;
        step = eff;
    }
    if (t0 >= te) return;

    Problem pr;                                     // hot fields only; the cold ones are re-read from the descriptor
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
    const ColdInDesc cold = {dp, a.needles};
    if ((d.bytes >> 24) & 1) {
        scan_tiles<0, 0, true, U, 1, FIND, false, true>(pr, cold, s_needle, t0, step, te, sink);
        return;
    }
    switch ((d.shifts >> 6) & 3) {                  // single stream, non-temporal loads
    case 0: scan_tiles<0, 0, false, U, 1, FIND, false, true>(pr, cold, s_needle, t0, step, te, sink); break;
    case 1: scan_tiles<1, 0, false, U, 1, FIND, false, true>(pr, cold, s_needle, t0, step, te, sink); break;
    case 2: scan_tiles<2, 0, false, U, 1, FIND, false, true>(pr, cold, s_needle, t0, step, te, sink); break;
    default: scan_tiles<3, 0, false, U, 1, FIND, false, true>(pr, cold, s_needle, t0, step, te, sink); break;
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

// ---- synthetic haystack generator (not part of the reference) ------------------------
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// 8 generated bytes of global word index w, with 0xFF remapped to 0x00.  The seed is hashed first:
// with a raw `seed ^ w` two seeds that differ in a few low bits would produce the same stream with
// permuted words (seed 1 and seed 3: word w of one is word w^2 of the other).
__host__ __device__ __forceinline__ uint64_t synth_word(uint64_t seed, uint64_t w)
{
    const uint64_t v = splitmix64(splitmix64(seed) ^ w);
    // bytes equal to 0xFF: ~v has a zero byte there.  Exact per-byte zero detection (no borrow):
    const uint64_t x = ~v;
    const uint64_t zero = ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x | 0x7F7F7F7F7F7F7F7Full);
    const uint64_t ffmask = (zero >> 7) * 0xFFull;   // 0xFF in every byte of v that equals 0xFF
    return v & ~ffmask;
}

__global__ void __launch_bounds__(kBlock) fill_random_kernel(uint8_t *dst, uint64_t global_offset,
                                                             uint64_t len, uint64_t seed)
{
    // word-granular body on the GLOBAL index grid; bytes outside [0, len) are not written.
    const uint64_t first_word = global_offset >> 3;
    const uint64_t last_word = (global_offset + len + 7) >> 3;          // exclusive
    const uint64_t nwords = last_word - first_word;
    for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < nwords;
         k += (uint64_t)gridDim.x * kBlock) {
        const uint64_t w = first_word + k;
        const uint64_t v = synth_word(seed, w);
        const int64_t o = (int64_t)(w << 3) - (int64_t)global_offset;    // dst offset of byte 0 of the word
        if (o >= 0 && (uint64_t)o + 8 <= len && (((uintptr_t)(dst + o)) & 7) == 0) {
            *reinterpret_cast<uint64_t *>(dst + o) = v;
        } else {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int64_t ob = o + b;
                if (ob >= 0 && (uint64_t)ob < len) dst[ob] = (uint8_t)(v >> (8 * b));
            }
        }
    }
}

// ---- plain streaming read: the empirical "achievable HBM read" reference ------------------------------
// Same access shape as the scan (workgroup-contiguous tiles of 4*U KiB per 4 waves, short-lived workgroups).
// V = u32x4 (16 bytes per lane, 1 KiB per wave instruction) or u32x2 (8 bytes per lane, the L8 shape).
template <int U, typename V>
__global__ void __launch_bounds__(kMaxBlock) read_ceiling_kernel(const V *src, uint64_t nvec, uint32_t *sink,
                                                                 uint64_t tiles_per_block)
{
    constexpr int kPerKiB = 1024 / (64 * (int)sizeof(V));              // wave instructions per KiB piece: 1 or 2
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint64_t wpb = blockDim.x / kWave;
    const uint64_t ntiles = nvec >> (__builtin_ctz(64 * kPerKiB * U) + __builtin_ctz((unsigned)wpb));   // ragged tail ignored
    uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_block;
    const uint64_t t1 = t0 + tiles_per_block < ntiles ? t0 + tiles_per_block : ntiles;
    V acc = {};
    for (; t0 < t1; ++t0) {
        const V *p = src + (t0 * (wpb * U) + (uint64_t)wave * U) * (64 * kPerKiB) + lane;
        V v[U * kPerKiB];
#pragma unroll
        for (int u = 0; u < U * kPerKiB; ++u) v[u] = __builtin_nontemporal_load(p + 64 * u);
#pragma unroll
        for (int u = 0; u < U * kPerKiB; ++u) acc ^= v[u];
    }
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < (int)(sizeof(V) / 4); ++k) r ^= acc[k];
    if (r == 0x9E3779B9u) sink[0] = r;      // practically never; keeps the loads alive
}

// ---- byte histogram (row f3 of SURVEY.md 8f: data for a rare-byte `position` policy) ----------------
// Per-wave private LDS histograms (4 x 256 counters per workgroup), flushed with one global atomic per
// non-zero counter.  `stride_chunks` > 1 samples every stride-th 16-byte chunk.
__global__ void __launch_bounds__(kBlock) byte_histogram_kernel(const uint8_t *hay, uint64_t len, uint64_t stride_chunks,
                                                                unsigned long long *hist)
{
    __shared__ uint32_t h[kWavesPerBlock][256];
    for (int k = threadIdx.x; k < kWavesPerBlock * 256; k += kBlock) (&h[0][0])[k] = 0;
    __syncthreads();
    const int wave = threadIdx.x / kWave;
    const uint64_t nchunks = len / 16;             // the ragged tail (< 16 bytes) is ignored: this is a sample
    const bool aligned = (((uintptr_t)hay) & 15) == 0;
    for (uint64_t c = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * stride_chunks; c < nchunks;
         c += (uint64_t)gridDim.x * kBlock * stride_chunks) {
        uint32_t w[4];
        if (aligned) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(hay + c * 16);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = reinterpret_cast<const UnalignedU32 *>(hay + c * 16 + 4 * j)->v;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(&h[wave][w[j] & 0xFF], 1u);
            atomicAdd(&h[wave][(w[j] >> 8) & 0xFF], 1u);
            atomicAdd(&h[wave][(w[j] >> 16) & 0xFF], 1u);
            atomicAdd(&h[wave][w[j] >> 24], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 256; k += kBlock) {
        const unsigned long long t = (unsigned long long)h[0][k] + h[1][k] + h[2][k] + h[3][k];
        if (t) atomicAdd(&hist[k], t);
    }
}

// Stream-ordered behind a scan (and the all-reduce of a sharded search): the answer word for a host that spins on pinned memory
// instead of waiting for the stream's completion signal (some 30 us quicker on this stack).  pair == 0: epoch << 1 | found.
// pair != 0 (behind the all-reduce(MAX) of a sharded search, whose flag is a PAIR - {found, a rank failed its local part}, both
// epoch-valued): epoch << 2 | failed << 1 | found.
__global__ void signal_flag_kernel(const int *d_flag, int epoch, long long *h_word, int pair)
{
    const unsigned long long f = __hip_atomic_load(d_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
    unsigned long long w = ((unsigned long long)(uint32_t)epoch << 1) | f;
    if (pair) {
        const unsigned long long e = __hip_atomic_load(d_flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
        w = ((unsigned long long)(uint32_t)epoch << 2) | (e << 1) | f;
    }
    __hip_atomic_store(h_word, (long long)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// find(): hands the final minimum to the host through its pinned mirror (one system-scope store), stream-ordered behind the scan
// - the read-back of ss_find_device without a device-to-host copy command.
// pair == 0 (ss_find_device): the slot is re-armed (all ones) for its next user - re-arm FIRST, publish second: the host releases
//   the slot the moment the pinned word changes, and the next find() on the slot may run on another stream; its atomicMin must
//   never meet a re-arm store that is still in flight (the release ordering of the system-scope store waits for the re-arm).
// pair != 0 (behind the all-reduce(MIN) of a sharded find): {leftmost offset, all ones unless a rank failed}; nothing to re-arm
//   (the pair is the communicator's scratch); the status word first, the offset - the word the host spins on - behind it.
__global__ void publish_best_kernel(uint64_t *d_best, uint64_t *h_best, int pair)
{
    const uint64_t v = __hip_atomic_load(d_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (pair) {
        const uint64_t ok = __hip_atomic_load(d_best + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(h_best + 1, ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (v != ~0ull) {
        __hip_atomic_store(d_best, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __hip_atomic_store(h_best, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- resident search service (ss_service_*) ---------------------------------------------------------------
// The launch path costs a search 8-10 us whatever its size: doorbell, command processor, dispatch, completion.  A host <->
// device round trip against a kernel that is ALREADY running costs 1.5-2.5 us (mailbox_echo_kernel below, tools/
// vram_mailbox_probe.hip).  The service is that kernel: `gridDim.x` workgroups that stay on the device and take one request
// at a time:
//   * the mailbox is 256 bytes of DEVICE memory that the host writes through the PCIe BAR (every MI300-class part exposes
//     all of its memory to the CPU): four 64-byte lines, each 15 payload dwords + the request's sequence number as its LAST
//     dword.  The host writes the payload with ZERO in the number's place, fences, then the four numbers: posted writes arrive in
//     order, so a line that shows a number holds that request's payload, whichever request a wave is waiting for;
//   * EVERY wave of every workgroup polls the mailbox itself - 64 lanes x 4 bytes, one instruction, served by the device's own
//     memory - and takes the request straight out of the polled registers: no leader, no hop between workgroups.  (Round 3
//     began with the mailbox in pinned HOST memory: every poll crossed PCIe, so only one wave could poll and had to hand
//     the request on through device memory - 64 pollers made a round trip 13 us, tools/vram_mailbox_probe.hip; with the
//     mailbox on the device's side of the link 64 workgroups answer in 3.4 us.)
//   * every workgroup scans tiles b, b + grid, ... of the haystack with the same scan_tiles<> as every other kernel, counts
//     itself out exactly like a completion-word launch of scan_kernel, and the workgroup that completes the count stores
//     found-count << 32 | sequence << 1 | found to the pinned answer word the host spins on.
// Measured: profiles/r03/service_experiments.md.
// Residency is a LEASE: without a request for `idle_ticks` (100 MHz s_memrealtime) the keeper (wave 0 of workgroup 0)
// announces that it is leaving, looks at the mailbox once more (a request posted meanwhile is served; host and device each
// write their word before reading the other's), sets the stop word the others poll beside the mailbox, and the kernel ends;
// the host starts it again with its next request.  A request that arrives while the stop word spreads may be taken by some
// waves and not by others: its count never completes, the host sees the kernel gone, waits for the stream, resets the
// counter and posts the request again to a new residency.  Nothing that waits for the whole device - hipDeviceSynchronize,
// hipFree - can wait longer than the lease.  Every spin in here is bounded.
struct ServiceRequest {
    Problem pr;
    uint32_t q;            // dword window of the second filter byte (the kernels' template parameter Q)
    uint32_t one_byte;
    uint32_t stop;         // != 0: no search - the service ends
    uint32_t settled;      // != 0: every byte this request reads was last written before an earlier request's acquire (or the
                           // kernel's start) - a bound haystack (ss_service_bind), a needle uploaded earlier: no acquire
    uint32_t active;       // workgroups 0 .. active-1 scan (tiles b, b + active, ...) and count out; the others only watch
    uint32_t pad_;
};
static_assert(sizeof(ServiceRequest) <= 240 && sizeof(ServiceRequest) % 8 == 0, "four mailbox lines of 60 payload bytes");
constexpr uint32_t kSvcRunning = 1, kSvcLeaving = 2, kSvcExited = 3;
constexpr unsigned long long kSvcStopSeq = ~0ull;

#ifndef SS_SERVICE_NT
#define SS_SERVICE_NT 0
#endif
template <int U>
__global__ void __launch_bounds__(kBlock)
service_kernel(const uint32_t *d_req, uint32_t *h_status, unsigned long long *h_answer, uint32_t *d_stop, unsigned long long *d_done,
               int *d_found, uint32_t first_seq, unsigned long long idle_ticks)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_needle[kWavesPerBlock * kNeedleLds];
    __shared__ int s_wg_found;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const bool keeper = blockIdx.x == 0 && wave == 0;                      // the wave that watches the lease
    constexpr unsigned long long kWorkerPatience = 300000000ull;           // 3 s of s_memrealtime: no wave ever waits longer
    constexpr int kDwords = (int)(sizeof(ServiceRequest) / 4);
    constexpr int kStopPayloadDword = (int)(offsetof(ServiceRequest, stop) / 4);
    constexpr int kStopLane = kStopPayloadDword + kStopPayloadDword / 15;
    if (keeper && lane == 0) __hip_atomic_store(h_status, kSvcRunning, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (uint32_t next = first_seq;; ++next) {
        // ---- 1. EVERY wave polls the mailbox: device memory the host writes through the BAR ----------------------------------
        // lane i <- dword i of the mailbox (one instruction, four lines); agent-scope loads are performed beyond the L2, where
        // the host's stores arrive.  A line that shows `next` in its last dword holds this request's payload (the host writes
        // the payload, fences, THEN the four sequence dwords).
        uint32_t v = 0;
        bool leave = false;
        {
            auto issue = [&]() { return __hip_atomic_load(d_req + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            // A request is there when all four lines show the SAME number m >= next (never 0: that is what the lines show while the
            // host writes a payload).  m > next: this wave never saw the requests in between - possible only for requests its
            // workgroup took no part in (a request completes when every ACTIVE workgroup has counted out; the others may lag),
            // so they are skipped.
            auto shows_next = [&](uint32_t x) {
                const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)x, 15);
                if (m == 0 || m < next || (uint32_t)__builtin_amdgcn_readlane((int)x, 31) != m ||
                    (uint32_t)__builtin_amdgcn_readlane((int)x, 47) != m || (uint32_t)__builtin_amdgcn_readlane((int)x, 63) != m)
                    return false;
                next = m;
                return true;
            };
            // TWO polls in flight, issued half a memory latency apart and each re-issued as it returns: the mailbox is sampled
            // twice per latency instead of once, a request waits a quarter of a latency less to be seen.  The stop word and the
            // lease are looked at every 32nd round only.
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            uint32_t pa = issue();
            __builtin_amdgcn_s_sleep(10);
            uint32_t pb = issue();
            for (unsigned round = 1;; ++round) {
                if (shows_next(pa)) { v = pa; break; }
                pa = issue();
                if (shows_next(pb)) { v = pb; break; }
                pb = issue();
                if ((round & 31) != 0) continue;
                const uint32_t stopw = (uint32_t)__builtin_amdgcn_readfirstlane(
                    (int)__hip_atomic_load(d_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (stopw != 0) { leave = true; break; }
                const unsigned long long waited = __builtin_amdgcn_s_memrealtime() - t0;
                if (keeper && waited > idle_ticks) {
                    // the lease is over: say so, THEN look once more (the host posts its request, THEN reads this word)
                    if (lane == 0) __hip_atomic_store(h_status, kSvcLeaving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
                    v = issue();
                    if (shows_next(v)) {
                        if (lane == 0) __hip_atomic_store(h_status, kSvcRunning, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    if (lane == 0) __hip_atomic_store(d_stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    leave = true;
                    break;
                }
                if (!keeper && waited > idle_ticks + kWorkerPatience) { leave = true; break; }   // the keeper is gone: leave, do not hang
            }
        }
        if (leave || __builtin_amdgcn_readlane((int)v, kStopLane) != 0) break;
        // ---- 2. the request, out of the polled registers into scalar registers ---------------------------------------------
        union {
            ServiceRequest rq;
            uint32_t w[kDwords];
        } u;
#pragma unroll
        for (int k = 0; k < kDwords; ++k) u.w[k] = (uint32_t)__builtin_amdgcn_readlane((int)v, k + k / 15);
        const ServiceRequest &rq = u.rq;
        // A kernel that never ends sees no kernel boundary: haystack or needle bytes written since it last looked (a copy, another
        // kernel) may still sit in this XCD's L2 / this CU's vector cache in their old state.  The acquire drops them - 2 us of
        // a request - unless the host vouches that nothing this request reads has changed (ServiceRequest::settled).
        // A small request is not worth every workgroup's count: the host names how many take part (one per tile at most).
        if (blockIdx.x >= rq.active) continue;
        if (!rq.settled) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // ---- 3. scan: workgroup b takes tiles b, b + active, ... -----------------------------------------------------------
        if (threadIdx.x == 0) __hip_atomic_store(&s_wg_found, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        const uint64_t ntiles = (rq.pr.npieces + kWavesPerBlock * U - 1) / (kWavesPerBlock * U);
        const ColdInRegisters cold = {&rq.pr};
        if (rq.one_byte) {
            scan_tiles<0, 0, true, U, SS_SERVICE_NT, false, false, false>(rq.pr, cold, s_needle, blockIdx.x, rq.active, ntiles, d_found, &s_wg_found);
        } else {
            switch (rq.q) {
            case 0: scan_tiles<0, 0, false, U, SS_SERVICE_NT, false, false, false>(rq.pr, cold, s_needle, blockIdx.x, rq.active, ntiles, d_found, &s_wg_found); break;
            case 1: scan_tiles<1, 0, false, U, SS_SERVICE_NT, false, false, false>(rq.pr, cold, s_needle, blockIdx.x, rq.active, ntiles, d_found, &s_wg_found); break;
            case 2: scan_tiles<2, 0, false, U, SS_SERVICE_NT, false, false, false>(rq.pr, cold, s_needle, blockIdx.x, rq.active, ntiles, d_found, &s_wg_found); break;
            default: scan_tiles<3, 0, false, U, SS_SERVICE_NT, false, false, false>(rq.pr, cold, s_needle, blockIdx.x, rq.active, ntiles, d_found, &s_wg_found); break;
            }
        }
        // ---- 4. count out; the workgroup that completes the count answers (scan_kernel's completion word) ---------------
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long f = (unsigned long long)__hip_atomic_load(&s_wg_found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (rq.active == 1) {
                // the only workgroup of this request: nobody to count with - the answer goes out a memory round trip earlier
                // (the counter and the host's copy of it stay as they are)
                __hip_atomic_store(h_answer, ((unsigned long long)rq.pr.done_hi << 32) | ((unsigned long long)next << 1) | f,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                const unsigned long long one = 1ull + (f << 32);
                const unsigned long long total = __hip_atomic_fetch_add(d_done, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + one;
                if ((uint32_t)total == rq.pr.done_target) {
                    const uint32_t hi = (uint32_t)(total >> 32);
                    __hip_atomic_store(h_answer, ((unsigned long long)hi << 32) | ((unsigned long long)next << 1) | (hi != rq.pr.done_hi ? 1ull : 0ull),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        __syncthreads();
    }
    if (keeper && lane == 0) __hip_atomic_store(h_status, kSvcExited, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Mailbox round trip (ss_mailbox_round_trip_us): ONE lane answers `iters` requests posted by the host to pinned memory -
// what a resident "search service" would pay per request before it has looked at a single haystack byte.  Every wait is
// bounded (s_memtime ticks), so the kernel ends by itself whatever the host does.
__global__ void mailbox_echo_kernel(const unsigned long long *req, unsigned long long *resp, unsigned iters, unsigned long long max_ticks)
{
    for (unsigned i = 1; i <= iters; ++i) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < i)
            if (__builtin_readcyclecounter() - t0 > max_ticks) return;
        __hip_atomic_store(resp, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Cross-lane self-test: the DPP controls and v_alignbyte the scan relies on, next to __shfl statements.
__global__ void dpp_probe_kernel(uint32_t *out)
{
    const uint32_t v = 1000u + threadIdx.x;
    out[threadIdx.x] = from_next_lane_or(0u, v);
    const uint32_t viaShfl = (uint32_t)__shfl_down((int)v, 1);
    out[64 + threadIdx.x] = (threadIdx.x == 63) ? 0u : viaShfl;
    out[128 + threadIdx.x] = __builtin_amdgcn_alignbyte(0x44332211u, 0xDDCCBBAAu, threadIdx.x & 3);
    out[192 + threadIdx.x] = rotate_from_next_lane(v);
    out[256 + threadIdx.x] = from_next_lane_or(7777u, v);
}

#endif  // SS_MISC_KERNELS

}  // namespace ss
