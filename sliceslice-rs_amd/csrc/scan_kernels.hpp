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
//   * d > 0 (a pair 16 or more apart, ss_searcher_set_filter3 only), two filter bytes: MODE 2 keeps ONE non-temporal load stream and
//     fetches the position-byte differences from the lane that owns chunk c+d with ds_bpermute (d <= 62).  A pair farther apart
//     than that has no kernel of its own (rounds 1-3 had one with a second load stream: 0.83 of the roofline, 6 % re-read traffic):
//     the host filters with the first byte and two partners close behind it (MODE 0) and the caller's far byte is the first thing
//     a surviving candidate is tested for in memory (Problem::far_off, verify_flags).
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
#include "scan_filters.hpp"

namespace ss {

#ifdef SS_CAND_PROF   // instrumented A/B builds only (tools/cand_prof.py): what a wave spends in the candidate path, in s_memtime ticks
constexpr int kCandProfSlots = 4096;                        // one 64-byte row per (workgroup mod 4096): no two neighbours on one line
static __device__ unsigned long long g_cand_prof[kCandProfSlots][8];   // tiles with candidates | ticks: total | cold part | second level | compare | - | per-piece tiles
#define SS_PROF_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define SS_PROF_ADD(k, d) do { if (lane == 0) atomicAdd(&g_cand_prof[blockIdx.x % kCandProfSlots][k], (unsigned long long)(d)); } while (0)
#else
#define SS_PROF_T(v)
#define SS_PROF_ADD(k, d)
#endif

// MODE selects where the position-byte flags of a candidate come from (position = 16*d + 4*Q + r):
//   0  d == 0: same chunk / next lane (DPP) - every needle of <= 16 bytes with the default position;
//   2  0 < d < 64, small: ONE (non-temporal) load stream; the flags computed by the lane that owns chunk
//      c+d are fetched across lanes with ds_bpermute (the wave loads d+1 halo chunks after its last piece).
// FIND = false: `sink` is the int found flag (0 -> 1).  FIND = true: `sink` is the uint64 leftmost-match
// offset (row f1 of SURVEY.md 8f: the `Option<usize>` shape of tests/i386.rs:6-10 and
// bench/sse4-strstr/src/lib.rs:4-15); a wave only skips work that lies to the RIGHT of the best so far.
// LAZY_ORDER (batched kernel): the descriptor arrives without the second-level schedule; a wave builds it
// when it first meets a candidate, next to staging the needle - not on every workgroup's way in.
// `pr` is read for its HOT fields only; `cold()` yields a pointer through which the cold ones are read where they are needed.
// The kernels that serve many problems per grid have no workgroup that peeks at a flag through the scalar cache (their entry polls
// are coherent loads), so a wave that leaves early has nothing to make visible there - and an invalidation costs the workgroups
// that start on that CU next their descriptor and kernel-argument lines: plan runs with matches 2-5 % (the i386 loop 0.122 ms
// instead of 0.128), the unplanned calls nothing (profiles/r05/ab_batch_no_dcache_inv.jsonl).
template <bool BATCHED> constexpr bool kNoForget = BATCHED;
// ColdT::kSingleLaunchPlan (batched_kernels.hpp, ColdInPlan): the problem's state word is the plan's own, the CALLER'S output is
// written behind it by the finding wave (state first, output second - see scan_batched_plan_kernel), and the first finder of a
// problem counts it into the plan's tally.
template <class T, class = void> struct SingleLaunchPlan : std::false_type {};
template <class T> struct SingleLaunchPlan<T, std::void_t<decltype(T::kSingleLaunchPlan)>> : std::bool_constant<T::kSingleLaunchPlan> {};
template <int Q, int MODE, bool ONE_BYTE, int U, int NTMODE, bool FIND = false, bool L8 = false, bool LAZY_ORDER = false,
          typename ColdT = ColdInRegisters>
__device__ __forceinline__ void scan_tiles(const Problem &pr, ColdT cold, uint8_t *s_needle_block, uint64_t tile0,
                                           uint64_t tile_step, uint64_t tile_end, void *sink, void *wg_sink = nullptr)
{
    static_assert(!L8 || (MODE == 0 && !FIND), "the 8-byte layout covers the single-stream bool kernels");
    static_assert(Q != kQDynamic || (MODE == 0 && !L8), "a run-time window is for the single-stream kernels' three-byte phase");
    static_assert(MODE == 0 || MODE == 2 || MODE == 3, "single-stream kernels only");
    constexpr bool SHIFTED = MODE >= 2;
    constexpr bool SHIFTED3 = MODE == 2;            // ... with a third byte close behind the first (MODE 3: the pair alone)
    int *found = static_cast<int *>(sink);
    uint64_t *best = static_cast<uint64_t *>(sink);
    // `wg_sink`, when given: a word in the workgroup's LDS that takes the match INSTEAD of the global sink (bool: int 0 -> 1;
    // FIND, batched kernels only: uint64 minimum) - for launches whose epilogue carries the answer on, so that no wave queues a
    // device-scope atomic in front of it.
    int *wg_found = static_cast<int *>(wg_sink);
    constexpr bool WG_FIND = LAZY_ORDER;            // (compiled into scan_kernel's FIND path it would be dead code that still moves registers)
    // LAZY_ORDER kernels fetch the cold part once per wave; whether the wave then BUILDS the schedule and the needle's dwords or
    // finds them there depends on where the cold part comes from (a plan's descriptors carry them: batched_kernels.hpp, BatchCold)
    constexpr bool BUILD_ORDER = LAZY_ORDER && !ColdT::kHasOrder;
    // ... or MAY find them there: the unplanned batched kernel, whose plan kernel leaves the needle's dwords where they cost it
    // nothing (needles of up to 16 bytes with the first filter byte at index 0) and says so in the record
    constexpr bool MAYBE_ORDER = BUILD_ORDER && ColdT::kMaybeOrder;
    constexpr bool NTA = NTMODE >= 1;
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
        u32x4 A[U], H = {0, 0, 0, 0};
        const uint64_t chunk0 = (tile * (uint64_t)(wpb * U) + (uint64_t)wave * U) * 64;   // wave-uniform
        // FIND polls first (oldest load, so waiting for it does not drain the data loads behind it); the value
        // is only made wave-uniform (readfirstlane = the wait) after the tile's data loads have been issued
        const uint64_t best_raw = FIND ? __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        uint64_t best_now = 0;
        // last chunk this wave touches: the halo chunk (MODE 0) or the d+1 halo chunks (MODE 2)
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
            // all three filter bytes within four bytes (Q == 0 and q3 == 0): the three-byte filter on the 8-byte layout
            if (THREE && Q == 0 && pr.q3 == 0) {                  // (wave-uniform)
#pragma unroll
                for (int h = 0; h < 2 * U; ++h) {
                    const uint32_t nx0 = h + 1 < 2 * U ? Hh[h + 1].x : halo8.x;
                    any8 = filter_half3_near(Hh[h], from_next_lane_or(rotate_from_next_lane(nx0), Hh[h].x), pr, any8);
                }
            } else {
                u32x2 tc = {Hh[0].x ^ pr.nlx4, Hh[0].y ^ pr.nlx4}, tn = {0, 0};
#pragma unroll
                for (int h = 0; h < 2 * U; ++h) {
                    if (!ONE_BYTE) {
                        const u32x2 nx = h + 1 < 2 * U ? Hh[h + 1] : halo8;
                        tn = u32x2{nx.x ^ pr.nlx4, nx.y ^ pr.nlx4};
                    }
                    any8 = filter_half<(Q < 0 ? 0 : Q), ONE_BYTE>(Hh[h], tc, tn, pr, lane, any8);   // (L8 kernels have a compile-time window)
                    tc = tn;
                }
            }
            if (stop8) {                                      // somebody has already found the needle
                forget_scalar_cache_unless(small_grid || kNoForget<LAZY_ORDER>);
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
        auto load_and_filter = [&](auto loaded_c, auto full_c, auto q3_c, auto q_c) {
            constexpr bool LOADED = decltype(loaded_c)::value, FULL = decltype(full_c)::value;
            constexpr int Q3 = decltype(q3_c)::value;
            constexpr int QQ = Q == kQDynamic ? decltype(q_c)::value : Q;       // the second byte's dword window
            if constexpr (!LOADED && FULL) {
#pragma unroll
                for (int u = 0; u < U; ++u) A[u] = load_chunk<NTA>(pr.base, chunk0 + 64 * u + lane);
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
            if (!ONE_BYTE && !THREE) position_diffs(A[0], pr.nlx4, wcur);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t *g = G[u];
                if (THREE) {
                    // lane 63's next lane: lane 0 of the next piece (raw dwords, rotated in), or the halo chunk
                    constexpr int QM = QQ > Q3 ? QQ : Q3;
                    uint32_t nx[4] = {0, 0, 0, 0};
                    if (u + 1 < U) {
                        nx[0] = rotate_from_next_lane(A[u + 1].x);
                        if (QM >= 1) nx[1] = rotate_from_next_lane(A[u + 1].y);
                        if (QM >= 2) nx[2] = rotate_from_next_lane(A[u + 1].z);
                        if (QM >= 3) nx[3] = rotate_from_next_lane(A[u + 1].w);
                    } else {
                        nx[0] = H.x; nx[1] = H.y; nx[2] = H.z; nx[3] = H.w;
                    }
                    filter_piece3<QQ, Q3>(A[u], nx, pr, g);
                } else if (SHIFTED) {
                    // flags of the following piece (or of the halo chunks), then the 8-dword window by lane distance
                    position_diffs(u + 1 < U ? A[u + 1] : H, pr.nlx4, wnext);
                    constexpr int QS = Q < 0 ? 0 : Q;        // (a run-time window never comes here: kQDynamic is MODE 0)
                    uint32_t x[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        x[j] = (j >= QS) ? from_lane_ahead(wcur[j], wnext[j], lane, d) : 0u;
                        x[4 + j] = (j <= QS) ? from_lane_ahead(wcur[j], wnext[j], lane, d + 1) : 0u;
                    }
                    // ... and a THIRD byte close behind the first one (within 15 bytes: this lane's chunk and the next lane's, one DPP
                    // hop like the single-stream kernels' third byte; lane 63 takes lane 0 of the next piece / of the halo chunks).
                    // A pair this far apart is the caller's (ss_searcher_set_filter3) - on text the reference's own pair (0, n-1)
                    // passes at percent rates and sent nearly every tile into the second level (0.79 of the roofline); a third
                    // byte in the first phase makes a candidate tile the exception again.  Whoever builds the Problem provides it.
                    if constexpr (SHIFTED3) {
                        const u32x4 &NP = u + 1 < U ? A[u + 1] : H;
                        uint32_t xx[8] = {A[u].x, A[u].y, A[u].z, A[u].w, 0, 0, 0, 0};
                        xx[4] = from_next_lane_or(rotate_from_next_lane(NP.x), A[u].x);
                        if (Q3 >= 1) xx[5] = from_next_lane_or(rotate_from_next_lane(NP.y), A[u].y);
                        if (Q3 >= 2) xx[6] = from_next_lane_or(rotate_from_next_lane(NP.z), A[u].z);
                        if (Q3 >= 3) xx[7] = from_next_lane_or(rotate_from_next_lane(NP.w), A[u].w);
                        uint32_t z[5];
#pragma unroll
                        for (int k = 0; k < 5; ++k) z[k] = xx[Q3 + k] ^ pr.n3x4;
                        g[0] = zero_byte_flags((A[u].x ^ pr.n0x4) | __builtin_amdgcn_alignbyte(x[QS + 1], x[QS + 0], pr.r) | __builtin_amdgcn_alignbyte(z[1], z[0], pr.r3));
                        g[1] = zero_byte_flags((A[u].y ^ pr.n0x4) | __builtin_amdgcn_alignbyte(x[QS + 2], x[QS + 1], pr.r) | __builtin_amdgcn_alignbyte(z[2], z[1], pr.r3));
                        g[2] = zero_byte_flags((A[u].z ^ pr.n0x4) | __builtin_amdgcn_alignbyte(x[QS + 3], x[QS + 2], pr.r) | __builtin_amdgcn_alignbyte(z[3], z[2], pr.r3));
                        g[3] = zero_byte_flags((A[u].w ^ pr.n0x4) | __builtin_amdgcn_alignbyte(x[QS + 4], x[QS + 3], pr.r) | __builtin_amdgcn_alignbyte(z[4], z[3], pr.r3));
                    } else {
                        // MODE 3: the pair alone - for haystacks on which it rarely matches (the census says so: ss_scan.hip), where
                        // the third byte's 6.25 LDS operations and dozen VALU per KiB thin out nothing
                        g[0] = zero_byte_flags((A[u].x ^ pr.n0x4) | __builtin_amdgcn_alignbyte(x[QS + 1], x[QS + 0], pr.r));
                        g[1] = zero_byte_flags((A[u].y ^ pr.n0x4) | __builtin_amdgcn_alignbyte(x[QS + 2], x[QS + 1], pr.r));
                        g[2] = zero_byte_flags((A[u].z ^ pr.n0x4) | __builtin_amdgcn_alignbyte(x[QS + 3], x[QS + 2], pr.r));
                        g[3] = zero_byte_flags((A[u].w ^ pr.n0x4) | __builtin_amdgcn_alignbyte(x[QS + 4], x[QS + 3], pr.r));
                    }
                } else {
                    if (!ONE_BYTE) {
                        // lane 63's "next lane": lane 0 of the next piece, or the halo chunk after the last piece
                        if (u + 1 < U) {
                            position_diffs(A[u + 1], pr.nlx4, wnext);
#pragma unroll
                            for (int j = 0; j < 4; ++j) wlast[j] = (j <= QQ) ? rotate_from_next_lane(wnext[j]) : 0u;
                        } else {
                            position_diffs(H, pr.nlx4, wlast);
                        }
                    }
                    filter_piece<QQ < 0 ? 0 : QQ, ONE_BYTE>(A[u], wcur, wlast, pr, g);
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
            using std::integral_constant;
            if constexpr (THREE && Q == kQDynamic) {
                // Kernels that serve many problems per grid (batched, service) cannot take the second byte's window from their
                // template arguments: ONE copy of everything else and one copy of the first phase per (Q, Q3) window pair - ten,
                // because whoever builds the Problem orders the two further bytes so that q3 <= q - instead of one whole
                // scan_tiles per Q (five per kernel, each with its own spill slots: 184-245 spilled scalar registers).
                switch (pr.q * 4 + pr.q3) {
                case 0: load_and_filter(loaded_c, full_c, integral_constant<int, 0>{}, integral_constant<int, 0>{}); break;
                case 4: load_and_filter(loaded_c, full_c, integral_constant<int, 0>{}, integral_constant<int, 1>{}); break;
                case 5: load_and_filter(loaded_c, full_c, integral_constant<int, 1>{}, integral_constant<int, 1>{}); break;
                case 8: load_and_filter(loaded_c, full_c, integral_constant<int, 0>{}, integral_constant<int, 2>{}); break;
                case 9: load_and_filter(loaded_c, full_c, integral_constant<int, 1>{}, integral_constant<int, 2>{}); break;
                case 10: load_and_filter(loaded_c, full_c, integral_constant<int, 2>{}, integral_constant<int, 2>{}); break;
                case 12: load_and_filter(loaded_c, full_c, integral_constant<int, 0>{}, integral_constant<int, 3>{}); break;
                case 13: load_and_filter(loaded_c, full_c, integral_constant<int, 1>{}, integral_constant<int, 3>{}); break;
                case 14: load_and_filter(loaded_c, full_c, integral_constant<int, 2>{}, integral_constant<int, 3>{}); break;
                default: load_and_filter(loaded_c, full_c, integral_constant<int, 3>{}, integral_constant<int, 3>{}); break;
                }
            } else if constexpr (THREE || SHIFTED3) {
                // (SHIFTED: the third byte's window is independent of the far second byte's)
                // Whoever builds the Problem orders the two further bytes so that q3 <= Q (they are interchangeable): the
                // copies with Q3 > Q are never taken.  They stay instantiated all the same: with them pruned the register
                // allocator needed 146 VGPRs instead of 121 for Q < 3 (build() records every kernel's registers in csrc/kernel_resources.json; tests/test_bindings_cpu.py keeps an eye on it).
                switch (pr.q3) {
                case 0: load_and_filter(loaded_c, full_c, integral_constant<int, 0>{}, integral_constant<int, 0>{}); break;
                case 1: load_and_filter(loaded_c, full_c, integral_constant<int, 1>{}, integral_constant<int, 0>{}); break;
                case 2: load_and_filter(loaded_c, full_c, integral_constant<int, 2>{}, integral_constant<int, 0>{}); break;
                default: load_and_filter(loaded_c, full_c, integral_constant<int, 3>{}, integral_constant<int, 0>{}); break;
                }
            } else {
                load_and_filter(loaded_c, full_c, integral_constant<int, 0>{}, integral_constant<int, 0>{});
            }
        };
        if (have16) run_phase1(std::true_type{}, std::true_type{});
        else if (full) run_phase1(std::false_type{}, std::true_type{});
        else run_phase1(std::false_type{}, std::false_type{});
        if (FIND) {
            const uint64_t first = chunk0 * 16 > pr.mis ? chunk0 * 16 - pr.mis : 0;   // lowest index this wave can report
            if (best_now <= pr.find_base + first) {                                     // all of it lies right of a match
                forget_scalar_cache_unless(small_grid || kNoForget<LAZY_ORDER>);
                return;
            }
        }

        // ---- phase 2 (rare on random bytes): the wave's "movemask != 0" ---------------------------------
        const bool cand_tile = __ballot((any_tile & 0x80808080u) != 0) != 0;
        if (L8) dense = cand_tile;      // stay in the 16-byte layout while tiles keep producing candidates
        if (stop) {                     // somebody has already found the needle: no point in verifying more
            forget_scalar_cache_unless(small_grid || kNoForget<LAZY_ORDER>);
            return;
        }
        if (cand_tile) {
            SS_PROF_T(prof_t0);
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
                bool take = !BUILD_ORDER;
                if constexpr (MAYBE_ORDER) take = __builtin_amdgcn_readfirstlane((int)c->ready) != 0;
                if (!ONE_BYTE && (MAYBE_ORDER || !BUILD_ORDER) && take) {
                    ro.n = c->norder;
                    ro.idx[0] = c->order_idx[0]; ro.idx[1] = c->order_idx[1];
                    ro.val[0] = c->order_val[0]; ro.val[1] = c->order_val[1];
                    if (EXACT_OK) {
                        exact_len = c->exact_len;
#pragma unroll
                        for (int j = 0; j < 4; ++j) tail16[j] = c->tail16[j];
                    }
                    if (LAZY_ORDER) {
                        // (a plan's cold part arrives through memory behind the kernel's first stores, i.e. by vector loads: what
                        // steers the second level must sit in scalar registers all the same)
                        ro.n = (uint32_t)__builtin_amdgcn_readfirstlane((int)ro.n);
                        for (int t = 0; t < 2; ++t) {
                            ro.idx[t] = uniform64(ro.idx[t]);
                            ro.val[t] = uniform64(ro.val[t]);
                        }
                        if (EXACT_OK) {
                            exact_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)exact_len);
#pragma unroll
                            for (int j = 0; j < 4; ++j) tail16[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)tail16[j]);
                        }
                    }
                }
                if (!ONE_BYTE && BUILD_ORDER && !take) {
                    // the descriptor came without the schedule (and without the needle's dwords): built here, by the waves
                    // that need them, not on every workgroup's way in
                    const uint64_t position = pr.d * 16 + 4 * (Q == kQDynamic ? pr.q : (uint32_t)Q) + pr.r;
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
            SS_PROF_T(prof_t1);
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
            constexpr bool TILE_WIDE = MODE < 2 || SS_MODE2_TILE_WIDE != 0;
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
                    if (max_steps != 0 && !refine_tile<U, MODE>(A, H, ro, G, max_steps)) {
                        SS_PROF_T(prof_tx);
                        SS_PROF_ADD(0, 1);
                        SS_PROF_ADD(1, prof_tx - prof_t0);
                        SS_PROF_ADD(2, prof_t1 - prof_t0);
                        SS_PROF_ADD(3, prof_tx - prof_t1);
                        continue;
                    }
                }
            }
            bool hit = false;
#ifdef SS_CAND_PROF
            unsigned long long prof_refine = 0, prof_verify = 0;
            SS_PROF_T(prof_t2);
#endif
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t *g = G[u];
                SS_PROF_T(prof_p0);
                if (!ONE_BYTE && per_piece) {
                    if (((pm >> u) & 1u) == 0) continue;
                    NextPiece np;
                    np.N = u + 1 < U ? A[u + 1] : H;
                    // lane 63's next lane: lane 0 of the next piece (rotated in); after the last piece the halo chunk
                    // sitting in lane 63 (MODE 0) or lanes 0..d of H (MODE 2)
                    np.kind = u + 1 < U ? 1 : (MODE == 0 ? 0 : 1);
                    // With the needle's dwords at hand the exact compare below settles a lane's candidates in ~50 VALU
                    // operations, all lanes at once - about what TWO steps of the byte-wise schedule cost - and a true match
                    // would sit through every one of its up to 13 steps first (a microsecond of ballots and branches).
                    if (!(EXACT_OK && exact_len != 0) && !refine_piece(A[u], np, ro, g)) {
#ifdef SS_CAND_PROF
                        prof_refine += __builtin_readcyclecounter() - prof_p0;
#endif
                        continue;
                    }
                } else if (__ballot(((g[0] | g[1] | g[2] | g[3]) & 0x80808080u) != 0) == 0) {
                    continue;
                }
                SS_PROF_T(prof_p1);
#ifdef SS_CAND_PROF
                prof_refine += prof_p1 - prof_p0;
#endif
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
                    // (the caller's far filter byte, if any: read where it is used - carried in registers it cost the kernels
                    // 17 spilled scalar registers for a field that is zero for every constructor-built searcher)
                    const uint64_t far_off = MODE == 0 && !ONE_BYTE ? uniform64(cold()->far_off) : 0;
                    h = verify_flags<ONE_BYTE>(g, chunk0 + 64 * u + lane, pr, va, s_needle, where, far_off);
                }
                hit |= h;
#ifdef SS_CAND_PROF
                prof_verify += __builtin_readcyclecounter() - prof_p1;
#endif
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
                        if (WG_FIND && wg_sink != nullptr) {
                            if (lane == 0)
                                __hip_atomic_fetch_min(static_cast<uint64_t *>(wg_sink), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else if (lane == 0 && mine < __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            if constexpr (SingleLaunchPlan<ColdT>::value) {
                                // a plan's run: the state word FIRST (the old value tells the first finder of the problem), the
                                // caller's output behind it - in that order: the workgroup that initialises
                                // the output re-reads the state word afterwards (scan_batched_plan_kernel)
                                const uint64_t old = __hip_atomic_fetch_min(best, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                // (the atomic has RETURNED - performed at the device's coherence point - before the output is touched.
                                // A wait, not a fence: a release / acquire fence at agent scope writes back and invalidates the whole
                                // L2 of the XCD, which is what the ordering of two device-scope atomics does not need)
                                __asm__ volatile("s_waitcnt vmcnt(0)" : : "v"(old) : "memory");
                                const auto c = cold();
                                uint64_t *out_best = reinterpret_cast<uint64_t *>(c->host_flag);
                                if (out_best) __hip_atomic_fetch_min(out_best, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (old == ~0ull && c->tally) __hip_atomic_fetch_add(c->tally, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            } else {
                                __hip_atomic_fetch_min(best, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if constexpr (LAZY_ORDER) {
                                    // (the unplanned batched find: `best` is the problem's state word in its cold record, which the
                                    // waves poll; the caller's output takes the minimum too - nobody polls that one)
                                    uint64_t *out_best = reinterpret_cast<uint64_t *>(cold()->host_flag);
                                    if (out_best) __hip_atomic_fetch_min(out_best, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                            }
                        }
                        forget_scalar_cache_unless(small_grid || kNoForget<LAZY_ORDER>);
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
                        if constexpr (SingleLaunchPlan<ColdT>::value) {
                            // a plan's run: the exchange has RETURNED (the state word is set) before the caller's output is
                            // written behind it; the first finder of the problem counts it into the plan's tally
                            if (old != pr.epoch) {                  // (the exchange has returned: performed at the device's coherence point)
                                const auto c = cold();
                                if (c->host_flag) (void)__hip_atomic_exchange(c->host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (c->tally) __hip_atomic_fetch_add(c->tally, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        } else {
                            int *host_flag = cold()->host_flag;
                            if (old != pr.epoch && host_flag)
                                __hip_atomic_store(host_flag, pr.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                    forget_scalar_cache_unless(small_grid || kNoForget<LAZY_ORDER>);
                    return;
                }
            }
#ifdef SS_CAND_PROF
            {
                SS_PROF_T(prof_tz);
                SS_PROF_ADD(0, 1);
                SS_PROF_ADD(1, prof_tz - prof_t0);
                SS_PROF_ADD(2, prof_t1 - prof_t0);
                SS_PROF_ADD(3, prof_refine);
                SS_PROF_ADD(4, prof_verify);
                SS_PROF_ADD(5, prof_t2 - prof_t1);
                if (per_piece) SS_PROF_ADD(6, 1);
            }
#endif
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

}  // namespace ss
