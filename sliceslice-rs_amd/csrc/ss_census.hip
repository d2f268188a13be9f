// ss_census.hip - what a searcher learns about a haystack by ASKING it (launch tuning; no search result depends on any of this):
//   * the candidate census (census_kernel): how often the searcher's filter bytes fire on this haystack -> four or six workgroups
//     per CU, the cross-lane kernels with or without their third byte;
//   * the byte histogram of the same sample (hist_sample_kernel), kept per DEVICE and shared by every searcher that meets the
//     haystack -> for searchers built by ss_searcher_new, whose caller did not choose, the three filter bytes themselves: row f3
//     of SURVEY.md 8f ("pick the needle bytes with the lowest corpus frequency"; the reference leaves `position` to its caller,
//     /root/reference/src/x86.rs:252-255) without the caller having to ask for a histogram.
//   * SURVIVAL (measured, not modelled): per needle position, how many of the sampled pair / triple candidates
//     match the needle there -> the third first-phase byte where the library owns it (the position that lets the fewest pair
//     candidates through), the near bytes that replace a far pair's second byte (one load stream instead of the cross-lane
//     kernels), and the ORDER of the second level's schedule (the byte that kills most of today's candidates first).
// Both kernels are enqueued in front of a scan on the scan's own stream and are never waited for by the host: the first scan of
// a (searcher, haystack) pair goes by static guesses, later ones by what has arrived.  ss_set_autotune(0) / SLICESLICE_AUTOTUNE=0
// switches all of it off (static choices only, no sampling kernels); ss_searcher_tuning_state reports what a handle holds.
// There is no CPU search path in this file.
#include "ss_internal.hpp"

#include <algorithm>

#define SS_AUX_CENSUS 1
#include "aux_kernels.hpp"

namespace ssh {

namespace {

// Workgroups per CU.  Four suit a scan that rarely meets a candidate, six one that keeps meeting them (ss_scan.hip, pick_variant,
// has the measurements), and which of the two a haystack is cannot be told from the needle: a text-like needle on binary data gave
// up 3-5 % under a needle-byte guess, a stock phrase of the manual with rare-looking bytes 15-20 % the other way.  Round 4 LEARNED
// the setting from the wall-clock time of a searcher's own full scans; its own records showed it misjudging by up to 10 % (a 1.5 %
// threshold against 1 % timing noise and 2-3 % drift), its explorations landed inside timed regions, and a call's cost depended on
// the calls before it.  Now the haystack is ASKED: the first scan of a (searcher, haystack) pair of at least kCensusMinBytes is
// preceded by census_kernel (aux_kernels.hpp), which puts kCensusTiles sampled wave-tiles through the searcher's own filter bytes
// and counts the tiles that hold a candidate.  From the second scan on the count decides: deterministic for a given haystack and
// needle, nothing in the scan kernels, nothing timed.  A searcher whose latest synchronous search FOUND the needle launches with
// four: a grid that leaves early drains faster with fewer workgroups resident (`the` on 1 GiB of text: 0.035 ms at four, 0.060 at
// six).
//
// Six workgroups per CU when at least kCensusDenseTiles of the kCensusTiles sampled tiles hold a candidate of the device's filter,
// or when the candidates crowd (kCensusDenseLanes candidate lanes in the sample).  Read from 96 (phrase, filter) cases on 1 GiB of
// the i386 text and on random bytes, each timed at forced four and six in one process (tools/occ_census.py,
// profiles/r05/occ_census_thresholds.jsonl): below ~40 candidate tiles in 1,024 four is 3-8 % faster, above ~70 six is - by 3 % at
// 70, 10-30 % from 150 on - and in between the two are within 3 % of each other; any threshold from 40 to 56 loses 0.3 % on average
// over the set against always picking the faster one (four everywhere: 7 %, six everywhere: 3 %).
//
// Round 6, second look (profiles/r06/shape_probe_1g.jsonl, _4g, _256m: 48 settled (phrase, triple) cases of the survival probe under
// forced workgroups per CU x TILES PER WORKGROUP, taking turns in one process): a wave that meets a candidate ends later than its three
// neighbours and the workgroup's slot is held until it does; with TWO tiles per workgroup the delay is spread over a workgroup that
// lives twice as long.  At four workgroups per CU that is worth 1-6 % between ~28 and ~56 candidate tiles of 1,024 at every size
// (1 GiB, 44 tiles: 0.865-0.870 -> 0.892-0.899 of the peak; 54 tiles: 0.846 -> 0.906, five workgroups of one tile: 0.883), below
// that one tile per workgroup stays ahead by 1-2 % at 1 GiB.  From ~56 tiles FIVE workgroups per CU of ONE tile are ahead - up to ~80
// tiles also from 2 GiB up, where a launch otherwise takes two tiles per workgroup (4 GiB, 63-78 tiles: 0.935-0.943 with one tile,
// 0.910-0.922 with two) - and six from ~160 (1 GiB alone would say ~220, 256 MiB and 4 GiB ~110) or when the candidates are DEEP ones by the
// hundred (a needle of box-drawing bytes, 869 deep candidates in the sample: 0.786 at five, 0.862 at six).
constexpr uint32_t kCensusTwoTilesFrom = 28;       // four workgroups per CU, two tiles each from here ...
constexpr uint32_t kCensusDenseTiles = 56, kCensusDenseLanes = 256;               // ... five of one tile from here
constexpr uint32_t kCensusVeryDenseTiles = 160, kCensusVeryDenseLanes = 1024;     // six workgroups per CU from here
constexpr uint32_t kCensusVeryDeepLanes = 256;     // ... or from this many deep candidates in the sample
// ... and ONE tile per workgroup at five / six workgroups per CU below these counts (it is what a launch below 2 GiB takes anyway; from
// 2 GiB up: 4 GiB, 52-69 tiles at five: +0.8-3.1 % with one tile, 81-123 tiles: -0.6-4.7 %; at six, needles of blanks - hundreds of
// candidate tiles, thousands of lanes - lose 5-10 % with one: profiles/r06/ab_text_shapes_v1_4g.jsonl; and gain 4-9 % with TWO below
// 2 GiB, where a launch takes one by itself - 1 GiB, 400 / 508 / 840 candidate tiles: 0.831 / 0.746 / 0.779 -> 0.866 / 0.813 / 0.831,
// profiles/r06/shape_probe_dense_1g.jsonl, columns 6x1 and 20x2: six workgroups per CU of two tiles)
constexpr uint32_t kCensusOneTileBelowAtFive = 80, kCensusOneTileBelowAtSix = 330;     // (six: 214-275 tiles want one tile at 1 and 4 GiB, 385-400 two: shape_probe_dense_*.jsonl)
// Filter pairs 16 or more apart (ss_searcher_set_filter3 only; the cross-lane kernels): the third first-phase byte pays on text,
// where the reference's own pair (0, n-1) passes at percent rates, and costs where the pair alone rarely matches (random bytes:
// equal at 1 GiB, 5-6 % at 8 GiB; profiles/r05/mode3_probe.jsonl).  The pair runs alone (MODE 3) when at most this many of the
// sampled tiles hold a candidate of the PAIR (random bytes: ~63 of 1,024; text: 900 and more).
constexpr uint32_t kCensusSparsePairTiles = 128;
constexpr uint32_t kCensusRefreshEvery = 256;      // scans of one (searcher, haystack) pair between two censuses of it
// Which bytes to filter on.  ss_searcher_new ranks the needle's bytes by a static, corpus-free guess (letters common, everything
// outside text rare: scan_filters.hpp byte_rarity_rank) - right for English text and binaries, exactly wrong where "rare-looking"
// bytes are the haystack's most frequent ones (UTF-8 text in a non-Latin script: every other byte is 0xD0 / 0xD1; box-drawing
// tables; padding patterns).  With the haystack's own histogram the triple is chosen again (ss_choose_filter_triple: cost of a byte
// = log2 of its count), and adopted for THIS haystack when it promises at least kTripleGainLog2 binary orders of magnitude fewer
// candidates than the static one - below that the static triple stays, so that a searcher does not flip between near-equal
// triples.  ss_searcher_filter3 keeps reporting the searcher's own triple; with_position and set_filter3 searchers keep theirs.
constexpr int kTripleGainLog2 = 4;                 // 16 x fewer expected candidates (byte costs are 8 * log2(count))
// The third byte by MEASUREMENT: pair_match[k] of the census is the number of sampled pair candidates that a first phase with k as
// its third byte lets through.  Moved when the best position at least halves what the current third byte lets through, and only
// when that is enough to matter (kThirdMinLanes of the 65,536 sampled lanes).
constexpr uint32_t kThirdMinLanes = 16;
// The near form of a far pair (propose_near_form) is kept when it meets candidates in at most this many more of the 1,024 sampled
// tiles than the pair did: the single-stream kernels are 2-3 % faster than the cross-lane ones before any candidate is met, which is
// what some 64 candidate tiles in 1,024 cost (profiles/r06/survival_probe_v1.jsonl).
constexpr uint32_t kNearFormSlackTiles = 64;
// What a proposal is judged by: its candidate lanes, a DEEP candidate - one that only the compare in memory can tell from a match: a
// dependent round trip with the workgroup's slot held - counting as kDeepWeight of them.  (profiles/r06/survival_probe_v2.jsonl: the
// i386 phrase through with_position(n-1) lost 3 % to a third byte with a quarter of the candidates, every one of them deep.)
constexpr uint32_t kDeepWeight = 8;
constexpr uint32_t kCensusDeepLanes = 24;          // five workgroups per CU from this many deep candidates in the sample
// The COMPACT form of a filter that has nothing to do (propose_compact): with no candidate in the sample the choice of bytes decides
// nothing but what the first phase costs, and that is less when the two further bytes lie within kCompactSpan bytes of the first - the
// next lane's dwords 0 and 1 instead of 0 ... 3 (one DPP move and one v_alignbyte window less per piece and dword).  Measured on 64 GiB
// of random bytes, fourteen pinned triples of one needle taking turns in one process (profiles/r06/headline_triple_probe_windows.jsonl):
// 0.9166-0.9200 of the peak with a byte 12-15 behind the first, 0.9245-0.9256 with the farthest 8-11 behind, 0.9267-0.9276 within 7;
// the same order at 8 GiB (0.920-0.924 / 0.925-0.929 / 0.928-0.931).  Proposed only from 12 bytes of span on.
constexpr size_t kCompactSpan = 7, kCompactFromSpan = 12;
constexpr uint32_t kDescentMaxRounds = 14;         // censuses a (searcher, haystack) pair may spend on improving its bytes, per 256 scans
constexpr uint32_t kOrderMinLanes = 4;             // triple candidates in the sample below which the static schedule order stays
static_assert(ss::kCensusStatWords == 2 * 64 + 3 && ss::kCensusCheck == 64, "PerDevice::Census and the control blocks are laid out for 64 positions");

struct CensusCounts {
    uint32_t tiles3, tiles2, match_tiles, lanes;
};
CensusCounts census_counts(uint64_t sums)
{
    return {(uint32_t)(sums & 2047u), (uint32_t)((sums >> 11) & 2047u), (uint32_t)((sums >> 22) & 2047u), (uint32_t)(sums >> 33)};
}

// The second level's schedule by MEASUREMENT: the needle bytes of the 31 behind the first filter byte `fa` (the first-phase bytes
// fb, fc left out), the position at which the FEWEST of the sampled triple candidates match the needle first - every step is then
// the one that kills most of what the haystack really holds - ties and positions beyond the census's 64 bytes by the static rarity
// rank.  Same packing as ss::build_refine_order.  Only the ORDER of necessary conditions: results cannot depend on it.
void build_measured_order(const ss_searcher *s, size_t fa, size_t fb, size_t fc, const uint16_t *triple_match, uint32_t *norder,
                          uint64_t idx[2], uint64_t val[2])
{
    struct Cand {
        uint32_t K, count;
        int rank;
    } cand[ss::kRefineWindow];
    uint32_t m = 0;
    const size_t lim = std::min<size_t>(s->n - fa, (size_t)ss::kRefineWindow);
    for (size_t K = 1; K < lim; ++K) {
        const size_t k = fa + K;
        if (k == fb || k == fc) continue;
        cand[m++] = {(uint32_t)K, k < ss::kCensusCheck ? (uint32_t)triple_match[k] : 0x10000u, ss::byte_rarity_rank(s->needle[k])};
    }
    std::stable_sort(cand, cand + m, [](const Cand &x, const Cand &y) { return x.count != y.count ? x.count < y.count : x.rank < y.rank; });
    idx[0] = idx[1] = val[0] = val[1] = 0;
    const uint32_t take = std::min<uint32_t>(m, 15u);
    for (uint32_t t = 0; t < take; ++t) {
        idx[t >> 3] |= (uint64_t)cand[t].K << (8 * (t & 7));
        val[t >> 3] |= (uint64_t)s->needle[fa + cand[t].K] << (8 * (t & 7));
    }
    *norder = take;
}

// first-phase triple as fill_problem / the kernels want it: the smallest index first
void normalised(const size_t slot[3], size_t tri[3])
{
    int lo = 0;
    for (int k = 1; k < 3; ++k)
        if (slot[k] < slot[lo]) lo = k;
    tri[0] = slot[lo];
    tri[1] = slot[(lo + 1) % 3];
    tri[2] = slot[(lo + 2) % 3];
    if (tri[1] == tri[0]) std::swap(tri[1], tri[2]);           // (a needle of two bytes: second == third)
}

bool same_triple(const size_t a[3], const size_t b[3])
{
    size_t x[3], y[3];
    normalised(a, x);
    normalised(b, y);
    return x[0] == y[0] && ((x[1] == y[1] && x[2] == y[2]) || (x[1] == y[2] && x[2] == y[1]));
}

// The census in flight, if its counts have arrived.  A census of cur[] refreshes the counts, the per-position match counts and
// the schedule order.  A census of a PROPOSAL is its trial: the match counts that suggested it are conditional on the bytes they
// were taken with (and the histogram prices bytes as independent); the census counts what really happens, and only a triple that
// really meets fewer candidates replaces cur[].
void complete_pending(const ss_searcher *s, PerDevice *pd)
{
    if (pd->census_pending < 0) return;
    PerDevice::Census &c = pd->census[pd->census_pending];
    if (__atomic_load_n(pd->h_census + 1, __ATOMIC_ACQUIRE) != (unsigned long long)c.tag) return;
    const uint64_t sums = __atomic_load_n(pd->h_census, __ATOMIC_RELAXED);
    pd->census_pending = -1;
    const uint32_t what = c.inflight;
    c.inflight = 0;
    if (c.gen != s->filter_gen) return;
    bool take = what == 1;
    if (what == 2) {
        const CensusCounts now = census_counts(sums), was = census_counts(c.sums);
        const uint32_t deep_now = __atomic_load_n(pd->h_stats + 2 * ss::kCensusCheck + 2, __ATOMIC_RELAXED);
        const uint64_t cost_now = (uint64_t)now.lanes + (uint64_t)kDeepWeight * deep_now, cost_was = (uint64_t)was.lanes + (uint64_t)kDeepWeight * c.deep_lanes;
        bool better;
        if (c.prop_kind == 3) better = now.tiles3 <= was.tiles3 + kNearFormSlackTiles && deep_now <= c.deep_lanes + kNearFormSlackTiles / 4;   // one load stream instead of the cross-lane kernels
        else if (c.prop_kind == 1) better = now.tiles3 <= was.tiles3 && cost_now <= cost_was;
        else if (c.prop_kind == 5) better = now.tiles3 == 0 && now.lanes == 0 && now.match_tiles == 0 && deep_now == 0;   // still nothing to do
        else better = now.tiles3 <= was.tiles3 && cost_now < cost_was;
        if (better) {
            if (c.prop_kind == 3) c.free_mask = 6;      // the near form keeps the caller's first byte; the other two are the library's
            c.cur[0] = c.prop[0];
            c.cur[1] = c.prop[1];
            c.cur[2] = c.prop[2];
            const size_t own[3] = {s->da, s->db, s->dc};
            c.adopted = !same_triple(c.cur, own);
            c.stale = 0;
            ++c.accepted;
            take = true;
        } else {
            ++c.stale;                                  // (the counters at hand describe the rejected triple; cur[]'s order stays)
            if (c.prop_kind == 4 || c.prop_kind == 5) c.stale = (uint32_t)__builtin_popcount(c.free_mask);      // a rejected jump / compact form ends the look
            c.stats_roles = -1;
        }
    }
    if (!take) return;
    c.sums = sums;
    c.state = 2;
    for (uint32_t k = 0; k < ss::kCensusCheck; ++k) {
        c.pair_match[k] = (uint16_t)std::min<uint32_t>(__atomic_load_n(pd->h_stats + k, __ATOMIC_RELAXED), 0xFFFFu);
        c.triple_match[k] = (uint16_t)std::min<uint32_t>(__atomic_load_n(pd->h_stats + ss::kCensusCheck + k, __ATOMIC_RELAXED), 0xFFFFu);
    }
    c.pair_lanes = __atomic_load_n(pd->h_stats + 2 * ss::kCensusCheck, __ATOMIC_RELAXED);
    c.triple_lanes = __atomic_load_n(pd->h_stats + 2 * ss::kCensusCheck + 1, __ATOMIC_RELAXED);
    c.deep_lanes = __atomic_load_n(pd->h_stats + 2 * ss::kCensusCheck + 2, __ATOMIC_RELAXED);
    c.stats_roles = c.roles;
    size_t tri[3];
    normalised(c.cur, tri);
    c.have_order = false;
    if (c.triple_lanes >= kOrderMinLanes && tri[1] - tri[0] <= 15) {    // (the cross-lane kernels keep the static order)
        build_measured_order(s, tri[0], tri[1], tri[2], c.triple_match, &c.norder, c.order_idx, c.order_val);
        c.have_order = c.norder != 0;
    }
}

bool stream_is_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return true;                                    // (a graph would replay the sampling kernels for nobody)
    }
    return false;
}

// ---- per-device haystack histograms ---------------------------------------------------------------------------------------
struct HayStats {
    const void *hay = nullptr;
    size_t len = 0;
    uint32_t state = 0;         // 0 = empty, 1 = launched (tag `tag`), 2 = the histogram is in
    uint32_t tag = 0;
    uint32_t uses = 0;
    uint64_t stamp = 0;
    uint64_t hist[256];
};
struct DeviceStats {
    std::mutex mu;
    HayStats e[4];
    int pending = -1;
    uint32_t tag = 0;
    uint64_t clock = 0;
    uint32_t *d_partial = nullptr;          // kHistBlocks x 256
    unsigned *d_counter = nullptr;
    unsigned long long *h_out = nullptr;    // pinned: tag, then 256 x uint32
    bool broken = false;                    // the scratch could not be allocated: no histograms on this device
};
DeviceStats *device_stats()                 // never destroyed (a search may come from a thread that outlives main)
{
    static DeviceStats *const t = new DeviceStats[kMaxDevices];
    return t;
}

// The histogram of (hay, len) on device `dev` if it is in (copied to `out`); launches the sampling in front of the caller's scan
// when nothing is known and nothing is in flight on the device.
bool stats_lookup(int dev, const void *d_hay, size_t len, hipStream_t st, uint64_t out[256])
{
    if (dev < 0 || dev >= kMaxDevices) return false;
    DeviceStats &ds = device_stats()[dev];
    std::unique_lock<std::mutex> lock(ds.mu, std::try_to_lock);
    if (!lock.owns_lock() || ds.broken) return false;
    if (ds.pending >= 0 && ds.h_out && __atomic_load_n(ds.h_out, __ATOMIC_ACQUIRE) == (unsigned long long)ds.e[ds.pending].tag) {
        HayStats &c = ds.e[ds.pending];
        const uint32_t *h = reinterpret_cast<const uint32_t *>(ds.h_out + 1);
        for (int b = 0; b < 256; ++b) c.hist[b] = __atomic_load_n(h + b, __ATOMIC_RELAXED);
        c.state = 2;
        ds.pending = -1;
    }
    HayStats *hit = nullptr, *victim = &ds.e[0];
    for (auto &c : ds.e) {
        if (c.state != 0 && c.hay == d_hay && c.len == len) hit = &c;
        if (c.state != 1 && (victim->state == 1 || c.stamp < victim->stamp)) victim = &c;
    }
    bool have = false;
    HayStats *target = victim;
    if (hit) {
        hit->stamp = ++ds.clock;
        if (hit->state != 2) return false;
        memcpy(out, hit->hist, sizeof hit->hist);
        have = true;
        if (++hit->uses % (4 * kCensusRefreshEvery) != 0) return true;           // (a buffer may be refilled in place)
        target = hit;
    }
    if (ds.pending >= 0 || (!hit && victim->state == 1)) return have;
    if (len < 2 * (size_t)ss::kCensusTileBytes + 16) return have;
    const uint64_t stride = ((len - 8 - ss::kCensusTileBytes) / (ss::kCensusTiles - 1)) & ~(uint64_t)(ss::kCensusTileBytes - 1);
    if (stride < ss::kCensusTileBytes || stream_is_capturing(st)) return have;
    if (!ds.d_partial) {                                                          // first use on this device
        // (zeroed on the CALL'S stream, in front of the sampling kernel that follows on it: a null-stream memset would be a
        // device-wide ordering point inside a search, whose contract is to synchronise nothing but the stream it was given.  The
        // next sampling - possibly on another stream - is not launched before this one's tag has arrived: `pending`.)
        hipError_t e = hipMalloc((void **)&ds.d_partial, ss::kHistBlocks * 256 * sizeof(uint32_t) + 64);
        if (e == hipSuccess) e = hipMemsetAsync(ds.d_partial, 0, ss::kHistBlocks * 256 * sizeof(uint32_t) + 64, st);
        if (e == hipSuccess) e = hipHostMalloc((void **)&ds.h_out, sizeof(unsigned long long) + 256 * sizeof(uint32_t), hipHostMallocPortable);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            ds.broken = true;
            return have;
        }
        ds.d_counter = reinterpret_cast<unsigned *>(ds.d_partial + ss::kHistBlocks * 256);
        ds.h_out[0] = 0;
    }
    ss::HistArgs a;
    a.hay = static_cast<const uint8_t *>(d_hay);
    a.stride = stride;
    if (++ds.tag == 0) ds.tag = 1;
    a.tag = ds.tag;
    a.d_partial = ds.d_partial;
    a.d_counter = ds.d_counter;
    a.h_out = ds.h_out;
    ss::hist_sample_kernel<<<dim3(ss::kHistBlocks), dim3(ss::kBlock), 0, st>>>(a);
    if (hipGetLastError() != hipSuccess) return have;
    if (!hit) {
        target->hay = d_hay;
        target->len = len;
        target->state = 1;
        target->uses = 0;
        target->stamp = ++ds.clock;
    }
    target->tag = a.tag;
    ds.pending = (int)(target - ds.e);
    return have;
}

// With the haystack's histogram: the triple ss_choose_filter_triple would pick, if it beats the searcher's own by kTripleGainLog2.
bool better_triple(const ss_searcher *s, const uint64_t hist[256], size_t tri[3])
{
    size_t a = 0, b = 0, c = 0;
    if (ss_choose_filter_triple(s->needle.data(), s->n, hist, &a, &b, &c) != SS_OK) return false;
    if (b < a || b - a > 15 || c <= a || c - a > 15 || c == b) return false;       // (needles of two bytes: nothing to choose)
    const ByteCost cost(hist);
    const int own = cost(s->needle[s->da]) + cost(s->needle[s->db]) + cost(s->needle[s->dc]);
    const int alt = cost(s->needle[a]) + cost(s->needle[b]) + cost(s->needle[c]);
    if (own - alt < 8 * kTripleGainLog2) return false;
    tri[0] = a;
    tri[1] = b;
    tri[2] = c;
    return true;
}

// Which of a searcher's three first-phase bytes the library may move on a haystack (bit j = slot j of {da, db, dc}):
//   ss_searcher_new                  all three;
//   ss_searcher_with_position        the caller's byte (slot 1) stays; the third byte is the library's, and so is the partner in front
//                                    of a position >= 16 (choose_anchor) - for position < 16 the partner is the reference's needle[0];
//   ss_searcher_set_filter3, a pair  the pair stays, the third byte is the library's; a pair 16 or more apart additionally has a NEAR
//                                    FORM (below) whose two other bytes are the library's;
//   ss_searcher_set_filter3, triple  nothing.
uint32_t free_slots(const ss_searcher *s)
{
    if (s->n < 3) return 0;
    if (s->auto_filter) return 7;
    if (!s->third_owned) return 0;
    if (s->db - s->da > 15) return 4;                   // (a far pair: its third byte; the rest only through its near form)
    return s->anchor_owned ? 5u : 4u;
}

// Slot j of cur[] moved to the needle position that, by the census's count, lets the fewest candidates through - when that at least
// halves what the current byte lets through and there are enough candidates to matter.  pair_match[] must have been gathered FOR
// slot j (the kernel's pair = the other two slots).  All three bytes stay within a 16-byte span.
bool propose_move(const ss_searcher *s, const PerDevice::Census &c, int j, size_t prop[3])
{
    const size_t u = c.cur[(j + 1) % 3], v = c.cur[(j + 2) % 3], now_at = c.cur[j];
    const size_t lo = std::min(u, v);
    size_t hi = std::max(u, v);
    if (hi - lo > 15) {                                 // a far pair: its third byte lives within 15 bytes of the first
        if (j != 2) return false;
        hi = lo;
    }
    if (now_at >= ss::kCensusCheck) return false;
    const size_t k0 = hi >= 15 ? hi - 15 : 0, k1 = std::min<size_t>(std::min<size_t>(s->n, ss::kCensusCheck), lo + 16);
    uint32_t best = ~0u;
    size_t at = now_at;
    for (size_t k = k0; k < k1; ++k) {
        if (k == u || k == v) continue;
        if ((uint32_t)c.pair_match[k] <= best) {        // (ties to the later byte, as everywhere)
            best = c.pair_match[k];
            at = k;
        }
    }
    const uint32_t now = c.pair_match[now_at];
    if (best == ~0u || at == now_at || now < kThirdMinLanes || (uint64_t)best * 2 > now) return false;
    prop[0] = c.cur[0];
    prop[1] = c.cur[1];
    prop[2] = c.cur[2];
    prop[j] = at;
    return true;
}

// A JUMP, where moving one byte at a time cannot get there: the needle position at which the FEWEST of today's candidates match
// (triple_match: the byte that tells the needle from the stock phrase the haystack is full of) made a first-phase byte, with
// partners close to it - a caller's byte if there is one (it must lie within 15 bytes), else the positions around it with the
// smallest counts, rarer bytes first among equals.  On trial like every proposal.
bool propose_jump(const ss_searcher *s, const PerDevice::Census &c, size_t prop[3])
{
    if (c.stats_roles < 0 || c.triple_lanes < kThirdMinLanes) return false;
    const size_t lim = std::min<size_t>(s->n, ss::kCensusCheck);
    size_t kstar = lim;
    uint32_t best = ~0u;
    for (size_t k = 0; k < lim; ++k)
        if ((uint32_t)c.triple_match[k] < best) {
            best = c.triple_match[k];
            kstar = k;
        }
    if (kstar >= lim || (uint64_t)best * 4 > c.triple_lanes) return false;                 // nothing kills three quarters of them
    if (kstar == c.cur[0] || kstar == c.cur[1] || kstar == c.cur[2]) return false;
    // the bytes that must stay: the caller's (with_position: slot 1)
    size_t keep[2];
    int nkeep = 0;
    for (int j = 0; j < 3; ++j)
        if (!((c.free_mask >> j) & 1u)) keep[nkeep++] = c.cur[j];
    if (nkeep >= 2) return false;
    size_t chosen[3] = {kstar, nkeep == 1 ? keep[0] : lim, lim};
    int have = nkeep == 1 ? 2 : 1;
    auto span_ok = [&](size_t k) {
        size_t lo = k, hi = k;
        for (int t = 0; t < have; ++t) {
            lo = std::min(lo, chosen[t]);
            hi = std::max(hi, chosen[t]);
        }
        return hi - lo <= 15;
    };
    if (nkeep == 1 && !span_ok(kstar)) return false;
    while (have < 3) {
        size_t at = lim;
        uint32_t bc = ~0u;
        int br = INT_MAX;
        for (size_t k = 0; k < lim; ++k) {
            bool taken = false;
            for (int t = 0; t < have; ++t) taken = taken || chosen[t] == k;
            if (taken || !span_ok(k)) continue;
            const uint32_t cnt = c.triple_match[k];
            const int rank = ss::byte_rarity_rank(s->needle[k]);
            if (cnt < bc || (cnt == bc && rank <= br)) {
                bc = cnt;
                br = rank;
                at = k;
            }
        }
        if (at >= lim) return false;
        chosen[have++] = at;
    }
    // slots: a kept byte stays in its slot (slot 1 for with_position); the others fill the free slots
    if (nkeep == 1) {
        prop[1] = keep[0];
        prop[0] = chosen[0];
        prop[2] = chosen[2];
    } else {
        prop[0] = chosen[0];
        prop[1] = chosen[1];
        prop[2] = chosen[2];
    }
    return true;
}

// A pair 16 or more apart (ss_searcher_set_filter3 only - the reference's pair (0, n-1) of a long needle) runs in the cross-lane
// kernels.  Its NEAR FORM: the caller's first byte and the TWO positions of the 15 behind it at which the fewest of the pair's
// candidates match the needle - one load stream (the single-stream kernels), the caller's far byte left to the second level and the
// compare (still a necessary condition, tested before a candidate is reported).  Put on trial like every proposal; kept when it does
// not meet candidates in noticeably more tiles than the pair did.
bool propose_near_form(const ss_searcher *s, const PerDevice::Census &c, size_t prop[3])
{
    const size_t fa = s->da;
    if (s->db - fa <= 15 || !s->third_owned || fa + 2 >= s->n) return false;
    auto best_of = [&](size_t skip, size_t *at) {
        uint32_t best = ~0u;
        for (size_t k = fa + 1; k < s->n && k <= fa + 15 && k < ss::kCensusCheck; ++k) {
            if (k == skip) continue;
            if ((uint32_t)c.pair_match[k] <= best) {
                best = c.pair_match[k];
                *at = k;
            }
        }
        return best;
    };
    size_t k1 = fa, k2 = fa;
    if (best_of(fa, &k1) == ~0u || best_of(k1, &k2) == ~0u) return false;
    prop[0] = fa;
    prop[1] = k1;
    prop[2] = k2;
    return true;
}

// The COMPACT form (see kCompactSpan): the bytes the library owns re-chosen so that all three lie within kCompactSpan of the smallest
// index - the cheapest such set by `cost` (the haystack's histogram when it is in, else the static classes), later positions among
// equals; a caller's bytes stay where they are (no compact form when they alone span more).  On trial like every proposal: kept only if
// its census meets no candidate either.
bool propose_compact(const ss_searcher *s, const PerDevice::Census &c, const ByteCost &cost, size_t prop[3])
{
    if (s->n < 3 || c.free_mask == 0) return false;
    size_t tri[3];
    normalised(c.cur, tri);
    if (std::max(tri[1], tri[2]) - tri[0] < kCompactFromSpan || tri[1] - tri[0] > 15) return false;
    const size_t lim = std::min<size_t>(s->n, ss::kCensusCheck);
    size_t lo_fixed = lim, hi_fixed = 0;
    for (int j = 0; j < 3; ++j)
        if (!((c.free_mask >> j) & 1u)) {
            lo_fixed = std::min(lo_fixed, c.cur[j]);
            hi_fixed = std::max(hi_fixed, c.cur[j]);
        }
    const bool any_fixed = lo_fixed <= hi_fixed;
    if (any_fixed && (hi_fixed - lo_fixed > kCompactSpan || hi_fixed >= lim)) return false;
    // candidates for a free slot: everything that can share a window of kCompactSpan with the fixed bytes (or anything, with none fixed)
    const size_t k0 = any_fixed ? (hi_fixed >= kCompactSpan ? hi_fixed - kCompactSpan : 0) : 0;
    const size_t k1 = any_fixed ? std::min(lim, lo_fixed + kCompactSpan + 1) : lim;
    int best = INT_MAX;
    size_t bt[3] = {0, 0, 0};
    size_t t[3];
    auto consider = [&]() {
        const size_t lo = std::min(t[0], std::min(t[1], t[2])), hi = std::max(t[0], std::max(t[1], t[2]));
        if (hi - lo > kCompactSpan || t[0] == t[1] || t[0] == t[2] || t[1] == t[2]) return;
        int total = 0;
        for (int j = 0; j < 3; ++j)
            if ((c.free_mask >> j) & 1u) total += cost(s->needle[t[j]]);
        if (total <= best) {                            // (later positions among equals, as everywhere)
            best = total;
            bt[0] = t[0];
            bt[1] = t[1];
            bt[2] = t[2];
        }
    };
    auto range = [&](int j, size_t *a, size_t *b) {     // positions slot j may take
        if ((c.free_mask >> j) & 1u) {
            *a = k0;
            *b = k1;
        } else {
            *a = c.cur[j];
            *b = c.cur[j] + 1;
        }
    };
    size_t a0, b0, a1, b1, a2, b2;
    range(0, &a0, &b0);
    range(1, &a1, &b1);
    range(2, &a2, &b2);
    for (t[0] = a0; t[0] < b0; ++t[0])
        for (t[1] = a1; t[1] < b1 && t[1] <= t[0] + kCompactSpan; ++t[1]) {
            if (t[1] + kCompactSpan < t[0]) continue;
            for (t[2] = a2; t[2] < b2; ++t[2]) consider();
        }
    if (best == INT_MAX) return false;
    if (same_triple(bt, c.cur)) return false;
    prop[0] = bt[0];
    prop[1] = bt[1];
    prop[2] = bt[2];
    return true;
}

}  // namespace

// ss_set_autotune / SLICESLICE_AUTOTUNE: everything this file does, on or off, process-wide.  Off: the constructors' static triple,
// the needle-byte guess for workgroups per CU, no sampling kernels, one plan layout - a call's cost then depends on its arguments
// alone (reproducible measurements).
namespace {
std::atomic<int> g_autotune{-1};
}
bool autotune_enabled()
{
    int v = g_autotune.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("SLICESLICE_AUTOTUNE");
        v = !(e && e[0] == '0');
        g_autotune.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}

namespace {

// One census of `slots` (cur[] or a proposal) in front of the caller's scan; per-position counts gathered FOR slot `roles`.
bool launch_census(const ss_searcher *s, PerDevice *pd, PerDevice::Census *c, const size_t slots[3], int roles, const void *d_hay, size_t len,
                   hipStream_t st)
{
    const size_t n = s->n, end = len - n + 1;
    if (end < 2 * (size_t)ss::kCensusTileBytes + 8) return false;
    const uint64_t room = end - 4 - ss::kCensusTileBytes;                                 // latest start of a sampled tile
    const uint64_t stride = (room / (ss::kCensusTiles - 1)) & ~(uint64_t)(ss::kCensusTileBytes - 1);
    if (stride < ss::kCensusTileBytes || stream_is_capturing(st)) return false;
    const size_t oa = slots[(roles + 1) % 3], ob = slots[(roles + 2) % 3], oc = slots[roles];
    ss::CensusArgs a;
    a.hay = static_cast<const uint8_t *>(d_hay);
    a.needle = pd->d_needle;
    a.stride = stride;
    a.oa = (uint32_t)oa;
    a.ob = (uint32_t)ob;
    a.oc = (uint32_t)oc;
    a.bytes = (uint32_t)s->needle[oa] | ((uint32_t)s->needle[ob] << 8) | ((uint32_t)s->needle[oc] << 16);
    a.ncheck = (uint32_t)std::min<size_t>(n, ss::kCensusCheck);
    a.nblocks = ss::kCensusTiles / ss::kWavesPerBlock;
    if (++pd->census_tag == 0) pd->census_tag = 1;
    a.tag = pd->census_tag;
    a.d_acc = pd->d_census;
    a.d_stats = pd->d_stats;
    a.h_out = pd->h_census;
    a.h_stats = pd->h_stats;
    ss::census_kernel<<<dim3(a.nblocks), dim3(ss::kBlock), 0, st>>>(a);
    if (hipGetLastError() != hipSuccess) return false;
    c->tag = a.tag;
    c->roles = roles;
    pd->census_pending = (int)(c - pd->census);
    return true;
}

int next_free(uint32_t mask, int after)
{
    for (int k = 1; k <= 3; ++k)
        if ((mask >> ((after + k) % 3)) & 1u) return (after + k) % 3;
    return 2;
}

}  // namespace

void launch_hints(const ss_searcher *s, PerDevice *pd, const void *d_hay, size_t len, hipStream_t st, LaunchHints *out)
{
    out->have_counts = false;
    out->have_triple = false;
    out->have_order = false;
    out->workgroups_per_cu = 0;
    out->tiles_per_workgroup = 0;
    out->sparse_pair = false;
    if (len < kCensusMinBytes || s->n < 2 || len < s->n || !autotune_enabled()) return;
    if (__atomic_exchange_n(&pd->census_lock, 1u, __ATOMIC_ACQUIRE) != 0) return;         // another thread is at it
    struct Unlock {
        uint32_t *w;
        ~Unlock() { __atomic_store_n(w, 0u, __ATOMIC_RELEASE); }
    } unlock{&pd->census_lock};
    complete_pending(s, pd);
    PerDevice::Census *c = nullptr, *victim = &pd->census[0];
    for (auto &e : pd->census) {
        if (e.state != 0 && e.hay == d_hay && e.len == len && e.gen == s->filter_gen) c = &e;
        if (e.inflight == 0 && (victim->inflight != 0 || e.stamp < victim->stamp)) victim = &e;
    }
    if (!c) {
        // first meeting: the pair only leaves its NAME - a searcher that scans a haystack once pays nothing for what it will never use
        // (the census and the histogram sampling are 10-13 us and 18-22 us of kernel time in front of the scan they are launched with,
        // a fifth of a 1 GiB scan's own time; the batched calls treat their batches the same way: ss_batched.hip)
        if (victim->inflight != 0) return;                                                // (every entry has a census in flight)
        PerDevice::Census fresh;
        fresh.hay = d_hay;
        fresh.len = len;
        fresh.gen = s->filter_gen;
        fresh.cur[0] = s->da;
        fresh.cur[1] = s->db;
        fresh.cur[2] = s->dc;
        fresh.free_mask = free_slots(s);
        fresh.coord = 2;
        fresh.stamp = ++pd->census_clock;
        fresh.state = 3;
        *victim = fresh;
        return;
    }
    c->stamp = ++pd->census_clock;
    if (c->state == 3) {
        // second meeting: a census of the searcher's own triple in front of this scan (and, for searchers built by ss_searcher_new, the
        // haystack's histogram next to it); the scan itself goes by the static choices
        if (pd->census_pending >= 0) return;                                              // one census in flight per searcher and device
        if (!launch_census(s, pd, c, c->cur, 2, d_hay, len, st)) return;                  // (a capturing stream: asked again next time)
        c->state = 1;
        c->inflight = 1;
        if (s->auto_filter && s->n >= 3) {
            uint64_t hist[256];
            (void)stats_lookup(pd->dev, d_hay, len, st, hist);
        }
        return;
    }
    if (c->adopted) {
        out->have_triple = true;
        normalised(c->cur, out->tri);
    }
    if (c->have_order) {
        out->have_order = true;
        out->norder = c->norder;
        out->order_idx[0] = c->order_idx[0];
        out->order_idx[1] = c->order_idx[1];
        out->order_val[0] = c->order_val[0];
        out->order_val[1] = c->order_val[1];
    }
    if (c->state != 2) return;                          // the first census is in flight
    const CensusCounts cc = census_counts(c->sums);
    out->have_counts = true;
    // four / five / six (round 6: profiles/r06/wg_probe.jsonl - pinned triples of 0 to 250 candidate tiles under forced shapes: four is
    // best up to ~45 candidate tiles in 1,024, FIVE from there to ~210 (3-5 % over six, 4-12 % over four), six beyond)
    out->workgroups_per_cu = cc.match_tiles != 0 ? 4
                             : (cc.tiles3 >= kCensusVeryDenseTiles || cc.lanes >= kCensusVeryDenseLanes || c->deep_lanes >= kCensusVeryDeepLanes ? 6
                                : (cc.tiles3 >= kCensusDenseTiles || cc.lanes >= kCensusDenseLanes || c->deep_lanes >= kCensusDeepLanes ? 5 : 4));
    // tiles per workgroup (single-stream kernels; 0 = the launch's own choice): two at four workgroups per CU once candidate tiles are
    // no rarity; ONE at five and six while the candidates are spread thinly enough for a workgroup to meet one or none - where they
    // crowd (needles of blanks, deep candidates by the hundred) longer-lived workgroups are ahead again
    out->tiles_per_workgroup = 0;
    if (cc.match_tiles == 0) {
        if (out->workgroups_per_cu == 4) out->tiles_per_workgroup = cc.tiles3 >= kCensusTwoTilesFrom ? 2 : 0;
        else if (out->workgroups_per_cu == 5) out->tiles_per_workgroup = cc.tiles3 < kCensusOneTileBelowAtFive ? 1 : 0;
        else out->tiles_per_workgroup = cc.tiles3 < kCensusOneTileBelowAtSix && cc.lanes < kCensusVeryDenseLanes && c->deep_lanes < kCensusVeryDeepLanes ? 1 : 2;
    }
    out->sparse_pair = cc.tiles2 <= kCensusSparsePairTiles;
    // A buffer may be refilled in place: everything is looked at again every kCensusRefreshEvery scans, starting from the bytes in
    // force (the old counts serve until the new ones are in).
    if (++c->uses % kCensusRefreshEvery == 0) {
        c->settled = c->near_tried = c->jump_tried = c->compact_tried = false;
        c->stale = c->rounds = 0;
        c->stats_roles = -1;
    }
    if (c->settled || c->inflight != 0 || pd->census_pending >= 0) return;
    // ---- the descent: one census per scan until no coordinate improves any more ------------------------------------------------
    if (c->rounds >= kDescentMaxRounds) {
        c->settled = true;
        return;
    }
    const bool far_own = !c->adopted && s->db - s->da > 15;
    size_t prop[3];
    uint32_t kind = 0;
    // (1) searchers built by ss_searcher_new: the triple the haystack's HISTOGRAM suggests, once, when it promises 16 x fewer candidates
    //     (... and the bytes in force meet candidates at all: a filter that never fires has nothing to gain - on random bytes the
    //     histogram would otherwise trade the searcher's triple for one with a byte that does not occur, to no effect)
    if (s->auto_filter && s->n >= 3 && !c->hist_tried && !c->adopted && cc.tiles3 != 0) {
        uint64_t hist[256];
        if (stats_lookup(pd->dev, d_hay, len, st, hist)) {
            c->hist_tried = true;
            if (better_triple(s, hist, prop)) kind = 1;
        }
    }
    // (2) a far pair: its near form, once its own counts (gathered for slot 2: the pair is the caller's) are in
    if (kind == 0 && far_own && c->stats_roles == 2 && !c->near_tried) {
        c->near_tried = true;                           // (once per look)
        if (propose_near_form(s, *c, prop)) kind = 3;
    }
    // (3) one coordinate moved by the census's own match counts
    int gather = -1;
    if (kind == 0) {
        const uint32_t nfree = (uint32_t)__builtin_popcount(c->free_mask);
        // (with fewer candidates in the sample than a move needs to be worth a trial - propose_move: kThirdMinLanes - no coordinate is
        // looked at: each look is a census in front of a scan, 10-13 us, and a handle on random bytes spent six of them on nothing)
        if (cc.lanes < kThirdMinLanes && c->stale < nfree) c->stale = nfree;
        for (uint32_t tries = 0; tries < 3 && kind == 0 && gather < 0; ++tries) {
            if (nfree == 0 || c->stale >= nfree) {
                c->settled = true;
                break;
            }
            const int j = ((c->free_mask >> c->coord) & 1u) ? (int)c->coord : next_free(c->free_mask, (int)c->coord);
            c->coord = (uint32_t)j;
            if (c->stats_roles != j) {
                gather = j;                             // counts for this coordinate first
            } else if (propose_move(s, *c, j, prop)) {
                kind = 2;
            } else {
                ++c->stale;
                c->coord = (uint32_t)next_free(c->free_mask, j);
            }
        }
        if (c->settled && kind == 0 && !c->jump_tried && (c->free_mask == 7u || c->free_mask == 5u) && c->stats_roles >= 0) {
            // no single byte improves any more: the jump (once per look), before the handle settles
            c->jump_tried = true;
            if (propose_jump(s, *c, prop)) {
                kind = 4;
                c->settled = false;
            }
        }
        if (c->settled && kind == 0 && !c->compact_tried && c->free_mask != 0 && cc.tiles3 == 0 && cc.lanes == 0 && cc.match_tiles == 0 &&
            c->deep_lanes == 0) {
            // a filter that meets no candidates on this haystack: its compact form (once per look), before the handle settles
            c->compact_tried = true;
            uint64_t hist[256];
            const bool have_hist = stats_lookup(pd->dev, d_hay, len, st, hist);
            if (propose_compact(s, *c, ByteCost(have_hist ? hist : nullptr), prop)) {
                kind = 5;
                c->settled = false;
            }
        }
        if (c->settled) {
            // the counts and the order at hand must describe cur[]: after a rejected trial they do not
            if (c->stats_roles >= 0) return;
            c->settled = false;
            gather = 2;
        }
    }
    if (kind != 0) {
        // the trial's census gathers its per-position counts for the coordinate that comes next, so that an accepted proposal goes on
        const int roles = kind == 2 ? next_free(c->free_mask, (int)c->coord) : 2;     // (a histogram / near-form / jump / compact proposal starts over at slot 2)
        if (!launch_census(s, pd, c, prop, roles, d_hay, len, st)) return;
        c->prop[0] = prop[0];
        c->prop[1] = prop[1];
        c->prop[2] = prop[2];
        c->prop_kind = kind;
        c->inflight = 2;
        ++c->trials;
        ++c->rounds;
        if (kind == 2) c->coord = (uint32_t)roles;
        else c->coord = 2;
        if (kind == 4) c->stale = 0;                    // (an accepted jump is followed by another look at every coordinate)
    } else if (gather >= 0) {
        if (!launch_census(s, pd, c, c->cur, gather, d_hay, len, st)) return;
        c->inflight = 1;
        ++c->rounds;
        if (c->stale >= (uint32_t)__builtin_popcount(c->free_mask)) c->settled = true;      // (this census only brings cur[]'s counts back)
    }
}

}  // namespace ssh

using namespace ssh;

extern "C" {

int ss_set_autotune(int enabled)
{
    const int before = autotune_enabled() ? 1 : 0;
    g_autotune.store(enabled != 0 ? 1 : 0, std::memory_order_relaxed);
    return before;
}

int ss_searcher_tuning_state(const ss_searcher *s, const void *d_haystack, size_t len, ss_tuning_state *out)
{
    if (!s || !out) return fail(SS_ERR_ARGUMENT, "NULL argument");
    memset(out, 0, sizeof *out);
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    out->autotune = autotune_enabled() ? 1u : 0u;
    out->own[0] = out->in_force[0] = (uint32_t)s->da;
    out->own[1] = out->in_force[1] = (uint32_t)s->db;
    out->own[2] = out->in_force[2] = (uint32_t)s->dc;
    out->workgroups_per_cu = (uint32_t)__atomic_load_n(&pd->last_occ, __ATOMIC_RELAXED);
    out->grid = __atomic_load_n(&pd->last_grid, __ATOMIC_RELAXED);
    out->kernel_mode = (uint32_t)__atomic_load_n(&pd->last_mode, __ATOMIC_RELAXED);
    out->last_found = (uint32_t)__atomic_load_n(&pd->last_found, __ATOMIC_RELAXED);
    while (__atomic_exchange_n(&pd->census_lock, 1u, __ATOMIC_ACQUIRE) != 0) cpu_relax();
    complete_pending(s, pd);                            // (reads pinned memory; launches nothing, waits for nothing)
    for (auto &c : pd->census) {
        if (c.state == 0 || c.hay != d_haystack || c.len != len || c.gen != s->filter_gen) continue;
        out->census_state = c.state;
        out->census_age = c.uses;
        if (c.state == 2) {
            const CensusCounts cc = census_counts(c.sums);
            out->tiles = ss::kCensusTiles;
            out->tiles3 = cc.tiles3;
            out->tiles2 = cc.tiles2;
            out->match_tiles = cc.match_tiles;
            out->lanes = cc.lanes;
        }
        if (c.stats_roles >= 0) {
            out->pair_lanes = c.pair_lanes;
            out->triple_lanes = c.triple_lanes;
            out->deep_lanes = c.deep_lanes;
        }
        out->triple_state = c.adopted ? 2u : (c.settled ? 1u : 0u);
        out->on_trial = c.inflight == 2 ? 1u : 0u;
        out->trials = c.trials;
        out->accepted = c.accepted;
        out->settled = c.settled ? 1u : 0u;
        out->proposal = c.prop_kind;
        {
            size_t tri[3];
            normalised(c.cur, tri);
            out->in_force[0] = (uint32_t)tri[0];
            out->in_force[1] = (uint32_t)tri[1];
            out->in_force[2] = (uint32_t)tri[2];
        }
        if (c.have_order) {
            out->order_measured = 1;
            out->norder = c.norder;
            for (uint32_t t = 0; t < c.norder && t < 16; ++t)
                out->order[t] = (uint8_t)(out->in_force[0] + (uint32_t)((c.order_idx[t >> 3] >> (8 * (t & 7))) & 0xFF));
        }
    }
    __atomic_store_n(&pd->census_lock, 0u, __ATOMIC_RELEASE);
    {
        DeviceStats &ds = device_stats()[pd->dev >= 0 && pd->dev < kMaxDevices ? pd->dev : 0];
        std::lock_guard<std::mutex> lock(ds.mu);
        for (auto &e : ds.e)
            if (e.state != 0 && e.hay == d_haystack && e.len == len) out->histogram_state = e.state;
    }
    return SS_OK;
}

#ifdef SS_TEST_HOOKS
// the census's per-position match counters of (searcher, haystack): pair_match[64] | triple_match[64] | pair lanes | triple lanes;
// *have = 0 when they are not in
int ss_debug_census_stats(const ss_searcher *s, const void *d_haystack, size_t len, uint32_t stats[131], int *have)
{
    if (!s || !stats || !have) return fail(SS_ERR_ARGUMENT, "NULL argument");
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    *have = 0;
    while (__atomic_exchange_n(&pd->census_lock, 1u, __ATOMIC_ACQUIRE) != 0) cpu_relax();
    complete_pending(s, pd);
    for (auto &c : pd->census) {
        if (c.state == 0 || c.hay != d_haystack || c.len != len || c.gen != s->filter_gen || c.stats_roles < 0) continue;
        for (int k = 0; k < 64; ++k) {
            stats[k] = c.pair_match[k];
            stats[64 + k] = c.triple_match[k];
        }
        stats[128] = c.pair_lanes;
        stats[129] = c.triple_lanes;
        stats[130] = c.deep_lanes;
        *have = 1 + c.stats_roles;                      // (1 + the slot the pair counts were gathered for)
    }
    __atomic_store_n(&pd->census_lock, 0u, __ATOMIC_RELEASE);
    return SS_OK;
}

int ss_debug_census(const ss_searcher *s, const void *d_haystack, size_t len, uint32_t counts[11])
{
    if (!s || !counts) return fail(SS_ERR_ARGUMENT, "NULL argument");
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    CensusCounts cc = {0, 0, 0, 0};
    counts[0] = 0;
    counts[6] = (uint32_t)s->da;
    counts[7] = (uint32_t)s->db;
    counts[8] = (uint32_t)s->dc;
    counts[9] = counts[10] = 0;
    while (__atomic_exchange_n(&pd->census_lock, 1u, __ATOMIC_ACQUIRE) != 0) cpu_relax();
    complete_pending(s, pd);
    for (auto &c : pd->census) {
        if (c.state != 0 && c.hay == d_haystack && c.len == len && c.gen == s->filter_gen) {
            if (c.state == 2) {
                cc = census_counts(c.sums);
                counts[0] = ss::kCensusTiles;
            }
            counts[9] = c.adopted ? 2u : (c.settled ? 1u : 0u);
            counts[10] = c.trials;
            {
                size_t tri[3];
                normalised(c.cur, tri);
                counts[6] = (uint32_t)tri[0];
                counts[7] = (uint32_t)tri[1];
                counts[8] = (uint32_t)tri[2];
            }
        }
    }
    __atomic_store_n(&pd->census_lock, 0u, __ATOMIC_RELEASE);
    counts[1] = cc.tiles3;
    counts[2] = cc.tiles2;
    counts[3] = cc.match_tiles;
    counts[4] = cc.lanes;
    counts[5] = (uint32_t)__atomic_load_n(&pd->last_mode, __ATOMIC_RELAXED);
    return SS_OK;
}
#endif

}  // extern "C"
