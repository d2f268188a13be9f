// ss_census.hip - what a searcher learns about a haystack by ASKING it (launch tuning; no search result depends on any of this):
//   * the candidate census (census_kernel): how often the searcher's filter bytes fire on this haystack -> four or six workgroups
//     per CU, the cross-lane kernels with or without their third byte;
//   * the byte histogram of the same sample (hist_sample_kernel), kept per DEVICE and shared by every searcher that meets the
//     haystack -> for searchers built by ss_searcher_new, whose caller did not choose, the three filter bytes themselves: row f3
//     of SURVEY.md 8f ("pick the needle bytes with the lowest corpus frequency"; the reference leaves `position` to its caller,
//     /root/reference/src/x86.rs:252-255) without the caller having to ask for a histogram.
// Both kernels are enqueued in front of a scan on the scan's own stream and are never waited for by the host: the first scan of
// a (searcher, haystack) pair goes by static guesses, later ones by what has arrived.  There is no CPU search path in this file.
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
constexpr uint32_t kCensusDenseTiles = 48, kCensusDenseLanes = 256;
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

struct CensusCounts {
    uint32_t tiles3, tiles2, match_tiles, lanes;
};
CensusCounts census_counts(uint64_t sums)
{
    return {(uint32_t)(sums & 2047u), (uint32_t)((sums >> 11) & 2047u), (uint32_t)((sums >> 22) & 2047u), (uint32_t)(sums >> 33)};
}

// The census in flight, if its counts have arrived.  A triple chosen from the histogram is on TRIAL until its first counts are in:
// the histogram prices bytes as independent, the census counts what really happens - if the new triple meets candidates in MORE
// tiles than the searcher's own did (bytes that come in runs), the searcher's own stays.
void complete_pending(PerDevice *pd)
{
    if (pd->census_pending < 0) return;
    PerDevice::Census &c = pd->census[pd->census_pending];
    if (__atomic_load_n(pd->h_census + 1, __ATOMIC_ACQUIRE) != (unsigned long long)c.tag) return;
    uint64_t sums = __atomic_load_n(pd->h_census, __ATOMIC_RELAXED);
    if (c.triple_state == 2 && c.trial) {
        c.trial = false;
        if (census_counts(sums).tiles3 > census_counts(c.sums_own).tiles3) {
            c.triple_state = 1;
            sums = c.sums_own;
        }
    }
    c.sums = sums;
    c.state = 2;
    pd->census_pending = -1;
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

}  // namespace

void launch_hints(const ss_searcher *s, PerDevice *pd, const void *d_hay, size_t len, hipStream_t st, LaunchHints *out)
{
    out->have_counts = false;
    out->have_triple = false;
    out->workgroups_per_cu = 0;
    out->sparse_pair = false;
    if (len < kCensusMinBytes || s->n < 2 || len < s->n) return;
    if (__atomic_exchange_n(&pd->census_lock, 1u, __ATOMIC_ACQUIRE) != 0) return;         // another thread is at it
    struct Unlock {
        uint32_t *w;
        ~Unlock() { __atomic_store_n(w, 0u, __ATOMIC_RELEASE); }
    } unlock{&pd->census_lock};
    complete_pending(pd);
    PerDevice::Census *hit = nullptr, *victim = &pd->census[0];
    for (auto &c : pd->census) {
        if (c.state != 0 && c.hay == d_hay && c.len == len && c.gen == s->filter_gen) hit = &c;
        if (c.state != 1 && (victim->state == 1 || c.stamp < victim->stamp)) victim = &c;
    }
    // the filter bytes themselves (searchers built by ss_searcher_new only): decided once per (searcher, haystack), when the
    // haystack's histogram is in
    bool recount = false;
    if (s->auto_filter && s->n >= 3 && (!hit || hit->triple_state == 0)) {       // (decided once per census entry)
        uint64_t hist[256];
        const bool have_hist = stats_lookup(pd->dev, d_hay, len, st, hist);
        if (have_hist && hit && hit->triple_state == 0 && hit->state == 2) {
            size_t tri[3];
            if (better_triple(s, hist, tri)) {
                hit->triple_state = 2;
                hit->trial = true;
                ++hit->trials;
                hit->sums_own = hit->sums;
                hit->tri[0] = tri[0];
                hit->tri[1] = tri[1];
                hit->tri[2] = tri[2];
                recount = true;                         // the counts at hand describe the old triple
            } else {
                hit->triple_state = 1;
            }
        }
    }
    PerDevice::Census *target = victim;
    if (hit) {
        hit->stamp = ++pd->census_clock;
        if (hit->triple_state == 2) {
            out->have_triple = true;
            out->tri[0] = hit->tri[0];
            out->tri[1] = hit->tri[1];
            out->tri[2] = hit->tri[2];
        }
        if (hit->state == 2 && !recount) {
            const CensusCounts cc = census_counts(hit->sums);
            out->have_counts = true;
            out->workgroups_per_cu = cc.match_tiles != 0 ? 4 : (cc.tiles3 >= kCensusDenseTiles || cc.lanes >= kCensusDenseLanes ? 6 : 4);
            out->sparse_pair = cc.tiles2 <= kCensusSparsePairTiles;
            // A buffer may be refilled in place: the counts are taken again every kCensusRefreshEvery scans (the old ones serve
            // until the new ones are in).
            if (++hit->uses % kCensusRefreshEvery != 0) return;
        } else if (hit->state == 1) {
            return;                                     // its census is in flight
        }
        target = hit;
    }
    if (pd->census_pending >= 0 || (!hit && victim->state == 1)) return;                  // one census in flight per searcher and device
    const size_t n = s->n, end = len - n + 1;
    if (end < 2 * (size_t)ss::kCensusTileBytes + 8) return;
    const uint64_t room = end - 4 - ss::kCensusTileBytes;                                 // latest start of a sampled tile
    const uint64_t stride = (room / (ss::kCensusTiles - 1)) & ~(uint64_t)(ss::kCensusTileBytes - 1);
    if (stride < ss::kCensusTileBytes || stream_is_capturing(st)) return;
    const bool alt = hit && hit->triple_state == 2;
    const size_t oa = alt ? hit->tri[0] : s->da, ob = alt ? hit->tri[1] : s->db, oc = alt ? hit->tri[2] : s->dc;
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
    a.h_out = pd->h_census;
    ss::census_kernel<<<dim3(a.nblocks), dim3(ss::kBlock), 0, st>>>(a);
    if (hipGetLastError() != hipSuccess) return;
    if (!hit) {
        target->hay = d_hay;
        target->len = len;
        target->gen = s->filter_gen;
        target->sums = 0;
        target->uses = 0;
        target->triple_state = 0;
        target->trial = false;
        target->trials = 0;
        target->stamp = ++pd->census_clock;
    }
    if (!hit || recount) {
        target->state = 1;                              // no counts (for this triple) yet
        out->have_counts = false;
    }
    target->tag = a.tag;
    pd->census_pending = (int)(target - pd->census);
}

}  // namespace ssh

using namespace ssh;

extern "C" {

#ifdef SS_TEST_HOOKS
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
    complete_pending(pd);
    for (auto &c : pd->census) {
        if (c.state != 0 && c.hay == d_haystack && c.len == len && c.gen == s->filter_gen) {
            if (c.state == 2) {
                cc = census_counts(c.sums);
                counts[0] = ss::kCensusTiles;
            }
            counts[9] = c.triple_state;
            counts[10] = c.trials;
            if (c.triple_state == 2) {
                counts[6] = (uint32_t)c.tri[0];
                counts[7] = (uint32_t)c.tri[1];
                counts[8] = (uint32_t)c.tri[2];
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
