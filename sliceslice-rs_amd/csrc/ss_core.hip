// ss_core.hip - errors, device facts, the control-block pools, and ss_searcher itself: the counterpart of
//   ss_searcher_with_position   DynamicAvx2Searcher::with_position   /root/reference/src/x86.rs:468-493
//   ss_searcher_new             DynamicAvx2Searcher::new             src/x86.rs:454-459
// plus the filter-byte choice the reference leaves to its caller (`position`, src/x86.rs:252-255).  The searches are in
// ss_scan.hip (device haystacks), ss_host.hip, ss_batched.hip, ss_service.hip, ss_comm.hip.  There is no CPU search path.
#include "ss_internal.hpp"

#include <algorithm>

namespace ssh {

namespace {
thread_local char g_err[512] = "";
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

const char *last_error() { return g_err; }

int device_info(int dev, DeviceInfo *out)
{
    static std::mutex mu;
    static DeviceInfo cache[kMaxDevices];
    if (dev < 0 || dev >= kMaxDevices) return fail(SS_ERR_ARGUMENT, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(mu);
    if (!cache[dev].ok) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        cache[dev].cus = prop.multiProcessorCount;
        cache[dev].gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        cache[dev].hdp_flush = prop.hdpMemFlushCntl;
        int large = 0;
        if (hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, dev) != hipSuccess) { large = 0; (void)hipGetLastError(); }
        cache[dev].large_bar = large != 0;
        cache[dev].ok = true;
    }
    *out = cache[dev];
    return SS_OK;
}

std::atomic<uint64_t> g_upload_ticket{0};

// Thread-local HIP objects (timing events, the small-slice pinned buffer and its streams) are destroyed by their thread's
// exit.  A thread that outlives exit() - detached workers, threads still unwinding while the process shuts down - would
// call hipEventDestroy / hipHostFree into a runtime whose own static state may already be gone.  exit() runs the
// calling thread's thread_local destructors FIRST (glibc: __call_tls_dtors), then the atexit handlers in reverse order of
// registration; this library registers its handler after libamdhip64 (a dependency, loaded earlier) has registered its
// own, so the mark below is set before the runtime tears anything down, and destructors that run later leak instead.
namespace {
std::atomic<bool> g_exiting{false};
struct ExitMark {
    ExitMark() { (void)atexit([]() { g_exiting.store(true, std::memory_order_release); }); }
} g_exit_mark;
}  // namespace
bool process_exiting() { return g_exiting.load(std::memory_order_acquire); }

namespace {

// ---- control blocks ---------------------------------------------------------------------------------------------------
// A searcher needs, per device, 2.3 KB of device memory (flag / minimum / completion slots, the census's counters), 1.8 KB of pinned
// host memory (their mirrors) and its needle (up to 1.5 KB inside the block).  Allocated one by one that was five hipMalloc, three hipHostMalloc, four hipMemset and a
// hipMemcpy per `new` - the better part of a millisecond for a constructor that costs the reference tens of nanoseconds, and
// every one of those calls waits for the whole device (a resident search service: for its lease).  Blocks come from slabs
// instead (512 blocks of 4 KiB device + 2 KiB pinned memory per slab, kept until the process ends), and a block is
// initialised by the CPU THROUGH THE PCIe BAR (every byte of an MI300-class part's memory is CPU-visible): `new` makes no
// runtime call at all once a slab exists.  The writes are pushed through the device's host data path and waited for (bar_write);
// a kernel's start drops the caches' copy of the block; a resident service kernel acquires what was uploaded after its last
// look (upload tickets).  Without a large BAR (or with SLICESLICE_NO_BAR_WRITES=1) the image goes by one hipMemcpy.
constexpr size_t kBlockDevBytes = 4096, kBlockHostBytes = 2048, kBlockNeedleOff = 2560, kBlockNeedleMax = kBlockDevBytes - kBlockNeedleOff;
constexpr size_t kCensusStatBytes = 4 * (2 * 64 + 3);           // ss::kCensusStatWords counters (aux_kernels.hpp; checked in ss_census.hip)
constexpr size_t kOffFlags = 0, kOffBest = 256, kOffDone = 768, kOffBestDone = 1280, kOffCensus = 1792, kOffStats = 1808, kCtlBytes = 2336;
constexpr size_t kHostOffFlags = 0, kHostOffBest = 256, kHostOffDone = 768, kHostOffCensus = 1280, kHostOffStats = 1296;
static_assert(kOffStats + kCensusStatBytes <= kCtlBytes && kCtlBytes <= kBlockNeedleOff && kHostOffStats + kCensusStatBytes <= kBlockHostBytes,
              "the control words fit their halves of a block");
constexpr uint32_t kBlocksPerSlab = 512;        // 2 MiB of device memory per slab: one page-table fragment

struct BlockPool {
    std::mutex mu;
    std::vector<uint8_t *> d_slabs, h_slabs;
    std::vector<uint32_t> free_blocks;          // slab << 16 | index
};
// The pools are allocated on first use and NEVER destroyed: a searcher may be dropped by a thread that outlives main(), after
// the static destructors of this library have run (a function-local or namespace-scope array of pools would be gone by then).
BlockPool *pools()
{
    static BlockPool *const p = new BlockPool[kMaxDevices];
    return p;
}

}  // namespace
bool bar_writes_allowed()
{
    static const bool allowed = []() { const char *off = getenv("SLICESLICE_NO_BAR_WRITES"); return !(off && off[0] == '1'); }();
    return allowed;
}
namespace {

int pool_acquire(int dev, uint32_t *id, uint8_t **d, uint8_t **h, bool *bar, volatile uint32_t **hdp_flush)
{
    DeviceInfo di;
    if (int rc = device_info(dev, &di)) return rc;
    BlockPool &bp = pools()[dev];
    std::lock_guard<std::mutex> lock(bp.mu);
    if (bp.free_blocks.empty()) {
        uint8_t *ds = nullptr, *hs = nullptr;
        hipError_t e = hipMalloc((void **)&ds, kBlocksPerSlab * kBlockDevBytes);
        if (e == hipSuccess) e = hipHostMalloc((void **)&hs, kBlocksPerSlab * kBlockHostBytes, hipHostMallocDefault);
        if (e != hipSuccess) {
            (void)hipFree(ds);
            return fail(e == hipErrorOutOfMemory ? SS_ERR_NOMEM : SS_ERR_HIP, "control-block slab: %s", hipGetErrorString(e));
        }
        const uint32_t slab = (uint32_t)bp.d_slabs.size();
        bp.d_slabs.push_back(ds);
        bp.h_slabs.push_back(hs);
        for (uint32_t k = kBlocksPerSlab; k-- > 0;) bp.free_blocks.push_back(slab << 16 | k);
    }
    *id = bp.free_blocks.back();
    bp.free_blocks.pop_back();
    *d = bp.d_slabs[*id >> 16] + (size_t)(*id & 0xFFFF) * kBlockDevBytes;
    *h = bp.h_slabs[*id >> 16] + (size_t)(*id & 0xFFFF) * kBlockHostBytes;
    *bar = di.large_bar && bar_writes_allowed();
    *hdp_flush = di.hdp_flush;
    return SS_OK;
}

void pool_release(int dev, uint32_t id)
{
    BlockPool &bp = pools()[dev];
    std::lock_guard<std::mutex> lock(bp.mu);
    bp.free_blocks.push_back(id);
}

}  // namespace

// `bytes` (a multiple of 16) from host memory into device memory through the BAR, complete before anything the caller does
// next can reach the device: CPU writes into device memory pass through the device's host data path (HDP), which may hold
// them back; writing its flush register pushes them out, and reading the register back waits until that write - and with
// it, in order, everything in front of it - has arrived (what the HIP runtime does for kernel arguments it places in device
// memory).  One PCIe read round trip: the microsecond of a `new`'s two.
void bar_write(uint8_t *d_dst, const uint8_t *src, size_t bytes, volatile uint32_t *hdp_flush)
{
    for (size_t k = 0; k < bytes; k += 16)
        _mm_store_si128(reinterpret_cast<__m128i *>(d_dst + k), _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + k)));
    _mm_sfence();
    if (hdp_flush) {
        // (atomic accesses: several threads - building searchers, posting service requests - write this register concurrently;
        // any write to it means "flush")
        __atomic_store_n(hdp_flush, 1u, __ATOMIC_RELAXED);
        (void)__atomic_load_n(hdp_flush, __ATOMIC_RELAXED);
    } else {
        (void)*reinterpret_cast<volatile uint32_t *>(d_dst);    // no register at hand: read what was written first back
    }
}
int get_per_device(const ss_searcher *s, PerDevice **out)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(s->mu);
    for (auto &p : s->per)
        if (p.dev == dev) {
            *out = &p;
            return SS_OK;
        }
    DeviceInfo di;
    if (int rc = device_info(dev, &di)) return rc;
    if (!di.gfx950)   // the code objects in this library are gfx950 only; fail here, not at the first launch
        return fail(SS_ERR_NO_DEVICE, "HIP device %d is not a gfx950 (MI355X-class) device", dev);
    PerDevice p;
    p.dev = dev;
    uint8_t *db = nullptr, *hb = nullptr;
    bool bar = false;
    volatile uint32_t *hdp_flush = nullptr;
    if (int rc = pool_acquire(dev, &p.block, &db, &hb, &bar, &hdp_flush)) return rc;
    p.d_flags = reinterpret_cast<int *>(db + kOffFlags);
    p.d_best = reinterpret_cast<uint64_t *>(db + kOffBest);
    p.d_done = reinterpret_cast<unsigned long long *>(db + kOffDone);
    p.d_best_done = reinterpret_cast<uint64_t *>(db + kOffBestDone);
    p.h_flags = reinterpret_cast<int *>(hb + kHostOffFlags);
    p.h_best = reinterpret_cast<uint64_t *>(hb + kHostOffBest);
    p.h_done = reinterpret_cast<long long *>(hb + kHostOffDone);
    p.d_census = reinterpret_cast<unsigned long long *>(db + kOffCensus);
    p.h_census = reinterpret_cast<unsigned long long *>(hb + kHostOffCensus);
    p.d_stats = reinterpret_cast<uint32_t *>(db + kOffStats);
    p.h_stats = reinterpret_cast<uint32_t *>(hb + kHostOffStats);
    memset(hb, 0, kBlockHostBytes);                    // blocks are recycled: a stale value must not equal an epoch
    for (int k = 0; k < kSlots; ++k) p.find_tag[k] = kFindTagMax;
    // the block's image: flags 0 | minima all ones | completion counters 0 | keyed minima all ones | census words and counters 0 | the needle
    const bool inside = s->n <= kBlockNeedleMax;
    alignas(16) uint8_t img[kBlockDevBytes];
    memset(img, 0, sizeof img);
    memset(img + kOffBest, 0xFF, kOffDone - kOffBest);
    memset(img + kOffBestDone, 0xFF, kOffCensus - kOffBestDone);
    if (inside && s->n) memcpy(img + kBlockNeedleOff, s->needle.data(), s->n);
    const size_t img_bytes = inside ? kBlockNeedleOff + ((s->n + 15) & ~(size_t)15) : kCtlBytes;
    hipError_t e = hipSuccess;
    if (bar) bar_write(db, img, img_bytes, hdp_flush);
    else e = hipMemcpy(db, img, img_bytes, hipMemcpyHostToDevice);
    p.d_needle = db + kBlockNeedleOff;
    if (e == hipSuccess && !inside) {                   // a needle too long for the block
        p.d_needle = nullptr;
        if ((e = hipMalloc((void **)&p.d_needle, s->n)) == hipSuccess) {
            p.needle_own = true;
            e = hipMemcpy(p.d_needle, s->needle.data(), s->n, hipMemcpyHostToDevice);
        }
    }
    if (e != hipSuccess) {                             // nothing half-built is left behind
        if (p.needle_own) (void)hipFree(p.d_needle);
        pool_release(dev, p.block);
        return fail(e == hipErrorNoDevice ? SS_ERR_NO_DEVICE : (e == hipErrorOutOfMemory ? SS_ERR_NOMEM : SS_ERR_HIP),
                    "per-device setup: %s", hipGetErrorString(e));
    }
    p.upload_ticket = g_upload_ticket.fetch_add(1, std::memory_order_acq_rel) + 1;
    p.free_mask = ~0ull;
    s->per.push_back(p);
    *out = &s->per.back();
    return SS_OK;
}

int acquire_slot(const ss_searcher *s, PerDevice *p)
{
    std::unique_lock<std::mutex> lock(s->mu);
    s->slot_cv.wait(lock, [p]() { return p->free_mask != 0; });   // > 64 concurrent searches on one handle and device wait here
    const int k = __builtin_ctzll(p->free_mask);
    p->free_mask &= p->free_mask - 1;
    return k;
}

// The value that means "found" for the call that owns slot k: fresh per call, never 0.  On the (2^31
// calls) wrap-around both copies of the flag are cleared so that no stale value can equal a new epoch.
// Completion-word state of slot k back to its initial values (the caller owns the slot): after a failed launch, and
// before the counter's low half or the find() key could run out.
void start_over(PerDevice *p, int k)
{
    (void)hipDeviceSynchronize();
    (void)hipMemset(p->d_done + k, 0, sizeof(unsigned long long));
    (void)hipMemset(p->d_best_done + k, 0xFF, sizeof(uint64_t));
    p->h_done[k] = 0;
    p->done_low[k] = p->done_hi[k] = 0;
    p->find_tag[k] = kFindTagMax;
}

int next_epoch(PerDevice *p, int k)
{
    if (p->epoch[k] >= INT_MAX - 1 || p->epoch[k] < 0) {
        (void)hipDeviceSynchronize();
        (void)hipMemset(p->d_flags + k, 0, sizeof(int));
        p->h_flags[k] = 0;
        p->epoch[k] = 0;
        start_over(p, k);
    }
    return ++p->epoch[k];
}

void release_slot(const ss_searcher *s, PerDevice *p, int k)
{
    {
        std::lock_guard<std::mutex> lock(s->mu);
        p->free_mask |= 1ull << k;
    }
    s->slot_cv.notify_one();
}

namespace {

// ---- filter-byte choice for `new` callers -------------------------------------------------------------
// The reference tests needle[0] and needle[position] and leaves `position` to the caller, defaulting to the last
// byte (x86.rs:252-255, 285); the result never depends on it (lib.rs:375-378).  On the GPU the bytes decide how
// often the second phase runs (text passes a {' ', ' '} filter at percent rates) and, through the distance
// between the first two, which kernel runs (a distance >= 16 needs cross-lane traffic or a second load stream).
// For `new` callers the library therefore picks all of them (choose_filter_triple below): a first byte among the
// first kFilterWindow needle bytes and the two cheapest of the 15 bytes behind it, cheapest sum first; the cost of
// a byte is a static, corpus-free rarity class (ss::byte_rarity_rank; bytes outside text are all "rare" alike) or,
// on request, the log of its count in a histogram of the haystack.  with_position callers keep their byte: with
// position < 16 the pair is the reference's (0, position) plus the cheapest other byte of needle[1..15]; a farther
// position gets a partner close in front of it (choose_anchor).  ss_searcher_set_filter3 sets any pair verbatim.
constexpr size_t kFilterWindow = 1024;

inline int rarity_class(uint8_t b)
{
    const int r = ss::byte_rarity_rank(b);
    return r < 64 ? 0 : r;          // everything that is not text-like counts as equally rare
}

// Cost of one filter byte: the static class above, or - with a byte histogram of (a sample of) the haystack -
// 8 * log2(count + 1): summing costs then compares PRODUCTS of frequencies, which is what the candidate rate of
// a multi-byte filter is (bytes taken as independent).
}  // namespace
ByteCost::ByteCost(const uint64_t *hist)
{
    for (int b = 0; b < 256; ++b) {
        if (!hist) {
            cost[b] = rarity_class((uint8_t)b);
        } else {
            const uint64_t c = hist[b] + 1;
            const int lg = 63 - __builtin_clzll(c);                      // floor(log2 c)
            const int frac = lg >= 3 ? (int)((c >> (lg - 3)) & 7) : (int)((c << (3 - lg)) & 7);
            cost[b] = 8 * lg + frac;                                     // ~8 * log2(c), 0 .. 511
        }
    }
}

// Third byte for a given pair: the rarest byte among needle[fa+1 .. fa+15] other than the indices fb and `other`; ties to the
// later byte.  Returns fb when there is none.
size_t choose_third(const uint8_t *needle, size_t n, size_t fa, size_t fb, const ByteCost &cost, size_t other)
{
    if (n < 3 || fb < fa) return fb;
    size_t best = fb;
    int bc = INT_MAX;
    for (size_t k = fa + 1; k < n && k <= fa + 15; ++k) {
        if (k == fb || k == other) continue;
        const int c = cost(needle[k]);
        if (c <= bc) {
            bc = c;
            best = k;
        }
    }
    return best;
}
namespace {

// (fa, fb, fc): the first byte plus the two rarest bytes of the 15 that follow it, for the first byte that makes
// that sum smallest.  Ties: the reference's first byte (0) when it is among the best, else the earliest; among
// equally rare followers the later ones (for needles of <= 16 bytes of equal rarity that is the reference's
// pair (0, n-1) plus n-2).  fb > fc is not required; fb is the rarer (or later) of the two.
void choose_filter_triple(const uint8_t *needle, size_t n, size_t *fa, size_t *fb, size_t *fc, const ByteCost &cost)
{
    *fa = *fb = *fc = 0;
    if (n < 2) return;
    const size_t w = n < kFilterWindow ? n : kFilterWindow;
    int best = INT_MAX;
    size_t ba = 0, bb = 1, bc = 1;
    for (size_t a = 0; a + 1 < w; ++a) {
        const int ca = cost(needle[a]);
        if (ca > best) continue;
        // two smallest costs among a+1 .. a+15 (later index wins ties)
        int c1 = INT_MAX, c2 = INT_MAX;
        size_t i1 = a + 1, i2 = a + 1;
        for (size_t b = a + 1; b < w && b <= a + 15; ++b) {
            const int c = cost(needle[b]);
            if (c <= c1) {
                c2 = c1; i2 = i1;
                c1 = c; i1 = b;
            } else if (c <= c2) {
                c2 = c; i2 = b;
            }
        }
        const bool has2 = c2 != INT_MAX;
        const int total = ca + c1 + (has2 ? c2 : 512);         // no third byte to offer: worse than the most common one
        if (total < best) {
            best = total;
            ba = a;
            bb = i1;
            bc = has2 ? i2 : i1;
        }
    }
    *fa = ba;
    *fb = bb;
    *fc = bc;
}

// with_position callers whose byte lies 16 or more behind needle[0]: the caller's byte stays a first-phase byte, but its
// PARTNER becomes a byte at most 15 in front of it instead of needle[0] (the result does not depend on which bytes are
// tested, lib.rs:375-378) - one 16-byte load then covers both, so the search runs in the single-stream kernel rather than
// the cross-lane (distance < 1,008) or two-stream one.  (fa, fc): the cheapest anchor of needle[position-15 .. position-1]
// together with its cheapest third byte of needle[fa+1 .. fa+15] other than `position`; ties to the later anchor.
void choose_anchor(const uint8_t *needle, size_t n, size_t position, size_t *fa, size_t *fc, const ByteCost &cost)
{
    int best = INT_MAX;
    size_t ba = position - 1, bc = position;
    for (size_t a = position - 15; a < position; ++a) {
        int c3 = 512;
        size_t i3 = position;
        for (size_t k = a + 1; k < n && k <= a + 15; ++k) {
            if (k == position) continue;
            const int c = cost(needle[k]);
            if (c <= c3) {
                c3 = c;
                i3 = k;
            }
        }
        const int total = cost(needle[a]) + c3;
        if (total <= best) {
            best = total;
            ba = a;
            bc = i3;
        }
    }
    *fa = ba;
    *fc = bc;
}

// with_position: the caller's byte plus the reference's partner needle[0] and one more byte when position < 16, else
// choose_anchor's partner.
void filter_for_position(const uint8_t *needle, size_t n, size_t position, size_t *fa, size_t *fb, size_t *fc, const ByteCost &cost)
{
    *fa = 0;
    *fb = *fc = n >= 2 ? position : 0;
    if (n < 2) return;
    if (position >= 16) choose_anchor(needle, n, position, fa, fc, cost);
    else *fc = choose_third(needle, n, *fa, *fb, cost);
}

// What the device tests for the triple (fa, fb, fc).  A pair up to 16 * kShiftMaxD + 15 bytes apart has a kernel (single-stream
// up to 15, cross-lane beyond): the triple as it is.  Farther apart (only ss_searcher_set_filter3 can ask for that, e.g. the
// reference's own pair (0, n-1) for a needle of kilobytes) there is none - the two-stream kernel of rounds 1-3 ran at 0.83 of the
// roofline with 6 % re-read traffic - so the device filters with needle[fa] and the two rarest of the 15 bytes behind it, like a
// constructor-built searcher, and the caller's byte needle[fb] is the first thing a surviving candidate is tested for in memory
// (`far`; scan_filters.hpp verify_flags).  Still tested before any compare, still a necessary condition: results cannot change
// (/root/reference/src/lib.rs:375-378).
// A pair WITHOUT a third byte (third == second: what ss_searcher_set_filter3 takes for "a plain two-byte filter", e.g. the
// reference's own pair (0, n-1)) still gets one on the device - the rarest of the 15 bytes behind the first - because every
// kernel's first phase tests three bytes anyway (a missing third is the second one tested twice) and on text a two-byte filter of
// common bytes sends nearly every tile into the second level: the reference's pair ran at 0.79 of the roofline on the i386 text,
// most of it spent there.  One more necessary condition; ss_searcher_filter3 keeps reporting what the caller set.
void derive_device_filter(ss_searcher *s)
{
    s->da = s->fa;
    s->db = s->fb;
    s->dc = s->fc;
    s->far = 0;
    if (s->n < 3) return;
    const ByteCost cost(nullptr);
    if ((s->fb - s->fa) / 16 > kShiftMaxD) {
        s->far = s->fb;
        s->db = choose_third(s->needle.data(), s->n, s->fa, s->fa, cost);              // the rarest of needle[fa+1 .. fa+15]
        s->dc = choose_third(s->needle.data(), s->n, s->fa, s->db, cost);              // ... and the next rarest
    } else if (s->fc == s->fb || s->fb - s->fa > 15) {
        const size_t third = choose_third(s->needle.data(), s->n, s->fa, s->fb, cost);
        s->dc = third > s->fa && third != s->fb ? third : s->fb;
    }
}

// The setters' side of ss_searcher::gate: the triple is rewritten only while no search holds the searcher.
int store_filter(ss_searcher *s, size_t fa, size_t fb, size_t fc)
{
    int idle = 0;
    if (!s->gate.compare_exchange_strong(idle, -1, std::memory_order_acq_rel))
        return fail(SS_ERR_ARGUMENT, "%d search(es) in flight on this searcher: the filter bytes cannot be changed now", idle);
    s->fa = fa;
    s->fb = fb;
    s->fc = fc;
    s->auto_filter = false;         // the caller chose
    s->third_owned = fc == fb;      // ... a plain pair: the third byte stays the library's (derive_device_filter)
    s->anchor_owned = false;
    ++s->filter_gen;
    derive_device_filter(s);
    s->gate.store(0, std::memory_order_release);
    return SS_OK;
}


int make_searcher(const uint8_t *needle, size_t n, size_t position, bool auto_filter, ss_searcher **out)
{
    if (!out) return fail(SS_ERR_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (n && !needle) return fail(SS_ERR_ARGUMENT, "needle is NULL");
    // x86.rs:468-493: [] -> N0 (position ignored); [c0] -> assert_eq!(position, 0); else position < n.
    if (n == 1 && position != 0) return fail(SS_ERR_POSITION, "position must be 0 for a one-byte needle");
    if (n >= 2 && position >= n) return fail(SS_ERR_POSITION, "position %zu out of range for needle of %zu bytes", position, n);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(SS_ERR_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    ss_searcher *s = new (std::nothrow) ss_searcher;
    if (!s) return fail(SS_ERR_NOMEM, "out of memory");
    static std::atomic<uint64_t> next_uid{1};
    s->uid = next_uid.fetch_add(1, std::memory_order_relaxed);
    s->needle.assign(needle, needle + n);
    s->n = n;
    s->position = position;
    const ByteCost cost(nullptr);
    if (auto_filter) choose_filter_triple(s->needle.data(), n, &s->fa, &s->fb, &s->fc, cost);
    else filter_for_position(s->needle.data(), n, position, &s->fa, &s->fb, &s->fc, cost);
    derive_device_filter(s);
    s->auto_filter = auto_filter;
    s->third_owned = true;
    s->anchor_owned = !auto_filter && n >= 2 && position >= 16;
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) {      // uploads the needle to the current device now
        delete s;
        return rc;
    }
    *out = s;
    return SS_OK;
}

}  // namespace

}  // namespace ssh

using namespace ssh;

extern "C" {

const char *ss_last_error(void) { return last_error(); }

int ss_searcher_with_position(const uint8_t *needle, size_t n, size_t position, ss_searcher **out)
{
    return make_searcher(needle, n, position, false, out);
}

int ss_searcher_new(const uint8_t *needle, size_t n, ss_searcher **out)
{
    // x86.rs:457: position = n.wrapping_sub(1) - what ss_searcher_info keeps reporting.  The filter bytes
    // the device tests are chosen by choose_filter_triple.
    return make_searcher(needle, n, n - 1, true, out);
}

int ss_searcher_set_filter3(ss_searcher *s, size_t first, size_t second, size_t third)
{
    if (!s) return fail(SS_ERR_ARGUMENT, "searcher is NULL");
    if (s->n < 2) {
        if (first != 0 || second != 0 || third != 0) return fail(SS_ERR_POSITION, "needles shorter than two bytes have no filter pair");
        return SS_OK;
    }
    if (first > second || second >= s->n) return fail(SS_ERR_POSITION, "filter pair (%zu, %zu) out of range for a needle of %zu bytes", first, second, s->n);
    if (third == second) return store_filter(s, first, second, second);      // a plain two-byte filter
    if (second - first > 15 || third <= first || third - first > 15 || third >= s->n)
        return fail(SS_ERR_POSITION, "third filter byte %zu must lie within 15 bytes behind the first (%zu), as must the second (%zu)", third, first, second);
    return store_filter(s, first, second, third);
}

int ss_searcher_filter3(const ss_searcher *s, size_t *first, size_t *second, size_t *third)
{
    if (!s || !first || !second || !third) return fail(SS_ERR_ARGUMENT, "NULL argument");
    *first = s->fa;
    *second = s->fb;
    *third = (s->n >= 2 && s->fb - s->fa <= 15) ? s->fc : s->fb;
    return SS_OK;
}

int ss_choose_filter_triple(const uint8_t *needle, size_t n, const uint64_t hist[256], size_t *first, size_t *second, size_t *third)
{
    if (!first || !second || !third || (n && !needle)) return fail(SS_ERR_ARGUMENT, "NULL argument");
    choose_filter_triple(needle, n, first, second, third, ByteCost(hist));
    return SS_OK;
}

int ss_choose_position(const uint8_t *needle, size_t n, const uint64_t hist[256], size_t *position)
{
    if (!position || (n && !needle)) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (n <= 1) {
        *position = 0;              // with_position demands 0 for one-byte needles (x86.rs:473)
        return SS_OK;
    }
    // rarest byte among needle[1..n); ties go to the later byte (further from the first-byte filter).
    // Without a histogram: the reference's default, the last byte (x86.rs:285).
    size_t best = n - 1;
    if (hist) {
        for (size_t k = n - 1; k >= 1; --k)
            if (hist[needle[k]] < hist[needle[best]]) best = k;
    }
    *position = best;
    return SS_OK;
}

void ss_searcher_free(ss_searcher *s)
{
    if (!s) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &p : s->per) {
        // A block goes back to its pool and may be handed out - and rewritten - at once: nothing of this searcher may still be
        // running.  The synchronous entry points have returned with their kernels' last stores made; only the *_async ones
        // leave work behind, and a searcher that used them waits for its device here (hipFree did, implicitly, for all).
        if (s->used_async.load(std::memory_order_acquire) || p.needle_own) {
            (void)hipSetDevice(p.dev);
            if (s->used_async.load(std::memory_order_acquire)) (void)hipDeviceSynchronize();
            if (p.needle_own) (void)hipFree(p.d_needle);
        }
        pool_release(p.dev, p.block);              // (the pools are never destroyed: safe even after exit() has begun)
    }
    (void)hipSetDevice(cur);
    timer_forget(s);
    delete s;
}

int ss_searcher_info(const ss_searcher *s, size_t *needle_len, size_t *position)
{
    if (!s) return fail(SS_ERR_ARGUMENT, "searcher is NULL");
    if (needle_len) *needle_len = s->n;
    if (position) *position = s->position;
    return SS_OK;
}

int ss_searcher_set_timing(ss_searcher *s, int enabled)
{
    if (!s) return fail(SS_ERR_ARGUMENT, "searcher is NULL");
    s->timing = enabled != 0;
    return SS_OK;
}

int ss_device_info(char *name, size_t name_cap, int *compute_units, size_t *total_mem)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (name && name_cap) snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (total_mem) *total_mem = prop.totalGlobalMem;
    return SS_OK;
}

#ifdef SS_TEST_HOOKS
// ---- tuning knobs and test hooks (sliceslice_hip_tuning.h; not compiled into the product library) ------------------------
const char *ss_version(void)
{
#ifdef SS_TUNING_VARIANTS
    return "sliceslice-hip 0.4 (gfx950, tuning build: every kernel variant, test hooks)";
#else
    return "sliceslice-hip 0.4 (gfx950, test hooks)";
#endif
}

int ss_choose_filter_for_position(const uint8_t *needle, size_t n, size_t position, size_t *first, size_t *second, size_t *third)
{
    if (!first || !second || !third || (n && !needle)) return fail(SS_ERR_ARGUMENT, "NULL argument");
    *first = *second = *third = 0;
    if (n == 1 && position != 0) return fail(SS_ERR_POSITION, "position must be 0 for a one-byte needle");
    if (n >= 2 && position >= n) return fail(SS_ERR_POSITION, "position %zu out of range for needle of %zu bytes", position, n);
    filter_for_position(needle, n, position, first, second, third, ByteCost(nullptr));
    return SS_OK;
}

// move the epoch counters close to the 2^31 wrap so that tests can cross it
int ss_debug_set_epochs(ss_searcher *s, int value)
{
    if (!s) return fail(SS_ERR_ARGUMENT, "searcher is NULL");
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    std::lock_guard<std::mutex> lock(s->mu);
    if (pd->free_mask != ~0ull) return fail(SS_ERR_ARGUMENT, "searches in flight");
    for (int k = 0; k < kSlots; ++k) pd->epoch[k] = value;
    return SS_OK;
}

int ss_debug_set_completion_state(ss_searcher *s, uint32_t workgroups, uint32_t found_workgroups, uint32_t find_key)
{
    if (!s) return fail(SS_ERR_ARGUMENT, "searcher is NULL");
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    std::lock_guard<std::mutex> lock(s->mu);
    if (pd->free_mask != ~0ull) return fail(SS_ERR_ARGUMENT, "searches in flight");
    if (find_key > kFindTagMax) return fail(SS_ERR_ARGUMENT, "find key above %u", kFindTagMax);
    std::vector<unsigned long long> counters(kSlots);
    for (int k = 0; k < kSlots; ++k) {
        pd->done_low[k] = workgroups;
        pd->done_hi[k] = found_workgroups;
        pd->find_tag[k] = find_key;
        counters[k] = ((unsigned long long)found_workgroups << 32) | workgroups;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(pd->d_done, counters.data(), kSlots * sizeof(unsigned long long), hipMemcpyHostToDevice));
    return SS_OK;
}

int ss_debug_fail_next_scans(ss_searcher *s, int count)
{
    if (!s || count < 0) return fail(SS_ERR_ARGUMENT, "bad argument");
    s->debug_fail_scans.store(count, std::memory_order_relaxed);
    return SS_OK;
}

int ss_searcher_set_variant(ss_searcher *s, int variant)
{
    if (!s) return fail(SS_ERR_ARGUMENT, "searcher is NULL");
    s->variant = variant;
    return SS_OK;
}

int ss_searcher_set_grid(ss_searcher *s, int blocks)
{
    if (!s) return fail(SS_ERR_ARGUMENT, "searcher is NULL");
    s->grid = blocks;
    return SS_OK;
}
#endif  // SS_TEST_HOOKS

}  // extern "C"
