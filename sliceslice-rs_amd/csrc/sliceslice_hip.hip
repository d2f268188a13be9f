// sliceslice_hip.hip - host side of the C ABI declared in include/sliceslice_hip.h.
//
// Mirrors, for the GPU, what the reference does on the CPU:
//   ss_searcher_with_position   DynamicAvx2Searcher::with_position   /root/reference/src/x86.rs:468-493
//   ss_searcher_new             DynamicAvx2Searcher::new             src/x86.rs:454-459
//   ss_search_*                 DynamicAvx2Searcher::search_in       src/x86.rs:498-525
//                               (N0 -> true x86.rs:500; N1 = MemchrSearcher lib.rs:130-136;
//                                len < n -> false / len == n -> equality x86.rs:357-359)
// The scan itself lives in scan_kernels.hpp.  There is no CPU search path in this file.
#include <hip/hip_runtime.h>
#include <emmintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sliceslice_hip.h"
#define SS_MISC_KERNELS 1
#include "scan_kernels.hpp"
#include "scan_launch.hpp"

namespace {

thread_local char g_err[512] = "";

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

// restores the calling thread's current device on scope exit
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorNoDevice ? SS_ERR_NO_DEVICE : SS_ERR_HIP, "%s: %s (%s:%d)",  \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

struct DeviceInfo {
    int cus = 0;
    bool ok = false;
    bool gfx950 = false;
};

int device_info(int dev, DeviceInfo *out)
{
    static std::mutex mu;
    static std::vector<DeviceInfo> cache;
    std::lock_guard<std::mutex> lock(mu);
    if ((int)cache.size() <= dev) cache.resize(dev + 1);
    if (!cache[dev].ok) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        cache[dev].cus = prop.multiProcessorCount;
        cache[dev].gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        cache[dev].ok = true;
    }
    *out = cache[dev];
    return SS_OK;
}

// Per-device state of a searcher: the needle copy and a small pool of found-flag slots so that
// concurrent ss_search_device calls on one handle never share mutable scratch.
struct PerDevice {
    int dev = -1;
    uint8_t *d_needle = nullptr;
    int *d_flags = nullptr;     // kSlots ints; "found" is the owning call's epoch, so slots are never cleared
    int *h_flags = nullptr;     // pinned-host mirror written by the finding wave (no D2H copy per call)
    uint64_t *d_best = nullptr; // kSlots uint64 for find(): all-ones whenever a slot is free
    uint64_t *h_best = nullptr; // pinned mirror
    // Completion word (small grids).  Nothing on the device side is ever reset between calls: the counter's low half
    // counts workgroups out towards a target the host names per launch, its high half counts the workgroups that found
    // the needle (the host remembers where it stood), and find() keys its minimum with a per-launch tag that decreases.
    // The host copies below belong to whoever owns the slot; start_over() resets a slot behind a device synchronise.
    unsigned long long *d_done = nullptr;   // kSlots counters: found-workgroups << 32 | workgroups
    long long *h_done = nullptr;            // pinned: the answer word, stored by the workgroup that completes the count
    uint64_t *d_best_done = nullptr;        // kSlots keyed minima of find()
    uint32_t done_low[64] = {0}, done_hi[64] = {0};
    uint32_t find_tag[64] = {0};            // next key of the slot; counts down from kFindTagMax
    uint64_t free_mask = 0;
    int epoch[64] = {0};        // per slot: the "found" value of the slot's latest call
    uint64_t upload_ticket = 0; // g_upload_ticket when d_needle had been written (a resident service acquires what is newer)
    uint32_t block = ~0u;       // control block of the device's pool (BlockPool): everything above points into it ...
    bool needle_own = false;    // ... except a needle too long for the block, which has an allocation of its own
};
std::atomic<uint64_t> g_upload_ticket{0};
constexpr int kSlots = 64;
constexpr uint32_t kFindTagMax = (1u << (64 - ss::kFindOffsetBits)) - 2;   // keys tag << kFindOffsetBits stay below all-ones
constexpr uint32_t kDoneLowMax = 0x7FFF0000u;                               // start over before the low half could carry
constexpr int kMaxDevices = 64;

// Kernel timing (ss_searcher_set_timing): the hipEvent pair that brackets a scan belongs to the CALLING
// THREAD (one pair per thread and device, created on first use), so concurrent calls on one handle never
// share events; ss_searcher_last_kernel_ms reports the calling thread's most recent timed scan.
// Thread-local HIP objects (timing events, the small-slice pinned buffer and its streams) are destroyed by their thread's
// exit.  A thread that outlives exit() - detached workers, threads still unwinding while the process shuts down - would
// call hipEventDestroy / hipHostFree into a runtime whose own static state may already be gone.  exit() runs the
// calling thread's thread_local destructors FIRST (glibc: __call_tls_dtors), then the atexit handlers in reverse order of
// registration; this library registers its handler after libamdhip64 (a dependency, loaded earlier) has registered its
// own, so the mark below is set before the runtime tears anything down, and destructors that run later leak instead.
std::atomic<bool> g_exiting{false};
struct ExitMark {
    ExitMark() { (void)atexit([]() { g_exiting.store(true, std::memory_order_release); }); }
} g_exit_mark;
inline bool process_exiting() { return g_exiting.load(std::memory_order_acquire); }

struct ThreadTimer {
    hipEvent_t ev0[kMaxDevices] = {nullptr}, ev1[kMaxDevices] = {nullptr};
    const void *owner = nullptr;      // the searcher of the most recent timed scan
    int dev = -1;
    ~ThreadTimer()
    {
        if (process_exiting()) return;                  // leak: see ExitMark
        for (int d = 0; d < kMaxDevices; ++d) {
            if (ev0[d]) (void)hipEventDestroy(ev0[d]);
            if (ev1[d]) (void)hipEventDestroy(ev1[d]);
        }
    }
};
thread_local ThreadTimer g_timer;

}  // namespace

struct ss_searcher {
    std::vector<uint8_t> needle;
    size_t n = 0;
    size_t position = 0;      // the API position (what ss_searcher_position reports; x86.rs:468)
    // The two needle bytes the device filter actually tests: needle[fa] and needle[fb], fa < fb (fa == fb == 0
    // for one-byte needles and for with_position(.., 0)).  with_position callers get (0, position) - the
    // reference's filter; `new` callers get a pair chosen by choose_filter_pair.  The result of a search never
    // depends on the pair (src/lib.rs:375-378 asserts that for every position).
    size_t fa = 0, fb = 0;
    // Third byte of the first-phase filter, fa < fc <= fa + 15 (== fb: none).  Only used when fb - fa <= 15 (the
    // single-stream kernels); an extra test, so it cannot change a result either.
    size_t fc = 0;
    int variant = 0;
    int grid = 0;
    bool timing = false;
    // Searches in flight (>= 0), or -1 while ss_searcher_set_filter* rewrites fa / fb / fc: the setters refuse
    // (SS_ERR_ARGUMENT) while a search runs instead of letting it read a half-written triple.
    mutable std::atomic<int> gate{0};
    mutable std::atomic<int> debug_fail_scans{0};       // test hook: the next k enqueue_scan calls fail (ss_debug_fail_next_scans)
    mutable std::atomic<bool> used_async{false};        // an *_async entry point may have left work behind (ss_searcher_free waits)
    mutable std::mutex mu;
    mutable std::condition_variable slot_cv;    // signalled when a flag slot is released
    mutable std::deque<PerDevice> per;          // deque: PerDevice pointers handed out stay valid as devices are added
};

namespace {

// One search's hold on the searcher's filter bytes (see ss_searcher::gate).  Counting, so entry points may nest.
struct SearchGate {
    const ss_searcher *s;
    explicit SearchGate(const ss_searcher *s_) : s(s_)
    {
        for (;;) {
            int v = s->gate.load(std::memory_order_acquire);
            if (v >= 0 && s->gate.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) return;
            if (v < 0) std::this_thread::yield();       // a setter is writing: a handful of stores
        }
    }
    ~SearchGate() { s->gate.fetch_sub(1, std::memory_order_acq_rel); }
    SearchGate(const SearchGate &) = delete;
    SearchGate &operator=(const SearchGate &) = delete;
};

// "One search at a time" scratch owners (communicators, communicator sets): the second concurrent call is refused.
struct BusyGuard {
    std::atomic<bool> *flag;
    bool mine;
    explicit BusyGuard(std::atomic<bool> *f) : flag(f), mine(!f->exchange(true, std::memory_order_acq_rel)) {}
    ~BusyGuard() { if (mine) flag->store(false, std::memory_order_release); }
    BusyGuard(const BusyGuard &) = delete;
    BusyGuard &operator=(const BusyGuard &) = delete;
};

// ---- control blocks ---------------------------------------------------------------------------------------------------
// A searcher needs, per device, 1.8 KB of device memory (flag / minimum / completion slots), 1.3 KB of pinned host memory
// (their mirrors) and its needle.  Allocated one by one that was five hipMalloc, three hipHostMalloc, four hipMemset and a
// hipMemcpy per `new` - the better part of a millisecond for a constructor that costs the reference tens of nanoseconds, and
// every one of those calls waits for the whole device (a resident search service: for its lease).  Blocks come from slabs
// instead (512 blocks of 4 KiB device + 2 KiB pinned memory per slab, kept until the process ends), and a block is
// initialised by the CPU THROUGH THE PCIe BAR (every byte of an MI300-class part's memory is CPU-visible): `new` makes no
// runtime call at all once a slab exists.  The writes are pushed through the device's host data path and waited for (bar_write);
// a kernel's start drops the caches' copy of the block; a resident service kernel acquires what was uploaded after its last
// look (upload tickets).  Without a large BAR (or with SLICESLICE_NO_BAR_WRITES=1) the image goes by one hipMemcpy.
constexpr size_t kBlockDevBytes = 4096, kBlockHostBytes = 2048, kBlockNeedleOff = 2048, kBlockNeedleMax = 2048;
constexpr size_t kOffFlags = 0, kOffBest = 256, kOffDone = 768, kOffBestDone = 1280, kCtlBytes = 1792;
constexpr uint32_t kBlocksPerSlab = 512;        // 2 MiB of device memory per slab: one page-table fragment

struct BlockPool {
    std::mutex mu;
    std::vector<uint8_t *> d_slabs, h_slabs;
    std::vector<uint32_t> free_blocks;          // slab << 16 | index
    int bar = -1;                               // 1: the CPU writes device memory directly
    volatile uint32_t *hdp_flush = nullptr;     // the device's HDP_MEM_COHERENCY_FLUSH_CNTL register (CPU-visible), or null
};
BlockPool g_pools[64];

int pool_acquire(int dev, uint32_t *id, uint8_t **d, uint8_t **h, bool *bar, volatile uint32_t **hdp_flush)
{
    BlockPool &bp = g_pools[dev];
    std::lock_guard<std::mutex> lock(bp.mu);
    if (bp.bar < 0) {
        int large = 0;
        const char *off = getenv("SLICESLICE_NO_BAR_WRITES");
        if (hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, dev) != hipSuccess) { large = 0; (void)hipGetLastError(); }
        bp.bar = large && !(off && off[0] == '1') ? 1 : 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) bp.hdp_flush = prop.hdpMemFlushCntl;
        else (void)hipGetLastError();
    }
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
    *bar = bp.bar == 1;
    *hdp_flush = bp.hdp_flush;
    return SS_OK;
}

void pool_release(int dev, uint32_t id)
{
    BlockPool &bp = g_pools[dev];
    std::lock_guard<std::mutex> lock(bp.mu);
    bp.free_blocks.push_back(id);
}

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
    p.h_flags = reinterpret_cast<int *>(hb + 0);
    p.h_best = reinterpret_cast<uint64_t *>(hb + 256);
    p.h_done = reinterpret_cast<long long *>(hb + 768);
    memset(hb, 0, kBlockHostBytes);                    // blocks are recycled: a stale value must not equal an epoch
    for (int k = 0; k < kSlots; ++k) p.find_tag[k] = kFindTagMax;
    // the block's image: flags 0 | minima all ones | completion counters 0 | keyed minima all ones | the needle
    const bool inside = s->n <= kBlockNeedleMax;
    alignas(16) uint8_t img[kBlockDevBytes];
    memset(img, 0, sizeof img);
    memset(img + kOffBest, 0xFF, kOffDone - kOffBest);
    memset(img + kOffBestDone, 0xFF, kCtlBytes - kOffBestDone);
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

// ---- kernel selection -------------------------------------------------------------------------------
// variant = 100000*B + 10000*OCC + 1000*LAYOUT + 100*MODE + 10*U + NT.  B: workgroup size (0/2 = 256 threads,
// 1 = 128, 3 = 512); OCC: at most OCC workgroups per CU through unused dynamic LDS (0 = no cap) - both are
// tuning aids (profiles/r01/workgroup_size_sweep.jsonl, occupancy_sweep.jsonl).  LAYOUT 0 = automatic, 1 = 16 bytes per lane throughout,
// 2 = 8-bytes-per-lane first phase (single-stream kernels).  U in {4,8} = pieces (KiB) per wave per tile; NT in {0,1} = plain /
// non-temporal first-byte stream; MODE (only meaningful for a filter pair 16 or more apart, i.e. d > 0): 0 = automatic,
// 1 = second load stream, 2 = one stream + cross-lane (ds_bpermute) position flags.  variant 0 = automatic:
// U = 4; d == 0 -> NT; 0 < d <= kShiftMaxD -> MODE 2 with NT; larger d -> MODE 1 with plain loads (a
// non-temporal line is not kept for the second stream's re-read; profiles/r01/readbench_8gib.txt).
struct Launch {
    int U;
    int nt;
    int mode;   // 0: d == 0, 1: two load streams, 2: shifted flags
    bool l8;    // 8-bytes-per-lane first phase (mode 0 / one-byte needles)
    uint32_t dyn_lds;   // unused dynamic LDS per workgroup (caps workgroups per CU; tuning: variant 10000*OCC)
    unsigned block;     // threads per workgroup: 128 / 256 / 512 (tuning: variant 100000*B, B = 1 / 2 / 3)
};

// Measured (tools/tune.py, profiles/r01/l8_short_needles.jsonl, two-tile workgroups): the 8-byte first phase
// is +5 % for one-byte needles (64 GiB: 7.46 vs 7.08 TB/s) but -5 % for two-byte filters on random bytes
// (positions 2, 3: 6.95 vs 7.35 TB/s): there one tile in sixteen holds a candidate and pays for the
// transposition into the 16-byte layout on top of the regular filter.  Automatic choice: one-byte needles
// only; 2xxx variants force it for tuning.

constexpr int kAutoU = 4;
constexpr int kAutoTilesPerBlock = 2;    // 32 KiB contiguous per workgroup at U = 4 (profiles/r01/tiles_per_block_sweep.jsonl)
constexpr uint64_t kShiftMaxD = 62;      // d + 1 halo chunks must fit one piece (tools/tune.py: wins up to d = 62)

// Unused dynamic LDS that leaves room for exactly `occ` workgroups of `block` threads per CU (160 KiB of LDS).
uint32_t occupancy_pad(int occ, unsigned block)
{
    const uint32_t per = (160u * 1024u) / (uint32_t)occ;
    const uint32_t fixed = (block / ss::kWave) * ss::kNeedleLds;
    // (1 KiB short of the share: the kernels also own a few bytes of static LDS - the completion word's workgroup flag -
    // and a workgroup's allocation is rounded up to the hardware's granule)
    uint32_t pad = per > fixed + 2048 ? ((per - fixed - 1024) & ~1023u) : 0;
    if (pad > 64u * 1024u - fixed) pad = 64u * 1024u - fixed;
    return pad;
}

// Workgroups per CU.  Since the cold half of the Problem left the registers (scan_kernels.hpp, ColdInKernarg) the multi-byte
// kernels need 77-83 VGPRs, so the register file would admit six workgroups of four waves per CU; how many actually run is
// set per launch through unused dynamic LDS.  Measured in one process on one buffer (tools/occ_probe.py,
// profiles/r03/occupancy_probe.jsonl; 16-byte needle on random bytes, 1 / 8 / 32 GiB): FOUR per CU 7.36 / 7.46 / 7.42 TB/s,
// five 7.01 / 7.20 / 7.17, six 7.00 / 7.24 / 7.23 - a streaming scan that rarely sees a candidate wants exactly one
// workgroup per SIMD quartet.  A scan that keeps meeting candidates wants latency hiding instead: on the i386 text, phrases of
// the manual's stock vocabulary run at 6.1-6.2 TB/s with four per CU and 7.1 with six, the reference's own pair (0, n-1) on
// text 3.9-4.4 against 5.0-5.8.  The library cannot see the haystack, so it goes by the needle: when EVERY filter byte is
// text-like (byte_rarity_rank >= 64: letters, digits, blanks, common punctuation, NUL) the haystack is presumably text and
// candidates are to be expected - six per CU; otherwise (a random or binary needle: its rarest bytes are in the filter) four.
// One-byte needles (8-byte loads, 36-68 VGPRs) stay at four: 7.28-7.44 TB/s either way.
Launch pick_variant(int variant, uint64_t d, bool one_byte, bool text_like)
{
    Launch l;
    l.U = kAutoU;
    l.mode = d == 0 ? 0 : (d <= kShiftMaxD ? 2 : 1);
    l.nt = l.mode == 1 ? 0 : 1;
    l.l8 = one_byte;
    l.block = ss::kBlock;
    l.dyn_lds = occupancy_pad(!one_byte && text_like ? 6 : 4, l.block);
    if (variant > 0) {
        l.dyn_lds = 0;
        if (variant >= 100000) {                                // Bxxxxx: workgroup size
            const int b = variant / 100000;
            l.block = b == 1 ? 128 : (b == 3 ? 512 : 256);
            variant %= 100000;
        }
        if (variant >= 10000) {                                 // OCCxxxx: at most OCC workgroups per CU (160 KiB LDS)
            l.dyn_lds = occupancy_pad(variant / 10000, l.block);
            variant %= 10000;
        }
        if (variant >= 1000) l.l8 = variant / 1000 == 2;       // 1xxx: 16-byte layout, 2xxx: 8-byte first phase
        variant %= 1000;
        const int m = variant / 100, u = (variant / 10) % 10;
        if (u == 4 || u == 8) l.U = u;
        if (d != 0 && m == 1) l.mode = 1;
        if (d != 0 && m == 2 && d <= 62) l.mode = 2;
        l.nt = (variant % 10) ? 1 : 0;
#ifndef SS_TUNING_VARIANTS
        // the default library holds ONE load flavour per mode (scan_launch.hpp::kernel_built): the digit selects among kernels
        // only in the tuning build, here the launch-shape digits of a variant keep working for every filter pair
        l.nt = l.mode == 1 ? 0 : 1;
#endif
    }
    return l;
}

// launch_scan_un<U, NT, FIND> is defined in scan_launch.hpp and explicitly instantiated in the
// scan_inst_*.hip translation units, so that the kernel families compile in parallel.
template <int U>
bool launch_scan_u(int nt, const ss::Problem &pr, int q, int mode, bool one_byte, const ss::Shape &shape, hipStream_t st,
                   void *flag, bool l8)
{
    if (nt == 0) return ss::launch_scan_un<U, 0, false>(pr, q, mode, one_byte, shape, st, flag, l8);
    return ss::launch_scan_un<U, 1, false>(pr, q, mode, one_byte, shape, st, flag, l8);
}

// Builds the Problem for (hay, len) and enqueues the scan.  find == false: *d_sink is an int flag, OR-ed
// (0 -> 1), never cleared.  find == true: *d_sink is a uint64, atomicMin'ed with find_base + offset of
// every match the grid sees (the leftmost one survives).  Preconditions: 1 <= n <= len.
// done_slot >= 0: the call owns flag slot `done_slot` and would like to wait on the slot's completion word
// instead of the stream; granted (*used_done = true) for grids of at most kDoneMaxBlocks workgroups.
constexpr uint64_t kDoneMaxBlocks = 256;         // one atomic per workgroup on ONE address: small grids only (4 MiB);
                                                 // measured: 1 KiB 8.6 vs 11.9 us per call, break-even near 1 MiB

// What a launch needs to know about a Problem besides the Problem itself.
struct ProblemShape {
    size_t position, position3;     // the second / third filter byte, relative to the first (ordered by dword: q3 <= Q)
    size_t fa;                      // index of the first filter byte
    bool one_byte;
};

// The Problem of (searcher, haystack): everything but the sink-side fields (epoch, host_flag, completion word), which the caller
// sets.  Preconditions: 1 <= n <= len.
void fill_problem(const ss_searcher *s, const uint8_t *d_needle, const void *d_hay, size_t len, uint64_t find_base, ss::Problem *out,
                  ProblemShape *shape)
{
    ss::Problem &pr = *out;
    const size_t n = s->n;
    const bool one_byte = n == 1;
    // The filter stream starts at the FIRST filter byte: candidate i is tested through hay[fa + i] == needle[fa]
    // and hay[fb + i] == needle[fb], so the kernel's aligned coordinates are those of hay + fa, while matches
    // are verified (and reported) at hay + i.  Bytes in front of hay + fa are never candidates (their index
    // wraps and fails `i < end`), and the last byte either stream touches is hay[len - n + fb] <= hay[len - 1].
    const size_t fa = one_byte ? 0 : s->fa, fb = one_byte ? 0 : s->fb;
    const uint8_t *hf = static_cast<const uint8_t *>(d_hay) + fa;
    pr.hay = static_cast<const uint8_t *>(d_hay);
    pr.mis = (uint32_t)((uintptr_t)hf & 15);
    pr.base = hf - pr.mis;
    pr.needle = d_needle;
    pr.n = n;
    pr.end = (uint64_t)len - n + 1;
    pr.nchunks_all = ((uint64_t)pr.mis + (len - fa) + 15) / 16;
    pr.npieces = (((uint64_t)pr.mis + pr.end + 15) / 16 + 63) / 64;
    size_t position = fb - fa;                          // distance between the two filter bytes
    pr.d = position / 16;
    // third first-phase byte (single-stream kernels only, i.e. d == 0); "none" = the second byte once more
    const bool three = !one_byte && pr.d == 0 && s->fc > fa && s->fc - fa <= 15 && s->fc < n;
    size_t position3 = three ? s->fc - fa : position % 16;
    // The two further bytes are interchangeable; the kernels are instantiated for "the third byte's dword is not behind
    // the second's" only (10 copies of the first phase instead of 16 - and two of the six others, second byte in dword 0
    // with the third in dword 1 or 3, came out of the compiler waiting for all four loads of a tile before the first
    // xor: 6.3-6.4 instead of 7.4 TB/s, profiles/r02/ab_filter_triples.jsonl).
    if (three && position3 / 4 > position / 4) std::swap(position, position3);
    const uint32_t sh = (uint32_t)(position % 16);
    pr.r = sh % 4;
    pr.n0x4 = 0x01010101u * s->needle[fa];
    pr.nlx4 = 0x01010101u * s->needle[one_byte ? 0 : fa + position];
    pr.q3 = (uint32_t)(position3 / 4);
    pr.r3 = (uint32_t)(position3 % 4);
    pr.n3x4 = 0x01010101u * s->needle[one_byte ? 0 : fa + position3];
    // second-level filter: up to 15 further needle bytes behind the first filter byte
    pr.norder = ss::build_refine_order(s->needle.data() + fa, n - fa, position, pr.order_idx, pr.order_val,
                                       pr.d == 0 ? (uint64_t)position3 : ~0ull);
    pr.find_base = find_base;
    pr.host_flag = nullptr;
    pr.epoch = 1;
    pr.done_counter = nullptr;
    pr.host_done = nullptr;
    pr.done_target = pr.done_hi = 0;
    pr.flags = 0;
    pr.pad_ = 0;
    // exact in-register verification: the needle ends at most 16 bytes behind the first filter byte (lib.rs:222-241)
    pr.exact_len = 0;
    pr.tail16[0] = pr.tail16[1] = pr.tail16[2] = pr.tail16[3] = 0;
    if (!one_byte && pr.d == 0 && n - fa <= 16) {
        // ... plus as many of the fa bytes in front of it as sixteen leave room for: a needle of up to 16 bytes is compared whole
        const size_t behind = n - fa, back = std::min(fa, 16 - behind), el = behind + back;
        pr.exact_len = (uint32_t)(el | (back << 8));
        uint8_t t16[16] = {0};
        memcpy(t16, s->needle.data() + fa - back, el);
        memcpy(pr.tail16, t16, 16);
    }
    shape->position = position;
    shape->position3 = position3;
    shape->fa = fa;
    shape->one_byte = one_byte;
}

int enqueue_scan(const ss_searcher *s, PerDevice *pd, const void *d_hay, size_t len, hipStream_t st,
                 void *d_sink, bool find = false, uint64_t find_base = 0, int *host_flag = nullptr, int epoch = 1,
                 int done_slot = -1, bool *used_done = nullptr)
{
    if (s->debug_fail_scans.load(std::memory_order_relaxed) > 0 && s->debug_fail_scans.fetch_sub(1) > 0)
        return fail(SS_ERR_HIP, "injected scan failure (ss_debug_fail_next_scans)");
    void *d_flag = d_sink;
    ss::Problem pr;
    ProblemShape ps;
    fill_problem(s, pd->d_needle, d_hay, len, find_base, &pr, &ps);
    pr.host_flag = host_flag;
    pr.epoch = epoch;
    const bool one_byte = ps.one_byte;
    const size_t fa = ps.fa, position = ps.position, position3 = ps.position3;
    const uint32_t sh = (uint32_t)(position % 16);

    const bool text_like = !one_byte && ss::byte_rarity_rank(s->needle[fa]) >= 64 && ss::byte_rarity_rank(s->needle[fa + position]) >= 64 &&
                           ss::byte_rarity_rank(s->needle[fa + position3]) >= 64;
    const Launch l = pick_variant(s->variant, pr.d, one_byte, text_like);
    const uint64_t wpb = l.block / ss::kWave;
    const uint64_t ntiles = (pr.npieces + wpb * l.U - 1) / (wpb * l.U);
    uint64_t blocks, tpb;
    if (s->grid > 0) {
        blocks = (uint64_t)s->grid;
        if (blocks > ntiles) blocks = ntiles;
        tpb = 0;
    } else {
        if (s->grid < 0) {
            tpb = (uint64_t)(-(int64_t)s->grid);
        } else {
            // Short-lived workgroups: two tiles (32 KiB) each from 2 GiB up (from 1 GiB for filter pairs >= 16 apart), one
            // below.  The hardware dispatcher hands out tiles in address order, so the set of lines in flight
            // stays one narrow, advancing window, and a fresh workgroup issues its loads the moment a slot
            // frees up.  Measured (profiles/r01/tiles_per_block_sweep.jsonl, tiles_1_vs_2.txt; 16-byte needle):
            // 64 GiB 7.40-7.43 TB/s at 2 tiles per workgroup vs 7.28 at 4, 7.21 at 8, 7.11 at 64.  One tile is
            // +2 % at 1 GiB, within +-0.7 % from 4 GiB up (and +1.5 % for one-byte needles), but twice as many
            // workgroups have to be drained after an early match (the entry peek in scan_kernel), and the
            // cross-lane kernels (pairs >= 16 apart) lose 5 % with it: each wave re-loads its halo chunks per tile.
            DeviceInfo di;
            if (int rc = device_info(pd->dev, &di)) return rc;
            tpb = ntiles / ((uint64_t)di.cus * (l.mode == 0 ? 256 : 128));
            if (tpb > (uint64_t)kAutoTilesPerBlock) tpb = kAutoTilesPerBlock;
            if (tpb < 1) tpb = 1;
        }
        blocks = (ntiles + tpb - 1) / tpb;
        while (blocks > 0x7fffffffull) {        // gridDim.x limit
            tpb *= 2;
            blocks = (ntiles + tpb - 1) / tpb;
        }
    }
    if (blocks < 1) blocks = 1;
    const ss::Shape shape = {(unsigned)blocks, l.block, tpb, l.dyn_lds};
    if (used_done) *used_done = false;
    if (done_slot >= 0 && used_done && blocks <= kDoneMaxBlocks && (!find || len < (1ull << ss::kFindOffsetBits))) {
        const int k = done_slot;
        if (pd->done_low[k] > kDoneLowMax || (find && pd->find_tag[k] == 0)) start_over(pd, k);
        pr.done_counter = pd->d_done + k;
        pr.host_done = pd->h_done + k;
        pr.done_target = pd->done_low[k] + (uint32_t)blocks;
        pr.done_hi = pd->done_hi[k];
        pr.flags |= ss::kProblemCounted;
        pd->done_low[k] = pr.done_target;                // (a launch that fails starts the slot over)
        pr.host_flag = nullptr;                          // the completion word carries the answer
        if (find) {                                      // keyed minimum in the slot's own word: see scan_kernel
            d_flag = pd->d_best_done + k;
            pr.find_base += (uint64_t)pd->find_tag[k]-- << ss::kFindOffsetBits;
        }
        *used_done = true;
    }

    ThreadTimer &tm = g_timer;
    const bool timed = s->timing && pd->dev >= 0 && pd->dev < kMaxDevices;
    if (timed) {
        if (!tm.ev0[pd->dev]) {
            HIP_TRY(hipEventCreate(&tm.ev0[pd->dev]));
            HIP_TRY(hipEventCreate(&tm.ev1[pd->dev]));
        }
        HIP_TRY(hipEventRecord(tm.ev0[pd->dev], st));
    }
    const int q = (int)(sh / 4);
    bool launched = false;
    if (find) {   // one tile shape for find(): U = 4
        if (l.U != 4) return fail(SS_ERR_ARGUMENT, "find supports the U = 4 kernels only");
        if (l.nt) launched = ss::launch_scan_un<4, 1, true>(pr, q, l.mode, one_byte, shape, st, d_flag, false);
        else launched = ss::launch_scan_un<4, 0, true>(pr, q, l.mode, one_byte, shape, st, d_flag, false);
    } else if (l.U == 8) {
#ifdef SS_TUNING_VARIANTS
        launched = launch_scan_u<8>(l.nt, pr, q, l.mode, one_byte, shape, st, d_flag, l.l8);
#endif
    } else {
        launched = launch_scan_u<4>(l.nt, pr, q, l.mode, one_byte, shape, st, d_flag, l.l8);
    }
    if (!launched)
        return fail(SS_ERR_ARGUMENT, "kernel variant %d (U = %d, %s loads, mode %d%s) is not part of this build: the default library holds "
                                     "the kernels the constructors and ss_searcher_set_filter* can select; the rest is in the tuning build "
                                     "(-DSS_TUNING_VARIANTS, libsliceslice_hip_tuning.so)",
                    s->variant, l.U, l.nt ? "non-temporal" : "plain", l.mode, l.l8 ? ", 8-byte first phase" : "");
    HIP_TRY(hipGetLastError());
    if (timed) {
        HIP_TRY(hipEventRecord(tm.ev1[pd->dev], st));
        tm.owner = s;
        tm.dev = pd->dev;
    }
    return SS_OK;
}

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
// position gets a partner close in front of it (choose_anchor).  ss_searcher_set_filter sets any pair verbatim.
constexpr size_t kFilterWindow = 1024;

inline int rarity_class(uint8_t b)
{
    const int r = ss::byte_rarity_rank(b);
    return r < 64 ? 0 : r;          // everything that is not text-like counts as equally rare
}

// Cost of one filter byte: the static class above, or - with a byte histogram of (a sample of) the haystack -
// 8 * log2(count + 1): summing costs then compares PRODUCTS of frequencies, which is what the candidate rate of
// a multi-byte filter is (bytes taken as independent).
struct ByteCost {
    int cost[256];
    explicit ByteCost(const uint64_t *hist)
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
    int operator()(uint8_t b) const { return cost[b]; }
};

// Third byte for a given pair (single-stream kernels: fb - fa <= 15): the rarest byte among needle[fa+1 .. fa+15]
// other than needle[fb]'s index; ties to the later byte.  Returns fb when there is none.
size_t choose_third(const uint8_t *needle, size_t n, size_t fa, size_t fb, const ByteCost &cost)
{
    if (n < 3 || fb < fa || fb - fa > 15) return fb;
    size_t best = fb;
    int bc = INT_MAX;
    for (size_t k = fa + 1; k < n && k <= fa + 15; ++k) {
        if (k == fb) continue;
        const int c = cost(needle[k]);
        if (c <= bc) {
            bc = c;
            best = k;
        }
    }
    return best;
}

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

// with_position: the caller's byte plus the reference's partner needle[0] and one more byte when position < 16 (or always,
// without the third byte at a distance >= 16, when the reference's exact pair is asked for), else choose_anchor's partner.
void filter_for_position(const uint8_t *needle, size_t n, size_t position, bool exact_pair, size_t *fa, size_t *fb, size_t *fc,
                         const ByteCost &cost)
{
    *fa = 0;
    *fb = *fc = n >= 2 ? position : 0;
    if (n < 2) return;
    if (position >= 16 && !exact_pair) choose_anchor(needle, n, position, fa, fc, cost);
    else *fc = choose_third(needle, n, *fa, *fb, cost);
}

void choose_filter_pair(const uint8_t *needle, size_t n, size_t *fa, size_t *fb)
{
    size_t fc;
    choose_filter_triple(needle, n, fa, fb, &fc, ByteCost(nullptr));
}

int make_searcher(const uint8_t *needle, size_t n, size_t position, bool auto_filter, ss_searcher **out, bool exact_pair = false)
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
    s->needle.assign(needle, needle + n);
    s->n = n;
    s->position = position;
    const ByteCost cost(nullptr);
    if (auto_filter) choose_filter_triple(s->needle.data(), n, &s->fa, &s->fb, &s->fc, cost);
    else filter_for_position(s->needle.data(), n, position, exact_pair, &s->fa, &s->fb, &s->fc, cost);
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) {      // uploads the needle to the current device now
        delete s;
        return rc;
    }
    *out = s;
    return SS_OK;
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
    s->gate.store(0, std::memory_order_release);
    return SS_OK;
}

}  // namespace

extern "C" {

const char *ss_last_error(void) { return g_err; }
const char *ss_version(void)
{
#ifdef SS_TUNING_VARIANTS
    return "sliceslice-hip 0.3 (gfx950, tuning build: every kernel variant)";
#else
    return "sliceslice-hip 0.3 (gfx950)";
#endif
}

int ss_searcher_with_position(const uint8_t *needle, size_t n, size_t position, ss_searcher **out)
{
    // SLICESLICE_AUTO_FILTER=0: the reference's pair (needle[0], needle[position]) at any distance
    const char *e = getenv("SLICESLICE_AUTO_FILTER");
    return make_searcher(needle, n, position, false, out, e && e[0] == '0');
}

int ss_searcher_new(const uint8_t *needle, size_t n, ss_searcher **out)
{
    // x86.rs:457: position = n.wrapping_sub(1) - what ss_searcher_position keeps reporting.  The filter bytes
    // the device tests are chosen by choose_filter_triple (SLICESLICE_AUTO_FILTER=0: as with_position(n-1)).
    const char *e = getenv("SLICESLICE_AUTO_FILTER");
    const bool exact = e && e[0] == '0';
    return make_searcher(needle, n, n - 1, !exact, out, exact);
}

int ss_searcher_set_filter(ss_searcher *s, size_t first, size_t second)
{
    if (!s) return fail(SS_ERR_ARGUMENT, "searcher is NULL");
    if (s->n < 2) {
        if (first != 0 || second != 0) return fail(SS_ERR_POSITION, "needles shorter than two bytes have no filter pair");
        return SS_OK;
    }
    if (first > second || second >= s->n) return fail(SS_ERR_POSITION, "filter pair (%zu, %zu) out of range for a needle of %zu bytes", first, second, s->n);
    return store_filter(s, first, second, second);      // a plain two-byte filter; ss_searcher_set_filter3 adds a third byte
}

int ss_searcher_set_filter3(ss_searcher *s, size_t first, size_t second, size_t third)
{
    if (!s) return fail(SS_ERR_ARGUMENT, "searcher is NULL");
    if (s->n < 2 || third == second) return ss_searcher_set_filter(s, first, second);
    if (first > second || second >= s->n) return fail(SS_ERR_POSITION, "filter pair (%zu, %zu) out of range for a needle of %zu bytes", first, second, s->n);
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

int ss_choose_filter_triple(const uint8_t *needle, size_t n, size_t *first, size_t *second, size_t *third)
{
    if (!first || !second || !third || (n && !needle)) return fail(SS_ERR_ARGUMENT, "NULL argument");
    choose_filter_triple(needle, n, first, second, third, ByteCost(nullptr));
    return SS_OK;
}

int ss_choose_filter_for_position(const uint8_t *needle, size_t n, size_t position, size_t *first, size_t *second, size_t *third)
{
    if (!first || !second || !third || (n && !needle)) return fail(SS_ERR_ARGUMENT, "NULL argument");
    *first = *second = *third = 0;
    if (n == 1 && position != 0) return fail(SS_ERR_POSITION, "position must be 0 for a one-byte needle");
    if (n >= 2 && position >= n) return fail(SS_ERR_POSITION, "position %zu out of range for needle of %zu bytes", position, n);
    const char *e = getenv("SLICESLICE_AUTO_FILTER");
    filter_for_position(needle, n, position, e && e[0] == '0', first, second, third, ByteCost(nullptr));
    return SS_OK;
}

int ss_choose_filter_triple_hist(const uint8_t *needle, size_t n, const uint64_t hist[256], size_t *first, size_t *second,
                                 size_t *third)
{
    if (!first || !second || !third || (n && !needle)) return fail(SS_ERR_ARGUMENT, "NULL argument");
    choose_filter_triple(needle, n, first, second, third, ByteCost(hist));
    return SS_OK;
}

int ss_choose_filter_pair(const uint8_t *needle, size_t n, size_t *first, size_t *second)
{
    if (!first || !second || (n && !needle)) return fail(SS_ERR_ARGUMENT, "NULL argument");
    choose_filter_pair(needle, n, first, second);
    return SS_OK;
}

int ss_searcher_filter(const ss_searcher *s, size_t *first, size_t *second)
{
    if (!s || !first || !second) return fail(SS_ERR_ARGUMENT, "NULL argument");
    *first = s->fa;
    *second = s->fb;
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
        // (once exit() has begun the pools - function-scope-free statics - may already be gone: a searcher dropped by a thread
        // that outlives main keeps its block)
        if (!process_exiting()) pool_release(p.dev, p.block);
    }
    (void)hipSetDevice(cur);
    if (g_timer.owner == s) g_timer.owner = nullptr;
    delete s;
}

size_t ss_searcher_needle_len(const ss_searcher *s) { return s ? s->n : 0; }
size_t ss_searcher_position(const ss_searcher *s) { return s ? s->position : 0; }

int ss_searcher_set_timing(ss_searcher *s, int enabled)
{
    if (!s) return fail(SS_ERR_ARGUMENT, "searcher is NULL");
    s->timing = enabled != 0;
    return SS_OK;
}

int ss_searcher_last_kernel_ms(const ss_searcher *s, float *ms)
{
    if (!s || !ms) return fail(SS_ERR_ARGUMENT, "NULL argument");
    ThreadTimer &tm = g_timer;
    if (tm.owner != s || tm.dev < 0) return fail(SS_ERR_ARGUMENT, "no timed scan has been launched through this searcher by the calling thread");
    HIP_TRY(hipEventSynchronize(tm.ev1[tm.dev]));
    HIP_TRY(hipEventElapsedTime(ms, tm.ev0[tm.dev], tm.ev1[tm.dev]));
    return SS_OK;
}

// Test hooks: move the epoch counters close to the 2^31 wrap so that tests can cross it.
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

int ss_search_device_async(const ss_searcher *s, const void *d_haystack, size_t len, void *hip_stream,
                           int *d_found)
{
    if (!s || !d_found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !d_haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    if (s->n == 0) {                                    // N0: true for every haystack (x86.rs:500)
        static const int one = 1;
        HIP_TRY(hipMemcpyAsync(d_found, &one, sizeof one, hipMemcpyHostToDevice, st));
        return SS_OK;
    }
    if (len < s->n) return SS_OK;                       // cannot occur; flag untouched
    s->used_async.store(true, std::memory_order_release);
    return enqueue_scan(s, pd, d_haystack, len, st, d_found);
}

namespace {

// Scans too large for the workgroup count of the completion word still answer through a pinned word when the scan is short
// enough to be waited for by spinning: a one-lane kernel behind the scan (behind the all-reduce, for a sharded search) stores
// epoch << 1 | found.  A host that reaches hipStreamSynchronize before the work is done pays a wake-up on top of it (a 16 MiB
// search: 15.2 us per call with the stream wait, 10.7 with the spin; an 8 GiB shard: 2-7 us per search, box to box).  The spin is bounded by twice the time the scan can
// possibly take at HBM speed (+ 300 us); after that - the stream was busy with other work - the stream wait takes over.
constexpr double kSpinMaxEstimateUs = 20000.0;
// ... and the sharded entry points only bother from a few MiB per shard: below, the stream wait returns at once (the work is
// done before the host gets there) and the extra launch costs 3-4 us (1 MiB shards: 24.6 -> 27.4 us per search with it).
constexpr double kSpinMinEstimateUs = 0.5;
inline double scan_estimate_us(size_t len) { return (double)len / 7.0e6; }          // 7 TB/s

bool spin_for_word(const long long *word, int epoch, double estimate_us, int *found)
{
    const auto t0 = std::chrono::steady_clock::now();
    const auto budget = std::chrono::microseconds((long long)(2.0 * estimate_us) + 300);
    for (unsigned spins = 0;; ++spins) {
        const long long v = __atomic_load_n(word, __ATOMIC_ACQUIRE);
        if (((uint32_t)v >> 1) == (uint32_t)epoch) {
            *found = (int)(v & 1);
            return true;
        }
        cpu_relax();
        if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > budget) return false;
    }
}

// The answer word of a sharded search (signal_flag_kernel (pair form)): epoch << 2 | "a rank failed" << 1 | found.
bool spin_for_shard_word(const long long *word, int epoch, double estimate_us, int *found, int *failed)
{
    const auto t0 = std::chrono::steady_clock::now();
    const auto budget = std::chrono::microseconds((long long)(2.0 * estimate_us) + 300);
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long v = (unsigned long long)__atomic_load_n(word, __ATOMIC_ACQUIRE);
        if ((v >> 2) == (unsigned long long)(uint32_t)epoch) {
            *found = (int)(v & 1);
            *failed = (int)((v >> 1) & 1);
            return true;
        }
        cpu_relax();
        if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > budget) return false;
    }
}

}  // namespace

}  // extern "C"

struct ss_service;
extern "C" int ss_service_search(ss_service *sv, const ss_searcher *s, const void *d_haystack, size_t len, int *found);
namespace {
ss_service *default_service(int dev);
constexpr size_t kServiceMaxLen = (size_t)8 << 20;        // beyond this the launch path's 8 us no longer matter
}

extern "C" {

int ss_search_device(const ss_searcher *s, const void *d_haystack, size_t len, void *hip_stream, int *found)
{
    if (!s || !found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !d_haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    if (s->n == 0) { *found = 1; return SS_OK; }        // x86.rs:500
    if (len < s->n) { *found = 0; return SS_OK; }       // x86.rs:357-359 (len == n is decided on the device)
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    // A resident search service on this device (ss_service_set_default, or SLICESLICE_SERVICE=1) answers small searches
    // without a launch.  It cannot be ordered behind work that is still pending on the caller's stream, so it is used
    // only when that stream is idle - anything else takes the launch path below, as do filter pairs 16 or more apart,
    // kernel-variant / grid overrides and timed searches.
    if (len <= kServiceMaxLen && s->variant == 0 && s->grid == 0 && !s->timing && (s->n == 1 || s->fb - s->fa < 16)) {
        if (ss_service *sv = default_service(pd->dev)) {
            if (hipStreamQuery(st) == hipSuccess) return ss_service_search(sv, s, d_haystack, len, found);
            (void)hipGetLastError();
        }
    }
    const int k = acquire_slot(s, pd);
    // The wave that finds a match stores the call's epoch to the device flag (polled by the grid for the
    // early exit) AND to its pinned-host mirror, so the answer needs neither a device-to-host copy nor a
    // reset of the slot afterwards: launch, wait for the stream, compare.
    const int epoch = next_epoch(pd, k);                // the slot is owned by this call
    static const bool spin_ok = []() { const char *v = getenv("SLICESLICE_SPIN_WAIT"); return !(v && v[0] == '0'); }();
    // the slot's completion word may hold a find()'s answer (offset + 1), which could pass for 2 * epoch + found
    __atomic_store_n(pd->h_done + k, 0ll, __ATOMIC_RELAXED);
    bool used_done = false;
    int rc = enqueue_scan(s, pd, d_haystack, len, st, pd->d_flags + k, false, 0, pd->h_flags + k, epoch, spin_ok ? k : -1,
                          &used_done);
    bool answered = false;
    // the answer word of a small grid: found-half of the slot's counter << 32 | epoch << 1 | found
    auto take = [&](long long v) {
        if (((uint32_t)v >> 1) != (uint32_t)epoch) return false;
        *found = (int)(v & 1);
        pd->done_hi[k] = (uint32_t)((unsigned long long)v >> 32);
        return true;
    };
    if (rc == SS_OK && used_done) {
        // Small grid: the workgroup that completes the count stores the answer word to the slot's pinned word.  Spin on it
        // for a bounded time (the whole call is a few microseconds); after that - a long kernel behind other work on the
        // stream, or a fault - fall back to the stream wait, which also reports errors.
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            if (take(__atomic_load_n(pd->h_done + k, __ATOMIC_ACQUIRE))) {
                answered = true;
                // every 256th call still waits for the stream, so that the runtime retires its completed commands
                // in bounded batches instead of whenever the caller next synchronises
                if ((epoch & 255) == 0) (void)hipStreamSynchronize(st);
                break;
            }
            cpu_relax();
            if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
        }
    }
    if (rc == SS_OK && !used_done && spin_ok && scan_estimate_us(len) <= kSpinMaxEstimateUs) {
        // larger grid: the word is written by a one-lane kernel behind the scan
        ss::signal_flag_kernel<<<1, 1, 0, st>>>(pd->d_flags + k, epoch, pd->h_done + k, 0);
        if (hipGetLastError() == hipSuccess && spin_for_word(pd->h_done + k, epoch, scan_estimate_us(len), found)) {
            answered = true;
            if ((epoch & 255) == 0) (void)hipStreamSynchronize(st);
        }
    }
    if (rc == SS_OK && !answered) {
        const hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(SS_ERR_HIP, "stream wait: %s", hipGetErrorString(e));
        else if (used_done) { if (!take(__atomic_load_n(pd->h_done + k, __ATOMIC_ACQUIRE))) rc = fail(SS_ERR_HIP, "the completion word was not written"); }
        else *found = __atomic_load_n(pd->h_flags + k, __ATOMIC_ACQUIRE) == epoch;
    }
    if (rc != SS_OK) start_over(pd, k);             // a failed launch may have left a partial workgroup count behind
    release_slot(s, pd, k);
    return rc;
}

int ss_find_device_async(const ss_searcher *s, const void *d_haystack, size_t len, uint64_t base_offset,
                         void *hip_stream, uint64_t *d_best)
{
    if (!s || !d_best) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !d_haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    if (len < s->n) return SS_OK;
    if (s->n == 0) {        // the empty needle matches at offset 0 of every haystack
        if (base_offset != 0) return SS_OK;     // only the first shard reports it
        static const uint64_t zero = 0;
        HIP_TRY(hipMemcpyAsync(d_best, &zero, sizeof zero, hipMemcpyHostToDevice, st));
        return SS_OK;
    }
    s->used_async.store(true, std::memory_order_release);
    return enqueue_scan(s, pd, d_haystack, len, st, d_best, true, base_offset);
}

int ss_find_device(const ss_searcher *s, const void *d_haystack, size_t len, void *hip_stream, uint64_t *position)
{
    if (!s || !position) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !d_haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    if (s->n == 0) { *position = 0; return SS_OK; }
    if (len < s->n) { *position = SS_NPOS; return SS_OK; }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    const int k = acquire_slot(s, pd);
    // Small grid: the workgroup that completes the count stores the answer - offset + 1, or all ones - to the slot's
    // completion word (zeroed here first: the word also serves ss_search_device, whose values carry an epoch), and the host
    // spins on the word as ss_search_device does; the minimum lives in the slot's keyed word (enqueue_scan), which needs no
    // re-arming.  Larger grids: slots of d_best are all-ones whenever they are free; a one-lane kernel behind the scan
    // stores the minimum to the slot's pinned mirror (no device-to-host copy command) and re-arms the slot.
    static const bool spin_ok = []() { const char *v = getenv("SLICESLICE_SPIN_WAIT"); return !(v && v[0] == '0'); }();
    __atomic_store_n(pd->h_done + k, 0ll, __ATOMIC_RELAXED);
    bool used_done = false;
    int rc = enqueue_scan(s, pd, d_haystack, len, st, pd->d_best + k, true, 0, nullptr, 1, spin_ok ? k : -1, &used_done);
    bool answered = false;
    if (rc == SS_OK && used_done) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            const long long v = __atomic_load_n(pd->h_done + k, __ATOMIC_ACQUIRE);
            if (v != 0) {
                *position = v == -1ll ? SS_NPOS : (uint64_t)v - 1;
                answered = true;
                // as in ss_search_device: a real stream wait now and then lets the runtime retire its commands
                if ((next_epoch(pd, k) & 255) == 0) (void)hipStreamSynchronize(st);
                break;
            }
            cpu_relax();
            if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
        }
        if (!answered) {
            const hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) {
                rc = fail(SS_ERR_HIP, "stream wait: %s", hipGetErrorString(e));
            } else {
                const long long v = __atomic_load_n(pd->h_done + k, __ATOMIC_ACQUIRE);
                if (v == 0) rc = fail(SS_ERR_HIP, "find: the completion word was not written");
                else *position = v == -1ll ? SS_NPOS : (uint64_t)v - 1;
                answered = rc == SS_OK;
            }
        }
    } else if (rc == SS_OK) {
        // the pinned mirror starts as "pending" (a value no minimum can take), so that a scan short enough to be waited for
        // by spinning (see spin_for_word) is: the one-lane kernel's store ends the wait
        constexpr uint64_t kPending = ~0ull - 1;
        __atomic_store_n(pd->h_best + k, kPending, __ATOMIC_RELAXED);
        ss::publish_best_kernel<<<1, 1, 0, st>>>(pd->d_best + k, pd->h_best + k, 0);
        hipError_t e = hipGetLastError();
        bool have = false;
        if (e == hipSuccess && spin_ok && scan_estimate_us(len) <= kSpinMaxEstimateUs) {
            const auto t0 = std::chrono::steady_clock::now();
            const auto budget = std::chrono::microseconds((long long)(2.0 * scan_estimate_us(len)) + 300);
            for (unsigned spins = 0; !have; ++spins) {
                have = __atomic_load_n(pd->h_best + k, __ATOMIC_ACQUIRE) != kPending;
                if (!have) {
                    cpu_relax();
                    if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > budget) break;
                }
            }
            if (have && (next_epoch(pd, k) & 255) == 0) (void)hipStreamSynchronize(st);
        }
        if (e == hipSuccess && !have) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(SS_ERR_HIP, "position read-back: %s", hipGetErrorString(e));
        else *position = __atomic_load_n(pd->h_best + k, __ATOMIC_ACQUIRE);
    }
    if (rc != SS_OK) {
        start_over(pd, k);                                   // a failed launch may have left a partial workgroup count behind
        (void)hipMemset(pd->d_best + k, 0xFF, sizeof(uint64_t));
    }
    release_slot(s, pd, k);
    return rc;
}

// ---- staging sets of the host-buffer and file front ends ---------------------------------------------------
// Device buffers + streams (+ pinned host buffers for the file reader).  Creating them per call costs ~0.3 ms
// (hipMalloc, stream create/destroy) and pinning 3 x 64 MiB ~10 ms - more than uploading and scanning a small
// haystack - so one set per device is kept for the life of the process and lent to one call at a time; a
// concurrent call builds a private set.
namespace {

constexpr int kStageBuf = 3;
struct Staging {
    uint8_t *h[kStageBuf] = {nullptr, nullptr, nullptr};
    uint8_t *d[kStageBuf] = {nullptr, nullptr, nullptr};
    hipStream_t st[kStageBuf] = {nullptr, nullptr, nullptr};
    size_t cap_d = 0, cap_h = 0;      // bytes per device / pinned buffer
    int nbuf = 0;
    void release()
    {
        for (int b = 0; b < kStageBuf; ++b) {
            if (st[b]) { (void)hipStreamSynchronize(st[b]); (void)hipStreamDestroy(st[b]); }
            if (d[b]) (void)hipFree(d[b]);
            if (h[b]) (void)hipHostFree(h[b]);
            st[b] = nullptr; d[b] = nullptr; h[b] = nullptr;
        }
        cap_d = cap_h = 0;
        nbuf = 0;
    }
    bool ensure(int want_nbuf, size_t want_cap, bool pinned)     // grow-only
    {
        if (want_cap < ((size_t)1 << 20)) want_cap = (size_t)1 << 20;      // do not regrow for every small call
        if (nbuf >= want_nbuf && cap_d >= want_cap && (!pinned || cap_h >= want_cap)) return true;
        if (want_cap < cap_d) want_cap = cap_d;
        if (want_nbuf < nbuf) want_nbuf = nbuf;
        const bool want_pinned = pinned || cap_h > 0;
        release();
        for (int b = 0; b < want_nbuf; ++b) {
            if ((want_pinned && hipHostMalloc((void **)&h[b], want_cap, hipHostMallocDefault) != hipSuccess) ||
                hipMalloc((void **)&d[b], want_cap) != hipSuccess ||
                hipStreamCreateWithFlags(&st[b], hipStreamNonBlocking) != hipSuccess) {
                release();
                return false;
            }
        }
        cap_d = want_cap;
        cap_h = want_pinned ? want_cap : 0;
        nbuf = want_nbuf;
        return true;
    }
};
std::mutex g_staging_mu[kMaxDevices];
Staging g_staging[kMaxDevices];

// Lends the device's cached set when it is free, `mine` otherwise; `mine` is released by its destructor-like
// call site (Lease::done).
struct Lease {
    Staging mine, *set = &mine;
    std::unique_lock<std::mutex> lock;
    explicit Lease(int dev)
    {
        if (dev >= 0 && dev < kMaxDevices) {
            lock = std::unique_lock<std::mutex>(g_staging_mu[dev], std::try_to_lock);
            if (lock.owns_lock()) set = &g_staging[dev];
        }
    }
    ~Lease()
    {
        if (set == &mine) {
            mine.release();
        } else {
            for (int b = 0; b < set->nbuf; ++b) (void)hipStreamSynchronize(set->st[b]);   // nothing of this call in flight
        }
    }
};

}  // namespace

namespace {

// Small host slices skip the upload command altogether: the bytes are copied (by the CPU) into a pinned, device-visible
// buffer that belongs to the calling thread, and the scan reads them straight over PCIe - one launch, one completion
// word, no hipMemcpyAsync (a copy command costs ~6 us whatever its size; 64 KiB over PCIe cost ~1 us).
constexpr size_t kZeroCopyMax = 64u << 10;
struct ThreadPinned {
    uint8_t *p = nullptr;
    hipStream_t st[kMaxDevices] = {nullptr};      // one non-blocking stream per device this thread has searched on
    ~ThreadPinned()
    {
        if (process_exiting()) return;                  // leak: see ExitMark
        if (p) (void)hipHostFree(p);
        for (hipStream_t q : st)
            if (q) (void)hipStreamDestroy(q);
    }
};
thread_local ThreadPinned g_small_host;

// the calling thread's pinned copy of a small host slice and its stream on the current device, or nullptr (too large,
// switched off, no pinned memory / stream)
const uint8_t *small_host_copy(const uint8_t *haystack, size_t len, hipStream_t *stream)
{
    if (len > kZeroCopyMax) return nullptr;
    static const bool zero_copy = []() { const char *v = getenv("SLICESLICE_HOST_ZERO_COPY"); return !(v && v[0] == '0'); }();
    if (!zero_copy) return nullptr;
    ThreadPinned &tp = g_small_host;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    if (!tp.st[dev] && hipStreamCreateWithFlags(&tp.st[dev], hipStreamNonBlocking) != hipSuccess) tp.st[dev] = nullptr;
    if (!tp.p && hipHostMalloc((void **)&tp.p, kZeroCopyMax + 64, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) tp.p = nullptr;
    if (!tp.p || !tp.st[dev]) return nullptr;
    memcpy(tp.p, haystack, len);
    *stream = tp.st[dev];
    return tp.p;
}

}  // namespace

}  // extern "C"

namespace {
// ss_search_host, with a word shared by the threads of ss_search_host_all: a thread that has found the needle says so, and the
// others stop issuing chunks (the answer is an OR: theirs no longer matters).
int search_host_impl(const ss_searcher *s, const uint8_t *haystack, size_t len, int *found, std::atomic<int> *somebody_found);
}  // namespace

extern "C" {

int ss_search_host(const ss_searcher *s, const uint8_t *haystack, size_t len, int *found)
{
    return search_host_impl(s, haystack, len, found, nullptr);
}

}  // extern "C"

namespace {

int search_host_impl(const ss_searcher *s, const uint8_t *haystack, size_t len, int *found, std::atomic<int> *somebody_found)
{
    if (!s || !found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    if (s->n == 0) { *found = 1; return SS_OK; }
    if (len < s->n) { *found = 0; return SS_OK; }
    hipStream_t small_st = nullptr;
    if (const uint8_t *pinned = small_host_copy(haystack, len, &small_st))
        return ss_search_device(s, pinned, len, small_st, found);           // the call waits for its own kernel: the buffer is free again
    // Chunked staging: chunk k covers haystack bytes [k*C - carry, (k+1)*C) with carry = n-1, so a
    // match straddling a chunk edge is seen by the later chunk.  Two device buffers / two streams:
    // the upload of chunk k+1 overlaps the scan of chunk k.
    const size_t carry = s->n - 1;
    size_t C = (size_t)64 << 20;
    if (C < 4 * s->n) C = 4 * s->n;
    if (C > len) C = len;
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    const size_t nbuf = len > C ? 2 : 1;
    Lease lease(pd->dev);
    if (!lease.set->ensure((int)nbuf, C + carry, false)) return fail(SS_ERR_HIP, "staging allocation failed");
    uint8_t **dbuf = lease.set->d;
    hipStream_t *st = lease.set->st;
    const int k = acquire_slot(s, pd);
    const int epoch = next_epoch(pd, k);                 // "found" value of this call (see ss_search_device)
    int rc = SS_OK;
    int result = 0;
    size_t idx = 0;
    for (size_t off = 0; off < len && rc == SS_OK && !result; off += C, ++idx) {
        if (somebody_found && somebody_found->load(std::memory_order_acquire)) break;   // another device's range holds the needle
        const int b = (int)(idx % nbuf);
        const size_t lead = off == 0 ? 0 : carry;
        const size_t bytes = (len - off < C ? len - off : C) + lead;
        if (bytes < s->n) break;                         // tail shorter than the needle: nothing new can start here
        hipError_t e = hipStreamSynchronize(st[b]);      // buffer b free again
        if (e == hipSuccess && idx >= nbuf &&          // result of the scan that last used this buffer
            __atomic_load_n(pd->h_flags + k, __ATOMIC_ACQUIRE) == epoch) {
            result = 1;
            break;
        }
        if (e == hipSuccess) e = hipMemcpyAsync(dbuf[b], haystack + off - lead, bytes, hipMemcpyHostToDevice, st[b]);
        if (e != hipSuccess) { rc = fail(SS_ERR_HIP, "upload: %s", hipGetErrorString(e)); break; }
        rc = enqueue_scan(s, pd, dbuf[b], bytes, st[b], pd->d_flags + k, false, 0, pd->h_flags + k, epoch);
    }
    for (size_t b = 0; b < nbuf; ++b)
        if (st[b]) (void)hipStreamSynchronize(st[b]);
    if (rc == SS_OK && !result) result = __atomic_load_n(pd->h_flags + k, __ATOMIC_ACQUIRE) == epoch;
    release_slot(s, pd, k);
    if (rc == SS_OK) {
        *found = result;
        if (result && somebody_found) somebody_found->store(1, std::memory_order_release);
    }
    return rc;
}

}  // namespace

extern "C" {

// The literal search_in(&[u8]) (src/x86.rs:523) for a HOST slice over several GPUs: the slice is range-partitioned (n-1 bytes of
// overlap, ss_shard_range) and every device uploads and scans ITS range over ITS OWN PCIe link - one host thread per device,
// each running ss_search_host on its range with that device current - and the booleans are OR-ed on the host.  No
// collective: nothing but the found flag is combined, and all of it happens in this process.  PCIe-bound like
// ss_search_host, G links wide: on an 8-GPU node ~8 x 55 GB/s, above what all cores of the host reach with the reference's
// own AVX2 path (bench.py cpu_baseline: 300-580 GB/s on 2 x EPYC 9575F).  devs == NULL: devices 0 .. ndev-1; a device may be
// listed more than once (its ranges are then uploaded through private staging sets).
int ss_search_host_all(const ss_searcher *s, const uint8_t *haystack, size_t len, int ndev, const int *devs, int *found)
{
    if (!s || !found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    if (s->n == 0) { *found = 1; return SS_OK; }
    if (len < s->n) { *found = 0; return SS_OK; }
    int visible = 0;
    HIP_TRY(hipGetDeviceCount(&visible));
    if (ndev < 1 || ndev > 64) return fail(SS_ERR_ARGUMENT, "1 .. 64 devices");
    for (int g = 0; g < ndev; ++g) {
        const int d = devs ? devs[g] : g;
        if (d < 0 || d >= visible) return fail(SS_ERR_ARGUMENT, "device %d out of range (%d visible)", d, visible);
    }
    SearchGate gate(s);
    std::vector<int> rcs((size_t)ndev, SS_OK), flags((size_t)ndev, 0);
    std::vector<std::string> msgs((size_t)ndev);
    std::vector<std::thread> pool;
    std::atomic<int> somebody_found{0};                  // early exit across the devices: checked by every thread between chunks
    for (int g = 0; g < ndev; ++g)
        pool.emplace_back([&, g]() {
            size_t b = 0, e = 0;
            int rc = ss_shard_range(len, s->n, ndev, g, &b, &e);
            if (rc == SS_OK && hipSetDevice(devs ? devs[g] : g) != hipSuccess) rc = fail(SS_ERR_HIP, "hipSetDevice(%d) failed", devs ? devs[g] : g);
            if (rc == SS_OK && e - b >= s->n) rc = search_host_impl(s, haystack + b, e - b, &flags[(size_t)g], &somebody_found);
            rcs[(size_t)g] = rc;
            if (rc != SS_OK) msgs[(size_t)g] = g_err;          // (the message lives in the worker thread's buffer)
        });
    for (auto &t : pool) t.join();
    int any = 0;
    for (int g = 0; g < ndev; ++g) {
        if (rcs[(size_t)g] != SS_OK) return fail(rcs[(size_t)g], "device %d: %s", devs ? devs[g] : g, msgs[(size_t)g].c_str());
        any |= flags[(size_t)g];
    }
    *found = any;
    return SS_OK;
}

// find() for a host haystack: the chunked upload of ss_search_host with the uint64 best-offset sink.
// Chunks are issued left to right, so once a finished chunk has reported a match no later chunk can
// improve on it: stop issuing, drain the (at most one) chunk still in flight, read the minimum.
int ss_find_host(const ss_searcher *s, const uint8_t *haystack, size_t len, uint64_t *position)
{
    if (!s || !position) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    if (s->n == 0) { *position = 0; return SS_OK; }
    if (len < s->n) { *position = SS_NPOS; return SS_OK; }
    hipStream_t small_st = nullptr;
    if (const uint8_t *pinned = small_host_copy(haystack, len, &small_st)) return ss_find_device(s, pinned, len, small_st, position);
    const size_t carry = s->n - 1;
    size_t C = (size_t)64 << 20;
    if (C < 4 * s->n) C = 4 * s->n;
    if (C > len) C = len;
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    const size_t nbuf = len > C ? 2 : 1;
    Lease lease(pd->dev);
    if (!lease.set->ensure((int)nbuf, C + carry, false)) return fail(SS_ERR_HIP, "staging allocation failed");
    uint8_t **dbuf = lease.set->d;
    hipStream_t *st = lease.set->st;
    const int k = acquire_slot(s, pd);
    int rc = SS_OK;
    size_t idx = 0;
    bool hit = false;
    for (size_t off = 0; off < len && rc == SS_OK && !hit; off += C, ++idx) {
        const int b = (int)(idx % nbuf);
        const size_t lead = off == 0 ? 0 : carry;
        const size_t bytes = (len - off < C ? len - off : C) + lead;
        if (bytes < s->n) break;
        hipError_t e = hipStreamSynchronize(st[b]);
        if (e == hipSuccess && idx >= nbuf) {
            e = hipMemcpy(pd->h_best + k, pd->d_best + k, sizeof(uint64_t), hipMemcpyDeviceToHost);
            if (e == hipSuccess && pd->h_best[k] != SS_NPOS) { hit = true; break; }
        }
        if (e == hipSuccess) e = hipMemcpyAsync(dbuf[b], haystack + off - lead, bytes, hipMemcpyHostToDevice, st[b]);
        if (e != hipSuccess) { rc = fail(SS_ERR_HIP, "upload: %s", hipGetErrorString(e)); break; }
        rc = enqueue_scan(s, pd, dbuf[b], bytes, st[b], pd->d_best + k, true, (uint64_t)(off - lead));
    }
    for (size_t b = 0; b < nbuf; ++b)
        if (st[b]) (void)hipStreamSynchronize(st[b]);
    if (rc == SS_OK) {
        const hipError_t e = hipMemcpy(pd->h_best + k, pd->d_best + k, sizeof(uint64_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(SS_ERR_HIP, "position read-back: %s", hipGetErrorString(e));
        else *position = pd->h_best[k];
    }
    (void)hipMemset(pd->d_best + k, 0xFF, sizeof(uint64_t));       // slots are all-ones whenever they are free
    release_slot(s, pd, k);
    return rc;
}

// ---- row f2: host-file front end (the shape of examples/grep.rs:42-56: open the file, one search_in) ----
// A three-stage pipeline: reader threads pread() the next chunk into a pinned buffer while the previous
// chunks are in flight as hipMemcpyAsync + scan on their own streams.  Chunk k carries the last n-1 bytes
// of chunk k-1 in front, so a match that straddles a chunk edge is seen by the later chunk.
namespace {

bool parallel_pread(int fd, uint8_t *dst, size_t bytes, off_t off, unsigned threads)
{
    if (threads < 1) threads = 1;
    const size_t part = (bytes + threads - 1) / threads;
    std::vector<std::thread> pool;
    std::atomic<bool> ok{true};
    for (unsigned t = 0; t < threads; ++t) {
        const size_t b = (size_t)t * part;
        if (b >= bytes) break;
        const size_t e = b + part < bytes ? b + part : bytes;
        pool.emplace_back([=, &ok]() {
            size_t done = b;
            while (done < e) {
                const ssize_t r = pread(fd, dst + done, e - done, off + (off_t)done);
                if (r <= 0) {
                    ok = false;
                    return;
                }
                done += (size_t)r;
            }
        });
    }
    for (auto &th : pool) th.join();
    return ok;
}

}  // namespace

int ss_search_file(const ss_searcher *s, const char *path, int *found)
{
    if (!s || !path || !found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    SearchGate gate(s);                                  // set_filter* are refused while this call runs
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(SS_ERR_ARGUMENT, "cannot open %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0) {
        close(fd);
        return fail(SS_ERR_ARGUMENT, "cannot stat %s", path);
    }
    const size_t len = (size_t)sb.st_size;
    if (s->n == 0 || len < s->n) {                       // answered without reading the file (x86.rs:500, 357-359)
        close(fd);
        *found = s->n == 0;
        return SS_OK;
    }
    const size_t carry = s->n - 1;
    // chunk: 64 MiB for large files, an eighth of the file (>= 8 MiB) for small ones so that reading, upload
    // and scan of a few-hundred-MiB file still overlap
    size_t C = len / 8;
    if (C > ((size_t)64 << 20)) C = (size_t)64 << 20;
    if (C < ((size_t)8 << 20)) C = (size_t)8 << 20;
    if (C < 4 * s->n) C = 4 * s->n;
    if (C > len) C = len;
    const int nbuf = len > C ? kStageBuf : 1;
    unsigned threads = std::thread::hardware_concurrency();
    if (threads > 8) threads = 8;
    if (const char *e = getenv("SLICESLICE_FILE_THREADS")) {      // tuning aid (tools/host_path_bench.py)
        const long v = atol(e);
        if (v > 0 && v <= 256) threads = (unsigned)v;
    }
    if (len < ((size_t)8 << 20)) threads = 1;

    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) {
        close(fd);
        return rc;
    }
    // the device's cached staging set when it is free, a private one otherwise
    Lease lease(pd->dev);
    Staging *fs = lease.set;
    if (!fs->ensure(nbuf, C + carry, true)) {
        close(fd);
        return fail(SS_ERR_HIP, "staging allocation failed");
    }
    uint8_t **hbuf = fs->h, **dbuf = fs->d;
    hipStream_t *st = fs->st;
    const int k = acquire_slot(s, pd);
    const int epoch = next_epoch(pd, k);
    int rc = SS_OK;
    int result = 0;
    size_t idx = 0, prev_total = 0;
    int prev_b = -1;
    for (size_t off = 0; off < len && rc == SS_OK && !result; off += C, ++idx) {
        const int b = (int)(idx % (size_t)nbuf);
        const size_t lead = off == 0 ? 0 : carry;
        const size_t fresh = len - off < C ? len - off : C;
        hipError_t e = hipStreamSynchronize(st[b]);      // the copy + scan that last used buffer b are done
        if (e != hipSuccess) { rc = fail(SS_ERR_HIP, "stream wait: %s", hipGetErrorString(e)); break; }
        if (__atomic_load_n(pd->h_flags + k, __ATOMIC_ACQUIRE) == epoch) { result = 1; break; }
        if (lead) memcpy(hbuf[b], hbuf[prev_b] + prev_total - carry, carry);
        if (!parallel_pread(fd, hbuf[b] + lead, fresh, (off_t)off, threads)) {
            rc = fail(SS_ERR_ARGUMENT, "read error on %s", path);
            break;
        }
        const size_t total = lead + fresh;
        prev_b = b;
        prev_total = total;
        if (total < s->n) break;                         // tail shorter than the needle: nothing new can start here
        e = hipMemcpyAsync(dbuf[b], hbuf[b], total, hipMemcpyHostToDevice, st[b]);
        if (e != hipSuccess) { rc = fail(SS_ERR_HIP, "upload: %s", hipGetErrorString(e)); break; }
        rc = enqueue_scan(s, pd, dbuf[b], total, st[b], pd->d_flags + k, false, 0, pd->h_flags + k, epoch);
    }
    for (int b = 0; b < fs->nbuf; ++b)
        if (st[b]) (void)hipStreamSynchronize(st[b]);
    if (rc == SS_OK && !result) result = __atomic_load_n(pd->h_flags + k, __ATOMIC_ACQUIRE) == epoch;
    close(fd);
    release_slot(s, pd, k);
    if (rc == SS_OK) *found = result;
    return rc;
}

// ---- row f3: data for a `position` policy ---------------------------------------------------------------
int ss_byte_histogram_device(const void *d_haystack, size_t len, size_t sample_bytes, void *hip_stream,
                             uint64_t hist[256])
{
    if (!hist) return fail(SS_ERR_ARGUMENT, "hist is NULL");
    memset(hist, 0, 256 * sizeof(uint64_t));
    if (len < 16) return SS_OK;
    if (!d_haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    DeviceInfo di;
    if (int rc = device_info(dev, &di)) return rc;
    unsigned long long *d_hist = nullptr;
    HIP_TRY(hipMalloc((void **)&d_hist, 256 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_hist, 0, 256 * sizeof(unsigned long long), st);
    uint64_t stride = 1;
    if (sample_bytes && sample_bytes < len) stride = (len + sample_bytes - 1) / sample_bytes;
    const uint64_t work = len / 16 / stride;
    uint64_t blocks = (work + ss::kBlock - 1) / ss::kBlock;
    if (blocks > (uint64_t)di.cus * 8) blocks = (uint64_t)di.cus * 8;
    if (blocks < 1) blocks = 1;
    if (e == hipSuccess) {
        ss::byte_histogram_kernel<<<dim3((unsigned)blocks), dim3(ss::kBlock), 0, st>>>(
            static_cast<const uint8_t *>(d_haystack), len, stride, d_hist);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hist, d_hist, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_hist);
    if (e != hipSuccess) return fail(SS_ERR_HIP, "histogram: %s", hipGetErrorString(e));
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

// ss_search_batched, planned form: tuning constants and the one-time set-up of the stream-ordered allocator.
// Total workgroups aimed at, per CU.  Measured in one process on one buffer (tools/batch_tune.py, profiles/r03/batch_tune_*.jsonl;
// kernel time: tools/shape_trace.py under rocprofv3): 96 per CU is best or within 1 % of the best on every shape at 1 GiB in
// total (1 / 64 / 256 / 1,024 problems: 149-151 us = 7.1-7.2 TB/s of kernel time; 160 per CU 150-153 us, 256 per CU 161-163 us,
// 512 per CU 190 us - surplus workgroups cost 0.3-0.4 us of a slot each) and for the i386 loop (0.151 ms; 160: 0.17, 256: 0.77);
// at 4 GiB in 4,096 problems 256 per CU is 2 % faster (583 vs 597 us), which is not worth the rest.
constexpr unsigned kPlanWgsPerCu = 96;
constexpr uint32_t kPlanMinTiles = 2;       // shortest slice worth a workgroup, in 16 KiB tiles

// Descriptor scratch of the planned form: one grow-only device buffer per (device, stream), kept for the life of the process.
// Launches on one stream execute in order, so a buffer that belongs to the stream can be reused by the next call on that
// stream without any wait; the entry's mutex keeps the two launches of one call adjacent when several threads share a
// stream.  (hipMallocAsync / hipFreeAsync per call did the same job at 5-10 us of extra latency per call.)  At most
// kPlanScratchEntries streams per device are remembered; beyond that the least recently used entry is freed (hipFree waits
// for the device, so nothing that still reads the buffer can be running).
constexpr int kPlanScratchEntries = 32;
struct PlanScratch {
    hipStream_t stream = nullptr;
    bool used = false;
    ss::BatchDesc *buf = nullptr;
    size_t cap = 0;             // descriptors
    uint64_t stamp = 0;
    std::mutex mu;              // held across the plan + scan launches of one call
};
static std::mutex g_plan_mu[kMaxDevices];
static PlanScratch g_plan[kMaxDevices][kPlanScratchEntries];
static uint64_t g_plan_clock[kMaxDevices];

// Returns the stream's entry with its mutex LOCKED and room for `count` descriptors, or nullptr (no memory).
static PlanScratch *plan_scratch_acquire(int dev, hipStream_t st, size_t count)
{
    if (dev < 0 || dev >= kMaxDevices) return nullptr;
    std::unique_lock<std::mutex> table(g_plan_mu[dev]);
    PlanScratch *e = nullptr, *victim = nullptr;
    for (auto &c : g_plan[dev]) {
        if (c.used && c.stream == st) {
            e = &c;
            break;
        }
        if (!victim || (!c.used && victim->used) || (c.used == victim->used && c.stamp < victim->stamp)) victim = &c;
    }
    if (e) {
        e->mu.lock();            // another thread's call on this stream is between its two launches: brief
    } else {
        e = victim;              // an unused entry, else the least recently used one (if a call is between its two launches
        e->mu.lock();            // on it right now: wait for that - microseconds)
        if (e->buf) (void)hipFree(e->buf);                      // hipFree waits for the device: nobody reads it any more
        e->buf = nullptr;
        e->cap = 0;
        e->stream = st;
        e->used = true;
    }
    e->stamp = ++g_plan_clock[dev];
    table.unlock();
    if (e->cap < count) {
        const size_t want = count < 4096 ? 4096 : count + count / 2;
        if (e->buf) (void)hipFree(e->buf);
        e->buf = nullptr;
        e->cap = 0;
        if (hipMalloc((void **)&e->buf, want * sizeof(ss::BatchDesc)) != hipSuccess) {
            (void)hipGetLastError();
            e->buf = nullptr;
            e->mu.unlock();
            return nullptr;
        }
        e->cap = want;
    }
    return e;
}

// The batched kernels (73-80 VGPRs) are held to four workgroups per CU like the single-problem scan on random bytes (see
// pick_variant); SLICESLICE_BATCH_OCC overrides (tuning aid).  Their needle slices are static LDS, hence the "fixed" part.
static uint32_t batch_lds_pad()
{
    int occ = 4;
    if (const char *e = getenv("SLICESLICE_BATCH_OCC")) { const int v = atoi(e); if (v >= 1 && v <= 8) occ = v; }
    return occupancy_pad(occ, ss::kBlock);
}

static int fill_batch_args(ss::BatchArgs *a, const void *d_haystacks, const uint64_t *d_hay_begin,
                           const uint64_t *d_hay_end, const void *d_needles, const uint64_t *d_needle_begin,
                           const uint64_t *d_needle_end, const uint64_t *d_position, int *d_found)
{
    if (!d_hay_begin || !d_hay_end || !d_needle_begin || !d_needle_end) return fail(SS_ERR_ARGUMENT, "NULL argument");
    a->haystacks = static_cast<const uint8_t *>(d_haystacks);
    a->hay_begin = d_hay_begin;
    a->hay_end = d_hay_end;
    a->needles = static_cast<const uint8_t *>(d_needles);
    a->needle_begin = d_needle_begin;
    a->needle_end = d_needle_end;
    a->position = d_position;
    a->found = d_found;
    a->best = nullptr;
    return SS_OK;
}

static int launch_batched(const ss::BatchArgs &a, size_t count, hipStream_t st);

int ss_search_batched(const void *d_haystacks, const uint64_t *d_hay_begin, const uint64_t *d_hay_end,
                      const void *d_needles, const uint64_t *d_needle_begin, const uint64_t *d_needle_end,
                      const uint64_t *d_position, size_t count, void *hip_stream, int *d_found)
{
    if (count == 0) return SS_OK;
    if (!d_found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    ss::BatchArgs a;
    if (int rc = fill_batch_args(&a, d_haystacks, d_hay_begin, d_hay_end, d_needles, d_needle_begin, d_needle_end,
                                 d_position, d_found))
        return rc;
    return launch_batched(a, count, static_cast<hipStream_t>(hip_stream));
}

/* Row f1 for many problems: the leftmost offset per problem (SS_NPOS: absent), the `Option<usize>` shape of
 * bench/sse4-strstr/src/lib.rs:4-15 for a whole batch - same plan kernel, same scan grid, FIND instantiation. */
int ss_find_batched(const void *d_haystacks, const uint64_t *d_hay_begin, const uint64_t *d_hay_end, const void *d_needles,
                    const uint64_t *d_needle_begin, const uint64_t *d_needle_end, size_t count, void *hip_stream, uint64_t *d_position)
{
    if (count == 0) return SS_OK;
    if (!d_position) return fail(SS_ERR_ARGUMENT, "NULL argument");
    ss::BatchArgs a;
    if (int rc = fill_batch_args(&a, d_haystacks, d_hay_begin, d_hay_end, d_needles, d_needle_begin, d_needle_end, nullptr, nullptr))
        return rc;
    a.best = d_position;
    return launch_batched(a, count, static_cast<hipStream_t>(hip_stream));
}

static int launch_batched(const ss::BatchArgs &a, size_t count, hipStream_t st)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    DeviceInfo di;
    if (int rc = device_info(dev, &di)) return rc;
    if (count > 0x3fffffffull) return fail(SS_ERR_ARGUMENT, "too many problems");
    // Planned form (the default; SLICESLICE_BATCH_PLAN=0 keeps the single-kernel form): batch_plan_kernel turns the range
    // arrays into one 64-byte descriptor per problem and writes the initial flags (no memset launch), the scan grid's
    // workgroups then start with one scalar load.  The descriptors live in per-stream scratch (plan_scratch_acquire); without it
    // the single-kernel form runs instead.
    const char *pm = getenv("SLICESLICE_BATCH_PLAN");          // read per call: tools A/B the two forms in one process
    if (!pm || pm[0] != '0') {
        // The haystack lengths live on the device, so the grid is still sized from the problem COUNT: kPlanWgsPerCu
        // workgroups per CU in total, at most 65,535 slices per problem (gridDim.y).  What changed is the price of being
        // wrong: a surplus slice costs one scalar round trip, and a slice as short as kPlanMinTiles tiles is worth a
        // workgroup because nothing but that load stands in front of its first haystack byte.
        uint64_t wg_target = (uint64_t)di.cus * kPlanWgsPerCu;
        uint32_t min_tiles = kPlanMinTiles;
        if (const char *e = getenv("SLICESLICE_BATCH_WGS")) { const long v = atol(e); if (v > 0) wg_target = (uint64_t)v; }
        if (const char *e = getenv("SLICESLICE_BATCH_MIN_TILES")) { const long v = atol(e); if (v > 0) min_tiles = (uint32_t)v; }
        uint64_t slices = (wg_target + count - 1) / count;
        if (slices < 1) slices = 1;
        while (slices > 1 && (uint64_t)count * slices > 0x7fffffffull) --slices;   // gridDim.x
        if (PlanScratch *ps = plan_scratch_acquire(dev, st, count)) {
            ss::BatchDesc *descs = ps->buf;
            const uint64_t pblocks = ((uint64_t)count + ss::kBlock - 1) / ss::kBlock;
            ss::batch_plan_kernel<<<dim3((unsigned)pblocks), dim3(ss::kBlock), 0, st>>>(a, (uint64_t)count, descs, (uint32_t)slices,
                                                                                      min_tiles, ss::kWavesPerBlock * 4);
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) {
                const dim3 grid((unsigned)((uint64_t)count * slices));
                if (a.best)
                    ss::scan_batched_plan_kernel<4, true><<<grid, dim3(ss::kBlock), batch_lds_pad(), st>>>(a, descs, (uint32_t)count, (uint32_t)slices);
                else
                    ss::scan_batched_plan_kernel<4, false><<<grid, dim3(ss::kBlock), batch_lds_pad(), st>>>(a, descs, (uint32_t)count, (uint32_t)slices);
                e = hipGetLastError();
            }
            ps->mu.unlock();
            if (e != hipSuccess) return fail(SS_ERR_HIP, "batched launch: %s", hipGetErrorString(e));
            return SS_OK;
        }
#ifndef SS_TUNING_VARIANTS
        return fail(SS_ERR_NOMEM, "no device memory for %zu problem descriptors", count);
#endif
    }
#ifndef SS_TUNING_VARIANTS
    return fail(SS_ERR_ARGUMENT, "SLICESLICE_BATCH_PLAN=0 selects the single-kernel form, which is part of the tuning build only");
#else
    if (a.best) return fail(SS_ERR_ARGUMENT, "ss_find_batched has the planned form only");
    HIP_TRY(hipMemsetAsync(a.found, 0, count * sizeof(int), st));
    // Single-kernel form (tuning build: the reference point of tools/batch_tune.py).  The haystack lengths live on the device,
    // so the grid is chosen from the problem count alone: 96 workgroups per CU in total; every workgroup rebuilds its problem's
    // descriptor from the range arrays (a chain of three dependent round trips in front of its first haystack load).
    uint64_t wg_target = (uint64_t)di.cus * 96;
    if (const char *e = getenv("SLICESLICE_BATCH_WGS")) {
        const long v = atol(e);
        if (v > 0) wg_target = (uint64_t)v;
    }
    uint64_t slices = (wg_target + count - 1) / count;
    if (slices < 1) slices = 1;
    if (slices > 4096) slices = 4096;
    ss::scan_batched_kernel<4><<<dim3((unsigned)count, (unsigned)slices), dim3(ss::kBlock), batch_lds_pad(), st>>>(a);
    HIP_TRY(hipGetLastError());
    return SS_OK;
#endif
}

int ss_search_pairs(const void *d_haystacks, const uint64_t *d_hay_begin, const uint64_t *d_hay_end,
                    const void *d_needles, const uint64_t *d_needle_begin, const uint64_t *d_needle_end,
                    const uint64_t *d_position, size_t count, void *hip_stream, int *d_found)
{
    if (count == 0) return SS_OK;
    if (!d_found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    ss::BatchArgs a;
    if (int rc = fill_batch_args(&a, d_haystacks, d_hay_begin, d_hay_end, d_needles, d_needle_begin, d_needle_end,
                                 d_position, d_found))
        return rc;
    const uint64_t blocks = ((uint64_t)count + ss::kBlock - 1) / ss::kBlock;
    if (blocks > 0x7fffffffull) return fail(SS_ERR_ARGUMENT, "too many problems");
    ss::scan_pairs_kernel<<<dim3((unsigned)blocks), dim3(ss::kBlock), 0, static_cast<hipStream_t>(hip_stream)>>>(a, count);
    HIP_TRY(hipGetLastError());
    return SS_OK;
}

int ss_fill_random_device(void *d_dst, uint64_t global_offset, size_t len, uint64_t seed, void *hip_stream)
{
    if (len == 0) return SS_OK;
    if (!d_dst) return fail(SS_ERR_ARGUMENT, "dst is NULL");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    DeviceInfo di;
    if (int rc = device_info(dev, &di)) return rc;
    uint64_t words = (len + 15) / 8;
    uint64_t blocks = (words + ss::kBlock - 1) / ss::kBlock;
    if (blocks > (uint64_t)di.cus * 16) blocks = (uint64_t)di.cus * 16;
    ss::fill_random_kernel<<<dim3((unsigned)blocks), dim3(ss::kBlock), 0, static_cast<hipStream_t>(hip_stream)>>>(
        static_cast<uint8_t *>(d_dst), global_offset, len, seed);
    HIP_TRY(hipGetLastError());
    return SS_OK;
}

int ss_fill_random_host(uint8_t *dst, uint64_t global_offset, size_t len, uint64_t seed)
{
    if (len && !dst) return fail(SS_ERR_ARGUMENT, "dst is NULL");
    size_t k = 0;
    while (k < len) {
        const uint64_t i = global_offset + k;
        uint64_t v = ss::synth_word(seed, i >> 3) >> (8 * (i & 7));
        size_t take = 8 - (size_t)(i & 7);
        if (take > len - k) take = len - k;
        for (size_t j = 0; j < take; ++j, v >>= 8) dst[k + j] = (uint8_t)v;
        k += take;
    }
    return SS_OK;
}

int ss_read_ceiling(const void *d_src, size_t len, void *hip_stream, int reps, float *ms_per_rep)
{
    if (!d_src || !ms_per_rep || reps < 1) return fail(SS_ERR_ARGUMENT, "bad argument");
    if (((uintptr_t)d_src & 15) != 0) return fail(SS_ERR_ARGUMENT, "source must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    uint32_t *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    constexpr int U = 4;
    // A handful of launch shapes (bytes per lane, tiles per workgroup, workgroups per CU through unused LDS);
    // the fastest is reported.  The 16-byte / two-tile shape is the scan's own.
    struct Shape { int lane_bytes; uint64_t tpb; uint32_t lds; };
    const Shape shapes[] = {{16, 2, 0}, {16, 2, 32 << 10}, {16, 1, 32 << 10}, {8, 2, 0}, {8, 2, 32 << 10}, {8, 1, 32 << 10}};
    auto launch = [&](const Shape &sh) {
        const uint64_t ntiles = (len / 1024) / (ss::kWavesPerBlock * U);
        uint64_t blocks = (ntiles + sh.tpb - 1) / sh.tpb;
        if (blocks < 1) blocks = 1;
        const dim3 grid((unsigned)blocks);
        if (sh.lane_bytes == 16)
            ss::read_ceiling_kernel<U, ss::u32x4><<<grid, dim3(ss::kBlock), sh.lds, st>>>(static_cast<const ss::u32x4 *>(d_src), len / 16, sink, sh.tpb);
        else
            ss::read_ceiling_kernel<U, ss::u32x2><<<grid, dim3(ss::kBlock), sh.lds, st>>>(static_cast<const ss::u32x2 *>(d_src), len / 8, sink, sh.tpb);
    };
    auto run = [&]() -> hipError_t {
        hipError_t e;
        if ((e = hipMalloc((void **)&sink, 64)) != hipSuccess) return e;
        if ((e = hipEventCreate(&e0)) != hipSuccess) return e;
        if ((e = hipEventCreate(&e1)) != hipSuccess) return e;
        float best = 0;
        for (const Shape &sh : shapes) {
            launch(sh);                                             // warm-up
            if ((e = hipEventRecord(e0, st)) != hipSuccess) return e;
            for (int r = 0; r < reps; ++r) launch(sh);
            if ((e = hipEventRecord(e1, st)) != hipSuccess) return e;
            if ((e = hipEventSynchronize(e1)) != hipSuccess) return e;
            if ((e = hipGetLastError()) != hipSuccess) return e;
            float t = 0;
            if ((e = hipEventElapsedTime(&t, e0, e1)) != hipSuccess) return e;
            if (getenv("SLICESLICE_CEILING_VERBOSE"))
                fprintf(stderr, "read ceiling: %2d B/lane, %llu tile(s)/workgroup, %2u KiB LDS pad: %.1f GB/s\n", sh.lane_bytes,
                        (unsigned long long)sh.tpb, sh.lds >> 10, (double)len * reps / (t * 1e6));
            if (best == 0 || t < best) best = t;
        }
        ms = best;
        return hipSuccess;
    };
    const hipError_t e = run();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return fail(SS_ERR_HIP, "read ceiling: %s", hipGetErrorString(e));
    *ms_per_rep = ms / (float)reps;
    return SS_OK;
}

int ss_shard_range(size_t len, size_t needle_len, int nranks, int rank, size_t *begin, size_t *end)
{
    if (!begin || !end || nranks < 1 || rank < 0 || rank >= nranks) return fail(SS_ERR_ARGUMENT, "bad shard arguments");
    const size_t S = (len + (size_t)nranks - 1) / (size_t)nranks;
    size_t b = (size_t)rank * S;
    if (b > len) b = len;
    const size_t overlap = needle_len ? needle_len - 1 : 0;
    size_t e = len - b <= S || len - b - S <= overlap ? len : b + S + overlap;
    *begin = b;
    *end = e;
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

// What a resident search service (a kernel that stays on the device and takes requests from a pinned mailbox instead of
// being launched per search) would pay per request BEFORE it looks at a haystack byte: the host posts request i to pinned
// memory, one device lane sees it and answers to pinned memory, the host sees the answer.  Median microseconds over `iters`
// round trips.  Measurement only (INTEGRATION.md section 6 sets it against the launch path's per-call time); all waits on
// both sides are bounded.
int ss_mailbox_round_trip_us(int iters, double *median_us, double *min_us)
{
    if (iters < 1 || iters > 1000000 || !median_us) return fail(SS_ERR_ARGUMENT, "bad argument");
    unsigned long long *box = nullptr;
    HIP_TRY(hipHostMalloc((void **)&box, 2 * 64, hipHostMallocPortable | hipHostMallocMapped));
    volatile unsigned long long *req = box, *resp = box + 8;       // separate cache lines
    *req = 0;
    *resp = 0;
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipHostFree(box); return fail(SS_ERR_HIP, "stream: %s", hipGetErrorString(e)); }
    // ~2 s at 100 MHz-class tick rates: far beyond any round trip, far below a test's patience
    ss::mailbox_echo_kernel<<<1, 1, 0, st>>>(box, box + 8, (unsigned)iters, 200000000ull);
    e = hipGetLastError();
    std::vector<double> us;
    us.reserve((size_t)iters);
    bool ok = e == hipSuccess;
    std::this_thread::sleep_for(std::chrono::milliseconds(2));     // the kernel is up and polling
    for (int i = 1; ok && i <= iters; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        __atomic_store_n((unsigned long long *)req, (unsigned long long)i, __ATOMIC_RELEASE);
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n((unsigned long long *)resp, __ATOMIC_ACQUIRE) >= (unsigned long long)i) break;
            cpu_relax();
            if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(500)) { ok = false; break; }
        }
        us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    __atomic_store_n((unsigned long long *)req, ~0ull, __ATOMIC_RELEASE);   // lets a kernel that is still waiting run through
    (void)hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
    (void)hipHostFree(box);
    if (!ok || us.empty()) return fail(SS_ERR_HIP, "mailbox round trip: no answer from the device");
    std::sort(us.begin(), us.end());
    *median_us = us[us.size() / 2];
    if (min_us) *min_us = us.front();
    return SS_OK;
}

// ---- resident search service ------------------------------------------------------------------------------------
// Host side of ss::service_kernel (scan_kernels.hpp has the protocol).  One request at a time per service (a mutex); the
// kernel is (re)started on demand - at the first request, and after every lease that ran out.

}  // extern "C"

struct ss_service {
    int dev = 0;
    int workgroups = 0;
    unsigned long long idle_ticks = 0;
    hipStream_t stream = nullptr;
    uint32_t *h_box = nullptr;              // pinned, 2 lines of 64 bytes: status | answer (written by the device)
    uint8_t *d_mem = nullptr;               // device: mailbox (256 B, written by the HOST through the BAR) | stop word | done counter | found flag
    uint32_t seq = 0;                       // last request posted
    volatile uint32_t *hdp_flush = nullptr; // the device's HDP flush register: pushes the mailbox writes out of the host data path
    volatile uint32_t *hdp_reg = nullptr;   // the same register, whatever SLICESLICE_SERVICE_HDP_FLUSH says (set-up writes)
    uint32_t done_low = 0, done_hi = 0;     // the never-reset completion counter, as the host knows it
    uint64_t requests = 0, launches = 0, settled_requests = 0;
    // ss_service_bind: a device range the caller vouches for (unchanged until unbound), `bound_settled` once a request has
    // acquired it; `settled_ticket`: needles uploaded up to this ticket were in memory before the latest acquire
    const uint8_t *bound_lo = nullptr, *bound_hi = nullptr;
    bool bound_settled = false;
    uint64_t settled_ticket = 0;
    std::mutex mu;
    volatile uint32_t *status() const { return h_box; }
    volatile unsigned long long *answer() const { return reinterpret_cast<volatile unsigned long long *>(h_box + 16); }
    volatile uint32_t *mailbox() const { return reinterpret_cast<volatile uint32_t *>(d_mem); }   // the host's view = the device's address
    uint32_t *d_stop() const { return reinterpret_cast<uint32_t *>(d_mem + 256); }
    unsigned long long *d_done() const { return reinterpret_cast<unsigned long long *>(d_mem + 320); }
    int *d_found() const { return reinterpret_cast<int *>(d_mem + 384); }
};

namespace {

constexpr int kServiceDefaultWorkgroups = 64;
constexpr double kServiceDefaultLeaseMs = 20.0;
std::atomic<ss_service *> g_default_service[kMaxDevices];

// The request, into the mailbox in device memory: payload first, with zero where the sequence number goes (16-byte stores: a
// write-combining mapping merges them into line writes, an uncached one sends each as it is), a store fence, then the four
// sequence dwords - posted writes reach the device in order, so a line that shows a number holds that request's payload.
void service_write_mailbox(ss_service *sv, const ss::ServiceRequest &rq, uint32_t seq)
{
    alignas(16) uint32_t img[64];
    uint32_t payload[60] = {0};
    memcpy(payload, &rq, sizeof rq);
    for (int line = 0; line < 4; ++line) {
        for (int j = 0; j < 15; ++j) img[line * 16 + j] = payload[line * 15 + j];
        img[line * 16 + 15] = 0;                        // no request's number: a line in this state is nobody's
    }
    volatile uint32_t *m = sv->mailbox();
    static const bool dbg = getenv("SLICESLICE_SERVICE_DEBUG") != nullptr;
    const auto w0 = dbg ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    for (int k = 0; k < 16; ++k)
        _mm_store_si128(reinterpret_cast<__m128i *>(const_cast<uint32_t *>(m)) + k, _mm_load_si128(reinterpret_cast<const __m128i *>(img) + k));
    _mm_sfence();
    for (int line = 0; line < 4; ++line) m[line * 16 + 15] = seq;
    _mm_sfence();
    if (sv->hdp_flush) __atomic_store_n(sv->hdp_flush, 1u, __ATOMIC_RELAXED);   // (no read-back: the kernel polls, nothing is ordered behind this)
    if (dbg) {
        static double total_us = 0;
        static unsigned long n = 0;
        total_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        if ((++n & 0x3FFF) == 0) fprintf(stderr, "[service] mailbox write: %.3f us average over %lu requests\n", total_us / n, n);
    }
}

// Mailbox, stop word, counter and flag start as zeros - written by the CPU through the BAR and waited for, like everything
// else the host puts there: a hipMemset is asynchronous to the host and would be free to run AFTER the first request has
// been written into the mailbox (it did, now and then: the kernel never saw that request, left when its lease was over, and
// the request was answered by a second residency).
void service_zero_device_memory(ss_service *sv)
{
    alignas(16) static const uint8_t zeros[512] = {0};
    bar_write(sv->d_mem, zeros, sizeof zeros, sv->hdp_reg);
}

int service_launch(ss_service *sv, uint32_t first_seq)
{
    __atomic_store_n(sv->status(), 0u, __ATOMIC_RELAXED);
    HIP_TRY(hipMemsetAsync(sv->d_stop(), 0, sizeof(uint32_t), sv->stream));   // (ordered behind the previous residency's end)
    ss::service_kernel<4><<<dim3((unsigned)sv->workgroups), dim3(ss::kBlock), 0, sv->stream>>>(
        const_cast<const uint32_t *>(reinterpret_cast<uint32_t *>(sv->d_mem)), const_cast<uint32_t *>(sv->status()),
        const_cast<unsigned long long *>(sv->answer()), sv->d_stop(), sv->d_done(), sv->d_found(), first_seq, sv->idle_ticks);
    HIP_TRY(hipGetLastError());
    ++sv->launches;
    return SS_OK;
}

// A residency has ended (lease, or never begun) with request `seq` unanswered: some of its waves may have taken the request
// before they saw the stop word and counted themselves out - a count that can no longer complete.  Wait for the kernel to be
// gone, start the counter over, name the new target in the request, post it again and start a new residency with it.
int service_restart_with(ss_service *sv, ss::ServiceRequest &rq, uint32_t seq)
{
    HIP_TRY(hipStreamSynchronize(sv->stream));
    HIP_TRY(hipMemsetAsync(sv->d_done(), 0, sizeof(unsigned long long), sv->stream));
    sv->done_low = sv->done_hi = 0;
    if (!rq.stop) {
        rq.pr.done_target = rq.active == 1 ? 0u : rq.active;
        rq.pr.done_hi = 0;
        rq.settled = 0;                                 // (a new kernel starts with clean caches anyway)
    }
    service_write_mailbox(sv, rq, seq);
    return service_launch(sv, seq);
}

// Posts one request and waits for its answer word (or, for a stop request, for the kernel to say it has left).
int service_post(ss_service *sv, ss::ServiceRequest &rq, uint32_t seq, unsigned long long *answer)
{
    service_write_mailbox(sv, rq, seq);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);                              // the request first, THEN the kernel's state (see service_kernel)
    uint32_t st = __atomic_load_n(sv->status(), __ATOMIC_ACQUIRE);
    if (rq.stop && (st == 0 || st == ss::kSvcExited)) return SS_OK;       // not resident: nothing to stop
    bool launched_now = false;
    if (st == 0 && sv->launches == 0) {                                   // first request of this service
        if (int rc = service_restart_with(sv, rq, seq)) return rc;
        launched_now = true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (!rq.stop) {
            const unsigned long long a = __atomic_load_n(sv->answer(), __ATOMIC_ACQUIRE);
            if ((uint32_t)((a >> 1) & 0x7FFFFFFFu) == seq) {
                *answer = a;
                return SS_OK;
            }
        }
        st = __atomic_load_n(sv->status(), __ATOMIC_ACQUIRE);
        if (st == ss::kSvcExited) {
            if (rq.stop) return SS_OK;
            // the lease ran out before (all of) the kernel saw this request: a new residency starts with it
            if (int rc = service_restart_with(sv, rq, seq)) return rc;
            launched_now = true;
        }
        cpu_relax();
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(launched_now ? 20 : 10))
            return fail(SS_ERR_HIP, "search service: no answer to request %u (kernel state %u)", seq, st);
    }
}

void service_free(ss_service *sv)
{
    if (sv->stream) {
        (void)hipStreamSynchronize(sv->stream);
        (void)hipStreamDestroy(sv->stream);
    }
    (void)hipHostFree(sv->h_box);
    (void)hipFree(sv->d_mem);
    delete sv;
}

}  // namespace

extern "C" {

int ss_service_start(int workgroups, double lease_ms, ss_service **out)
{
    if (!out) return fail(SS_ERR_ARGUMENT, "out is NULL");
    *out = nullptr;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    DeviceInfo di;
    if (int rc = device_info(dev, &di)) return rc;
    if (!di.gfx950) return fail(SS_ERR_NO_DEVICE, "HIP device %d is not a gfx950 (MI355X-class) device", dev);
    if (workgroups == 0) workgroups = kServiceDefaultWorkgroups;
    if (workgroups < 1 || workgroups > di.cus) return fail(SS_ERR_ARGUMENT, "1 .. %d service workgroups (one per compute unit at most)", di.cus);
    if (lease_ms == 0) lease_ms = kServiceDefaultLeaseMs;
    if (!(lease_ms >= 0.05 && lease_ms <= 10000.0)) return fail(SS_ERR_ARGUMENT, "lease of 0.05 .. 10000 ms");
    // the mailbox lives in device memory and is written by the CPU: every byte of an MI300-class part's memory is behind its
    // PCIe BAR; a platform that hides it cannot run the service (searches take the launch path, as ever)
    int large_bar = 0;
    HIP_TRY(hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev));
    if (!large_bar) return fail(SS_ERR_NO_DEVICE, "device %d does not expose its memory to the CPU (no large BAR): no search service", dev);
    ss_service *sv = new (std::nothrow) ss_service;
    if (!sv) return fail(SS_ERR_NOMEM, "out of memory");
    sv->dev = dev;
    sv->workgroups = workgroups;
    {
        static const bool flush = []() { const char *v = getenv("SLICESLICE_SERVICE_HDP_FLUSH"); return !(v && v[0] == '0'); }();
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            sv->hdp_reg = prop.hdpMemFlushCntl;
            if (flush) sv->hdp_flush = prop.hdpMemFlushCntl;
        } else {
            (void)hipGetLastError();
        }
    }
    sv->idle_ticks = (unsigned long long)(lease_ms * 1e5);             // s_memrealtime: 100 MHz
    hipError_t e = hipStreamCreateWithFlags(&sv->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void **)&sv->h_box, 6 * 64, hipHostMallocPortable | hipHostMallocMapped);
    if (e == hipSuccess) memset(sv->h_box, 0, 6 * 64);
    if (e == hipSuccess) e = hipMalloc((void **)&sv->d_mem, 512);
    if (e == hipSuccess) service_zero_device_memory(sv);
    if (e != hipSuccess) {
        service_free(sv);
        return fail(SS_ERR_HIP, "search service set-up: %s", hipGetErrorString(e));
    }
    *out = sv;
    return SS_OK;
}

int ss_service_search(ss_service *sv, const ss_searcher *s, const void *d_haystack, size_t len, int *found)
{
    if (!sv || !s || !found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (len && !d_haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    if (s->n == 0) { *found = 1; return SS_OK; }        // x86.rs:500
    if (len < s->n) { *found = 0; return SS_OK; }       // x86.rs:357-359
    static const bool dbg = getenv("SLICESLICE_SERVICE_DEBUG") != nullptr;
    const auto c0 = dbg ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != sv->dev) return fail(SS_ERR_ARGUMENT, "the service runs on device %d, the current device is %d", sv->dev, dev);
    SearchGate gate(s);
    PerDevice *pd = nullptr;
    if (int rc = get_per_device(s, &pd)) return rc;
    ss::ServiceRequest rq;
    memset(&rq, 0, sizeof rq);
    ProblemShape ps;
    fill_problem(s, pd->d_needle, d_haystack, len, 0, &rq.pr, &ps);
    if (rq.pr.d != 0) return fail(SS_ERR_ARGUMENT, "the service runs the single-stream kernels: filter pairs 16 or more apart take the launch path");
    rq.q = (uint32_t)((ps.position % 16) / 4);
    rq.one_byte = ps.one_byte ? 1u : 0u;
    std::lock_guard<std::mutex> lock(sv->mu);
    if (sv->seq >= 0x7FFFFF00u || sv->done_low > kDoneLowMax) {
        // sequence numbers (31 bits in the answer word) or the workgroup count about to run out: a fresh start
        ss::ServiceRequest bye;
        memset(&bye, 0, sizeof bye);
        bye.stop = 1;
        unsigned long long ignored = 0;
        if (int rc = service_post(sv, bye, ++sv->seq, &ignored)) return rc;
        HIP_TRY(hipStreamSynchronize(sv->stream));
        service_zero_device_memory(sv);
        memset(sv->h_box, 0, 6 * 64);
        sv->seq = sv->done_low = sv->done_hi = 0;
        sv->launches = 0;
    }
    const uint32_t seq = ++sv->seq;
    rq.pr.epoch = (int)seq;
    rq.pr.flags = ss::kProblemCounted;
    // one workgroup per tile at most: the count-out of a 1 KiB search is one atomic, not sixty-four
    const uint64_t tiles = (rq.pr.npieces + ss::kWavesPerBlock * 4 - 1) / (ss::kWavesPerBlock * 4);
    rq.active = (uint32_t)std::min<uint64_t>((uint64_t)sv->workgroups, std::max<uint64_t>(tiles, 1));
    rq.pr.done_target = sv->done_low + (rq.active == 1 ? 0u : rq.active);     // a single workgroup answers without the counter
    rq.pr.done_hi = sv->done_hi;
    // Inside a bound range that an earlier request has acquired, with a needle that was in device memory by then: nothing this
    // request reads has changed, the workgroups skip their acquire (2 us of a request's 8).
    const uint8_t *lo = static_cast<const uint8_t *>(d_haystack);
    const bool in_bound = sv->bound_lo && lo >= sv->bound_lo && lo + len <= sv->bound_hi;
    rq.settled = in_bound && sv->bound_settled && pd->upload_ticket <= sv->settled_ticket ? 1u : 0u;
    const uint64_t ticket_now = g_upload_ticket.load(std::memory_order_acquire);     // uploads are synchronous: all in memory by now
    unsigned long long a = 0;
    const auto c1 = dbg ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    if (int rc = service_post(sv, rq, seq, &a)) return rc;
    if (dbg) {
        static double prep_us = 0, post_us = 0;
        static unsigned long n = 0;
        const auto c2 = std::chrono::steady_clock::now();
        prep_us += std::chrono::duration<double, std::micro>(c1 - c0).count();
        post_us += std::chrono::duration<double, std::micro>(c2 - c1).count();
        if ((++n & 0x3FFF) == 0) fprintf(stderr, "[service] per request: %.3f us before the post, %.3f us post + wait (%lu requests)\n", prep_us / n, post_us / n, n);
    }
    if (!rq.settled) {
        sv->settled_ticket = ticket_now;
        if (in_bound) sv->bound_settled = true;
    } else {
        ++sv->settled_requests;
    }
    sv->done_low = rq.pr.done_target;
    sv->done_hi = (uint32_t)(a >> 32);
    ++sv->requests;
    *found = (int)(a & 1);
    return SS_OK;
}

int ss_service_bind(ss_service *sv, const void *d_haystack, size_t len)
{
    if (!sv) return fail(SS_ERR_ARGUMENT, "service is NULL");
    if (len && !d_haystack) return fail(SS_ERR_ARGUMENT, "haystack is NULL");
    std::lock_guard<std::mutex> lock(sv->mu);
    sv->bound_lo = len ? static_cast<const uint8_t *>(d_haystack) : nullptr;
    sv->bound_hi = sv->bound_lo ? sv->bound_lo + len : nullptr;
    sv->bound_settled = false;                          // the next request inside the range acquires it, the ones after that do not
    return SS_OK;
}

int ss_service_unbind(ss_service *sv) { return ss_service_bind(sv, nullptr, 0); }

int ss_service_settled_requests(ss_service *sv, uint64_t *settled)
{
    if (!sv || !settled) return fail(SS_ERR_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lock(sv->mu);
    *settled = sv->settled_requests;
    return SS_OK;
}

int ss_service_counters(ss_service *sv, uint64_t *requests, uint64_t *kernel_launches)
{
    if (!sv) return fail(SS_ERR_ARGUMENT, "service is NULL");
    std::lock_guard<std::mutex> lock(sv->mu);
    if (requests) *requests = sv->requests;
    if (kernel_launches) *kernel_launches = sv->launches;
    return SS_OK;
}

int ss_service_set_default(ss_service *sv, int enabled)
{
    if (!sv) return fail(SS_ERR_ARGUMENT, "service is NULL");
    if (sv->dev < 0 || sv->dev >= kMaxDevices) return fail(SS_ERR_ARGUMENT, "device index out of range");
    if (enabled) {
        g_default_service[sv->dev].store(sv, std::memory_order_release);
    } else {
        ss_service *expect = sv;
        g_default_service[sv->dev].compare_exchange_strong(expect, nullptr, std::memory_order_acq_rel);
    }
    return SS_OK;
}

}  // extern "C"

namespace {

// The device's default service: what ss_service_set_default installed, or - with SLICESLICE_SERVICE=1 in the environment - a
// service started here on first use (default size and lease; it lives until the process ends).
ss_service *default_service(int dev)
{
    if (dev < 0 || dev >= kMaxDevices) return nullptr;
    ss_service *sv = g_default_service[dev].load(std::memory_order_acquire);
    if (sv) return sv;
    static const bool auto_on = []() { const char *v = getenv("SLICESLICE_SERVICE"); return v && v[0] == '1'; }();
    if (!auto_on) return nullptr;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    sv = g_default_service[dev].load(std::memory_order_acquire);
    if (sv) return sv;
    int workgroups = 0;
    double lease = 0;
    if (const char *e = getenv("SLICESLICE_SERVICE_WORKGROUPS")) workgroups = atoi(e);
    if (const char *e = getenv("SLICESLICE_SERVICE_LEASE_MS")) lease = atof(e);
    static bool refused[kMaxDevices];                   // (no large BAR, out of memory ...: asked once, not per search)
    if (refused[dev]) return nullptr;
    if (ss_service_start(workgroups, lease, &sv) != SS_OK) {
        refused[dev] = true;
        return nullptr;
    }
    g_default_service[dev].store(sv, std::memory_order_release);
    return sv;
}

}  // namespace

extern "C" {

void ss_service_stop(ss_service *sv)
{
    if (!sv) return;
    if (sv->dev >= 0 && sv->dev < kMaxDevices) {
        ss_service *expect = sv;
        g_default_service[sv->dev].compare_exchange_strong(expect, nullptr, std::memory_order_acq_rel);
    }
    DeviceGuard guard;
    (void)hipSetDevice(sv->dev);
    {
        std::lock_guard<std::mutex> lock(sv->mu);               // (a search still in its wait loop finishes first)
        ss::ServiceRequest bye;
        memset(&bye, 0, sizeof bye);
        bye.stop = 1;
        unsigned long long ignored = 0;
        (void)service_post(sv, bye, ++sv->seq, &ignored);      // (a kernel that does not answer leaves when its lease runs out)
    }
    service_free(sv);                                           // outside the lock: the mutex is part of what is freed
}

// DPP / alignbyte self-test used by the GPU tests: out must hold 320 uint32 (host memory).
int ss_selftest_dpp(uint32_t *out)
{
    if (!out) return fail(SS_ERR_ARGUMENT, "out is NULL");
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 320 * sizeof(uint32_t)));
    ss::dpp_probe_kernel<<<1, 64>>>(d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d, 320 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return SS_OK;
}

// ---- RCCL (dlopen'ed) -----------------------------------------------------------------------------

}  // extern "C"

namespace {

struct Id128 {
    char b[128];
};

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128 /* ncclUniqueId, by value */, int) = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            r.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
        if (!r.h) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.h, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))dlsym(r.h, "ncclCommCount");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.h, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.h, "ncclGroupEnd");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    });
    if (!r.h || !r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.CommCount || !r.GroupStart ||
        !r.GroupEnd || !r.AllReduce)
        return nullptr;
    return &r;
}

constexpr int kNcclInt32 = 2;   // ncclInt32 / ncclInt  (rccl.h ncclDataType_t)
constexpr int kNcclUint64 = 5;  // ncclUint64
constexpr int kNcclMax = 2;     // ncclMax              (rccl.h ncclRedOp_t: sum 0, prod 1, max 2, min 3)
constexpr int kNcclMin = 3;     // ncclMin

int rccl_fail(Rccl *r, int code, const char *what)
{
    return fail(SS_ERR_RCCL, "%s: %s", what, r && r->GetErrorString ? r->GetErrorString(code) : "rccl error");
}

}  // namespace

// One rank's end of a communicator (one process per GPU).  The found flag of a sharded search is never
// cleared: "found" is the call's epoch - every rank makes the same sequence of collective calls on a
// communicator, so the ranks' epochs agree - and since a rank's flag only ever holds epochs of earlier
// calls or of this one, max over the ranks == epoch exactly when some rank found the needle in THIS call.
struct ss_comm {
    void *comm = nullptr;
    int nranks = 1, rank = 0;
    int dev = 0;
    int epoch = 0;
    // A PAIR of ints goes through the all-reduce(MAX): [0] = this rank's found flag, [1] = "this rank failed its local
    // part", both epoch-valued and never cleared.  A rank whose scan could not be enqueued still takes part in the
    // collective (nobody is left waiting in ncclAllReduce, the ranks' epochs stay in step) and every rank learns of it:
    // the failing rank returns its own error, the others SS_ERR_PEER.
    int *d_flag = nullptr;      // int[2]
    int *d_recv = nullptr;      // int[2]: all-reduce(MAX) result
    int *h_flag = nullptr;      // pinned int[2]: read-back
    int *h_err = nullptr;       // pinned: source of the copy that raises d_flag[1]
    long long *h_word = nullptr;// pinned answer word of signal_flag_kernel (pair form) (spinning read-back)
    unsigned finds = 0;         // ss_find_sharded calls (every 256th still waits for the stream)
    uint64_t *d_best = nullptr; // uint64[2] scratch of ss_find_sharded: [0] = offset (MIN), [1] = all ones unless a rank failed
    uint64_t *h_best = nullptr; // pinned uint64[2]
    std::atomic<bool> busy{false};   // one search at a time per communicator: a second concurrent call is refused
};

// All ranks of a communicator inside ONE process (ncclCommInitAll): one stream, flag and read-back per device.
struct ss_comm_set {
    int ndev = 0;
    int combine = 0;            // SS_COMBINE_RCCL / SS_COMBINE_HOST
    int epoch = 0;
    std::vector<int> devs;
    std::vector<void *> comms;
    std::vector<hipStream_t> streams;
    std::vector<int *> d_flag, d_recv, h_flag;          // h_flag[g]: pinned mirror written by device g's finding wave
    std::vector<uint64_t *> d_best, d_best_recv;
    bool no_rccl = false;                               // test sets (SLICESLICE_COMM_SET_NO_RCCL): host combine only
    int *h_recv = nullptr;                              // pinned: device 0's all-reduce result
    long long *h_words = nullptr;                       // pinned: ndev answer words of signal_flag_kernel (spinning read-back)
    uint64_t *h_best = nullptr;                         // pinned: ndev offsets (host combine) / [0] = all-reduce result
    std::atomic<bool> busy{false};                      // one search at a time per set: a second concurrent call is refused
    // Cross-device early exit of ss_search_sharded_all: the host, which waits for the answer words anyway, watches the pinned
    // mirrors the finding waves write and stores the epoch into every OTHER device's flag through that device's PCIe BAR;
    // their workgroups see it at their next poll and leave.  Possible when every device's memory is CPU-visible.
    bool relay_ok = false;
    std::vector<volatile uint32_t *> hdp_flush;         // per device: HDP flush register (pushes the store out of the host data path)
};

namespace {

int next_comm_epoch(int *epoch, int *const *d_flags, const int *devs, int ndev, int *const *h_flags = nullptr)
{
    if (*epoch >= INT_MAX - 1 || *epoch < 0) {          // 2^31 calls: clear the flags so that no stale value equals a new epoch
        DeviceGuard guard;
        for (int g = 0; g < ndev; ++g) {
            (void)hipSetDevice(devs[g]);
            (void)hipDeviceSynchronize();
            (void)hipMemset(d_flags[g], 0, 2 * sizeof(int));       // flag + "a rank failed" (every flag word is a pair)
            if (h_flags) *h_flags[g] = 0;
        }
        *epoch = 0;
    }
    return ++*epoch;
}

void free_comm(ss_comm *c)
{
    Rccl *r = rccl();
    if (r && c->comm) r->CommDestroy(c->comm);
    (void)hipFree(c->d_flag);
    (void)hipFree(c->d_recv);
    (void)hipHostFree(c->h_flag);
    (void)hipHostFree(c->h_err);
    (void)hipHostFree(c->h_word);
    (void)hipFree(c->d_best);
    (void)hipHostFree(c->h_best);
    delete c;
}

void free_comm_set(ss_comm_set *set)
{
    Rccl *r = rccl();
    DeviceGuard guard;
    for (int g = 0; g < set->ndev; ++g) {
        (void)hipSetDevice(set->devs[g]);
        if (g < (int)set->streams.size() && set->streams[g]) {
            (void)hipStreamSynchronize(set->streams[g]);
            (void)hipStreamDestroy(set->streams[g]);
        }
        if (r && g < (int)set->comms.size() && set->comms[g]) r->CommDestroy(set->comms[g]);
        if (g < (int)set->d_flag.size()) (void)hipFree(set->d_flag[g]);
        if (g < (int)set->d_recv.size()) (void)hipFree(set->d_recv[g]);
        if (g < (int)set->h_flag.size()) (void)hipHostFree(set->h_flag[g]);
        if (g < (int)set->d_best.size()) (void)hipFree(set->d_best[g]);
        if (g < (int)set->d_best_recv.size()) (void)hipFree(set->d_best_recv[g]);
    }
    (void)hipHostFree(set->h_recv);
    (void)hipHostFree(set->h_words);
    (void)hipHostFree(set->h_best);
    delete set;
}

}  // namespace

extern "C" {

int ss_comm_unique_id(uint8_t id[SS_UNIQUE_ID_BYTES])
{
    Rccl *r = rccl();
    if (!r) return fail(SS_ERR_RCCL, "librccl could not be loaded: %s", dlerror());
    static_assert(SS_UNIQUE_ID_BYTES == sizeof(Id128), "ncclUniqueId is 128 bytes");
    if (int rc = r->GetUniqueId(id)) return rccl_fail(r, rc, "ncclGetUniqueId");
    return SS_OK;
}

int ss_comm_init_rank(const uint8_t id[SS_UNIQUE_ID_BYTES], int nranks, int rank, ss_comm **out)
{
    if (!out || !id) return fail(SS_ERR_ARGUMENT, "NULL argument");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(SS_ERR_ARGUMENT, "bad rank %d of %d", rank, nranks);
    Rccl *r = rccl();
    if (!r) return fail(SS_ERR_RCCL, "librccl could not be loaded: %s", dlerror());
    ss_comm *c = new (std::nothrow) ss_comm;
    if (!c) return fail(SS_ERR_NOMEM, "out of memory");
    Id128 uid;
    memcpy(uid.b, id, sizeof uid);
    if (int rc = r->CommInitRank(&c->comm, nranks, uid, rank)) {
        c->comm = nullptr;
        free_comm(c);
        return rccl_fail(r, rc, "ncclCommInitRank");
    }
    c->nranks = nranks;
    c->rank = rank;
    hipError_t e = hipGetDevice(&c->dev);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_flag, 2 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->d_flag, 0, 2 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_recv, 2 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->d_recv, 0, 2 * sizeof(int));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_flag, 2 * sizeof(int), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_err, sizeof(int), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_word, sizeof(long long), hipHostMallocDefault);
    if (e == hipSuccess) *c->h_word = 0;
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_best, 2 * sizeof(uint64_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_best, 2 * sizeof(uint64_t), hipHostMallocDefault);
    if (e != hipSuccess) {                               // nothing half-built is left behind (communicator included)
        free_comm(c);
        return fail(SS_ERR_HIP, "communicator scratch: %s", hipGetErrorString(e));
    }
    *out = c;
    return SS_OK;
}

void ss_comm_free(ss_comm *c)
{
    if (c) free_comm(c);
}

int ss_debug_set_comm_epoch(ss_comm *c, ss_comm_set *set, int value)
{
    if (c) c->epoch = value;
    if (set) set->epoch = value;
    return SS_OK;
}

int ss_comm_count(const ss_comm *c, int *nranks)
{
    if (!c || !nranks) return fail(SS_ERR_ARGUMENT, "NULL argument");
    Rccl *r = rccl();
    if (!r) return fail(SS_ERR_RCCL, "librccl not loaded");
    if (int rc = r->CommCount(c->comm, nranks)) return rccl_fail(r, rc, "ncclCommCount");   // what RCCL itself says
    return SS_OK;
}

int ss_comm_allreduce_flag(ss_comm *c, int *d_flag, void *hip_stream, int *found)
{
    if (!c || !d_flag) return fail(SS_ERR_ARGUMENT, "NULL argument");
    Rccl *r = rccl();
    if (!r) return fail(SS_ERR_RCCL, "librccl not loaded");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    // OR over {0,1} == MAX; RCCL has no bitwise-OR reduction.
    if (int rc = r->AllReduce(d_flag, d_flag, 1, kNcclInt32, kNcclMax, c->comm, st)) return rccl_fail(r, rc, "ncclAllReduce");
    if (found) {
        HIP_TRY(hipMemcpyAsync(c->h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        *found = *c->h_flag != 0;
    }
    return SS_OK;
}

int ss_search_sharded(const ss_searcher *s, const void *d_shard, size_t shard_len, ss_comm *c,
                      void *hip_stream, int *found)
{
    if (!s || !c || !found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (shard_len && !d_shard) return fail(SS_ERR_ARGUMENT, "shard is NULL");
    Rccl *r = rccl();
    if (!r) return fail(SS_ERR_RCCL, "librccl not loaded");
    if (s->n == 0) { *found = 1; return SS_OK; }            // N0 (x86.rs:500): the same on every rank, nothing to combine
    BusyGuard busy(&c->busy);
    if (!busy.mine) return fail(SS_ERR_ARGUMENT, "this communicator is in use by another search (one search at a time)");
    SearchGate gate(s);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int epoch = next_comm_epoch(&c->epoch, &c->d_flag, &c->dev, 1);
    // The local part.  Whatever happens here, this rank ENTERS THE COLLECTIVE below: a rank that returned early would
    // leave the others waiting in ncclAllReduce for good and put the ranks' epochs out of step.
    int local_rc = SS_OK;
    char local_msg[sizeof g_err] = "";
    if (shard_len >= s->n) {                                // a shard shorter than the needle holds no candidate
        PerDevice *pd = nullptr;
        local_rc = get_per_device(s, &pd);
        if (local_rc == SS_OK) local_rc = enqueue_scan(s, pd, d_shard, shard_len, st, c->d_flag, false, 0, nullptr, epoch);
        if (local_rc != SS_OK) {
            snprintf(local_msg, sizeof local_msg, "%s", g_err);
            *c->h_err = epoch;                              // contributes "not found" and raises the pair's second word
            (void)hipMemcpyAsync(c->d_flag + 1, c->h_err, sizeof(int), hipMemcpyHostToDevice, st);
        }
    }
    auto done = [&](int any_failed) {
        if (local_rc != SS_OK) return fail(local_rc, "%s", local_msg);
        if (any_failed) return fail(SS_ERR_PEER, "another rank failed the local part of this sharded search; no answer");
        return (int)SS_OK;
    };
    if (int rc = r->AllReduce(c->d_flag, c->d_recv, 2, kNcclInt32, kNcclMax, c->comm, st)) return rccl_fail(r, rc, "ncclAllReduce");
    static const bool spin_ok = []() { const char *v = getenv("SLICESLICE_SPIN_WAIT"); return !(v && v[0] == '0'); }();
    const double estimate = scan_estimate_us(shard_len) + 100.0;      // + the collective
    if (spin_ok && estimate <= kSpinMaxEstimateUs && scan_estimate_us(shard_len) >= kSpinMinEstimateUs) {
        // the answer word behind the all-reduce, and a bounded spin on it (see spin_for_word); ranks that arrive late in
        // the collective make the others' spins run out, which costs those nothing but the stream wait they had before
        __atomic_store_n(c->h_word, 0ll, __ATOMIC_RELAXED);
        ss::signal_flag_kernel<<<1, 1, 0, st>>>(c->d_recv, epoch, c->h_word, 1);
        HIP_TRY(hipGetLastError());
        int failed = 0;
        if (spin_for_shard_word(c->h_word, epoch, estimate, found, &failed)) {
            if ((epoch & 255) == 0) HIP_TRY(hipStreamSynchronize(st));
            return done(failed);
        }
        HIP_TRY(hipStreamSynchronize(st));
        if (!spin_for_shard_word(c->h_word, epoch, 0.0, found, &failed)) return fail(SS_ERR_HIP, "the answer word was not written");
        return done(failed);
    }
    HIP_TRY(hipMemcpyAsync(c->h_flag, c->d_recv, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *found = c->h_flag[0] == epoch;
    return done(c->h_flag[1] == epoch);
}

// Sharded find: every rank lowers its uint64 with shard_begin + local offset of its leftmost match, ONE
// all-reduce(MIN) over a uint64 PAIR gives the global leftmost offset (SS_NPOS = all ones = absent everywhere) and
// tells every rank whether some rank failed its local part (second word: all ones unless so) - collective-safe the
// same way as ss_search_sharded.
int ss_find_sharded(const ss_searcher *s, const void *d_shard, size_t shard_len, uint64_t shard_begin, ss_comm *c,
                    void *hip_stream, uint64_t *position)
{
    if (!s || !c || !position) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (shard_len && !d_shard) return fail(SS_ERR_ARGUMENT, "shard is NULL");
    Rccl *r = rccl();
    if (!r) return fail(SS_ERR_RCCL, "librccl not loaded");
    BusyGuard busy(&c->busy);
    if (!busy.mine) return fail(SS_ERR_ARGUMENT, "this communicator is in use by another search (one search at a time)");
    SearchGate gate(s);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    int local_rc = SS_OK;
    char local_msg[sizeof g_err] = "";
    hipError_t e0 = hipMemsetAsync(c->d_best, 0xFF, 2 * sizeof(uint64_t), st);
    if (e0 != hipSuccess) local_rc = fail(SS_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e0));
    if (local_rc == SS_OK) local_rc = ss_find_device_async(s, d_shard, shard_len, shard_begin, hip_stream, c->d_best);
    if (local_rc != SS_OK) {                                // still enter the collective: see ss_search_sharded
        snprintf(local_msg, sizeof local_msg, "%s", g_err);
        (void)hipMemsetAsync(c->d_best + 1, 0, sizeof(uint64_t), st);
    }
    auto done = [&](uint64_t status) {
        if (local_rc != SS_OK) return fail(local_rc, "%s", local_msg);
        if (status != ~0ull) return fail(SS_ERR_PEER, "another rank failed the local part of this sharded find; no answer");
        return (int)SS_OK;
    };
    if (int rc = r->AllReduce(c->d_best, c->d_best, 2, kNcclUint64, kNcclMin, c->comm, st)) return rccl_fail(r, rc, "ncclAllReduce");
    static const bool spin_ok = []() { const char *v = getenv("SLICESLICE_SPIN_WAIT"); return !(v && v[0] == '0'); }();
    const double estimate = scan_estimate_us(shard_len) + 100.0;
    if (spin_ok && estimate <= kSpinMaxEstimateUs && scan_estimate_us(shard_len) >= kSpinMinEstimateUs) {
        // as ss_search_sharded: the pinned mirror starts as "pending", a one-lane kernel behind the all-reduce stores the
        // pair (and re-arms nothing: d_best is this communicator's scratch, set to all ones at the top of every call)
        constexpr uint64_t kPending = ~0ull - 1;
        __atomic_store_n(c->h_best, kPending, __ATOMIC_RELAXED);
        ss::publish_best_kernel<<<1, 1, 0, st>>>(c->d_best, c->h_best, 1);
        HIP_TRY(hipGetLastError());
        const auto t0 = std::chrono::steady_clock::now();
        const auto budget = std::chrono::microseconds((long long)(2.0 * estimate) + 300);
        for (unsigned spins = 0;; ++spins) {
            const uint64_t v = __atomic_load_n(c->h_best, __ATOMIC_ACQUIRE);
            if (v != kPending) {
                *position = v;
                if ((++c->finds & 255) == 0) HIP_TRY(hipStreamSynchronize(st));
                return done(__atomic_load_n(c->h_best + 1, __ATOMIC_RELAXED));
            }
            cpu_relax();
            if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > budget) break;
        }
        HIP_TRY(hipStreamSynchronize(st));
        *position = __atomic_load_n(c->h_best, __ATOMIC_ACQUIRE);
        return done(__atomic_load_n(c->h_best + 1, __ATOMIC_RELAXED));
    }
    HIP_TRY(hipMemcpyAsync(c->h_best, c->d_best, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *position = c->h_best[0];
    return done(c->h_best[1]);
}

// ---- multi-GPU inside ONE process ---------------------------------------------------------------------
// What a drop-in `search_in(&self, &[u8]) -> bool` over the 8 GPUs of a node calls (x86.rs:523 has no
// launcher to lean on): ncclCommInitAll once, then per search one scan per device on that device's stream,
// the G all-reduces inside ONE ncclGroupStart/End, one read-back.
int ss_comm_init_all(int ndev, const int *devs, ss_comm_set **out)
{
    if (!out) return fail(SS_ERR_ARGUMENT, "out is NULL");
    *out = nullptr;
    int visible = 0;
    HIP_TRY(hipGetDeviceCount(&visible));
    // SLICESLICE_COMM_SET_NO_RCCL=1 (tests): no communicators are created and a device may be listed several times, so
    // that the per-device bookkeeping of a G > 1 set can run on a one-GPU box; such a set only offers the host combine.
    const char *nr = getenv("SLICESLICE_COMM_SET_NO_RCCL");
    const bool no_rccl = nr && nr[0] == '1';
    Rccl *r = no_rccl ? nullptr : rccl();
    if (!no_rccl && !r) return fail(SS_ERR_RCCL, "librccl could not be loaded: %s", dlerror());
    if (ndev < 1 || ndev > (no_rccl ? 64 : visible)) return fail(SS_ERR_ARGUMENT, "%d devices requested, %d visible", ndev, visible);
    ss_comm_set *set = new (std::nothrow) ss_comm_set;
    if (!set) return fail(SS_ERR_NOMEM, "out of memory");
    set->ndev = ndev;
    set->no_rccl = no_rccl;
    if (no_rccl) set->combine = SS_COMBINE_HOST;
    for (int g = 0; g < ndev; ++g) {
        const int d = devs ? devs[g] : g;
        if (d < 0 || d >= visible) {
            delete set;
            return fail(SS_ERR_ARGUMENT, "device %d out of range (%d visible)", d, visible);
        }
        for (int k = 0; k < g && !no_rccl; ++k)
            if (set->devs[k] == d) {
                delete set;
                return fail(SS_ERR_ARGUMENT, "device %d listed twice", d);
            }
        set->devs.push_back(d);
    }
    set->comms.assign(ndev, nullptr);
    set->streams.assign(ndev, nullptr);
    set->d_flag.assign(ndev, nullptr);
    set->d_recv.assign(ndev, nullptr);
    set->h_flag.assign(ndev, nullptr);
    set->d_best.assign(ndev, nullptr);
    set->d_best_recv.assign(ndev, nullptr);
    DeviceGuard guard;
    if (int rc = no_rccl ? 0 : r->CommInitAll(set->comms.data(), ndev, set->devs.data())) {
        set->comms.assign(ndev, nullptr);
        free_comm_set(set);
        return rccl_fail(r, rc, "ncclCommInitAll");
    }
    hipError_t e = hipSuccess;
    for (int g = 0; g < ndev && e == hipSuccess; ++g) {
        e = hipSetDevice(set->devs[g]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&set->streams[g], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void **)&set->d_flag[g], 2 * sizeof(int));
        if (e == hipSuccess) e = hipMemset(set->d_flag[g], 0, 2 * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&set->d_recv[g], sizeof(int));
        if (e == hipSuccess) e = hipHostMalloc((void **)&set->h_flag[g], sizeof(int), hipHostMallocPortable | hipHostMallocMapped);
        if (e == hipSuccess) *set->h_flag[g] = 0;
        if (e == hipSuccess) e = hipMalloc((void **)&set->d_best[g], sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void **)&set->d_best_recv[g], sizeof(uint64_t));
    }
    if (e == hipSuccess) e = hipHostMalloc((void **)&set->h_recv, sizeof(int), hipHostMallocPortable | hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc((void **)&set->h_words, (size_t)ndev * sizeof(long long), hipHostMallocPortable | hipHostMallocMapped);
    if (e == hipSuccess) memset(set->h_words, 0, (size_t)ndev * sizeof(long long));
    if (e == hipSuccess) e = hipHostMalloc((void **)&set->h_best, (size_t)ndev * sizeof(uint64_t), hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) {
        free_comm_set(set);
        return fail(SS_ERR_HIP, "communicator set scratch: %s", hipGetErrorString(e));
    }
    {
        const char *off = getenv("SLICESLICE_CROSS_EXIT");
        set->relay_ok = !(off && off[0] == '0');
        set->hdp_flush.assign((size_t)ndev, nullptr);
        for (int g = 0; g < ndev; ++g) {
            int large = 0;
            hipDeviceProp_t prop;
            if (hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, set->devs[g]) != hipSuccess || !large) set->relay_ok = false;
            if (hipGetDeviceProperties(&prop, set->devs[g]) == hipSuccess) set->hdp_flush[g] = prop.hdpMemFlushCntl;
            (void)hipGetLastError();
            (void)hipSetDevice(set->devs[g]);
            (void)hipDeviceSynchronize();               // the memsets above are asynchronous to the host: done before anyone stores there
        }
    }
    *out = set;
    return SS_OK;
}

void ss_comm_set_free(ss_comm_set *set)
{
    if (set) free_comm_set(set);
}

int ss_comm_set_size(const ss_comm_set *set) { return set ? set->ndev : 0; }

int ss_comm_set_device(const ss_comm_set *set, int index, int *device)
{
    if (!set || !device || index < 0 || index >= set->ndev) return fail(SS_ERR_ARGUMENT, "bad communicator-set index");
    *device = set->devs[index];
    return SS_OK;
}

int ss_comm_set_combine(ss_comm_set *set, int combine)
{
    if (!set || (combine != SS_COMBINE_RCCL && combine != SS_COMBINE_HOST)) return fail(SS_ERR_ARGUMENT, "bad combine mode");
    if (set->no_rccl && combine == SS_COMBINE_RCCL) return fail(SS_ERR_RCCL, "this set was created without communicators");
    set->combine = combine;
    return SS_OK;
}

int ss_search_sharded_all(const ss_searcher *s, const void *const *d_shards, const size_t *shard_lens, ss_comm_set *set,
                          int *found)
{
    if (!s || !d_shards || !shard_lens || !set || !found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (s->n == 0) { *found = 1; return SS_OK; }            // N0 (x86.rs:500)
    Rccl *r = set->combine == SS_COMBINE_RCCL ? rccl() : nullptr;
    if (set->combine == SS_COMBINE_RCCL && !r) return fail(SS_ERR_RCCL, "librccl not loaded");
    const int G = set->ndev;
    for (int g = 0; g < G; ++g)
        if (shard_lens[g] && !d_shards[g]) return fail(SS_ERR_ARGUMENT, "shard %d is NULL", g);
    // the set's epoch, streams, flags and pinned words are ONE search's scratch: a second thread would corrupt both answers
    BusyGuard busy(&set->busy);
    if (!busy.mine) return fail(SS_ERR_ARGUMENT, "this communicator set is in use by another search (one search at a time per set)");
    SearchGate gate(s);
    DeviceGuard guard;
    const int epoch = next_comm_epoch(&set->epoch, set->d_flag.data(), set->devs.data(), G, set->h_flag.data());
    int rc = SS_OK;
    // 1. one scan per device, each on its device's stream; the finding wave also writes the epoch to that device's
    //    pinned-host mirror
    for (int g = 0; g < G && rc == SS_OK; ++g) {
        if (shard_lens[g] < s->n) continue;                 // shorter than the needle: no candidate, flag stays old
        if (hipSetDevice(set->devs[g]) != hipSuccess) { rc = fail(SS_ERR_HIP, "hipSetDevice(%d) failed", set->devs[g]); break; }
        PerDevice *pd = nullptr;
        rc = get_per_device(s, &pd);
        if (rc == SS_OK) rc = enqueue_scan(s, pd, d_shards[g], shard_lens[g], set->streams[g], set->d_flag[g], false, 0, set->h_flag[g], epoch);
    }
    // 2. combine: G all-reduce(MAX) calls as ONE group (the default), or no collective at all - the host ORs the
    //    G pinned mirrors (possible only because all ranks live in this process)
    if (rc == SS_OK && set->combine == SS_COMBINE_RCCL) {
        int nrc = r->GroupStart();
        for (int g = 0; g < G && nrc == 0; ++g)
            nrc = r->AllReduce(set->d_flag[g], set->d_recv[g], 1, kNcclInt32, kNcclMax, set->comms[g], set->streams[g]);
        const int erc = r->GroupEnd();
        if (nrc == 0) nrc = erc;
        if (nrc != 0) rc = rccl_fail(r, nrc, "grouped ncclAllReduce");
        if (rc == SS_OK) {
            hipError_t e = hipSetDevice(set->devs[0]);
            if (e == hipSuccess) e = hipMemcpyAsync(set->h_recv, set->d_recv[0], sizeof(int), hipMemcpyDeviceToHost, set->streams[0]);
            if (e != hipSuccess) rc = fail(SS_ERR_HIP, "flag read-back: %s", hipGetErrorString(e));
        }
    }
    // 3. every stream is drained before the call returns: the haystacks are only borrowed for the call.  Scans short
    //    enough to be waited for by spinning (spin_for_word) end with a one-lane kernel per device that stores the device's
    //    answer word - behind the scan and the all-reduce, so a word that has arrived says its stream is done - and the
    //    host collects the G words; whatever is missing when the spin budget runs out is waited for on the stream.
    static const bool spin_ok = []() { const char *v = getenv("SLICESLICE_SPIN_WAIT"); return !(v && v[0] == '0'); }();
    size_t longest = 0;
    for (int g = 0; g < G; ++g) longest = shard_lens[g] > longest ? shard_lens[g] : longest;
    const double estimate = scan_estimate_us(longest) + 100.0;
    bool spun = false;
    int any_word = 0;
    if (rc == SS_OK && spin_ok && estimate <= kSpinMaxEstimateUs && scan_estimate_us(longest) >= kSpinMinEstimateUs) {
        bool launched = true;
        for (int g = 0; g < G && launched; ++g) {
            __atomic_store_n(set->h_words + g, 0ll, __ATOMIC_RELAXED);
            launched = hipSetDevice(set->devs[g]) == hipSuccess;
            if (launched) {
                ss::signal_flag_kernel<<<1, 1, 0, set->streams[g]>>>(set->combine == SS_COMBINE_RCCL ? set->d_recv[g] : set->d_flag[g],
                                                                     epoch, set->h_words + g, 0);
                launched = hipGetLastError() == hipSuccess;
            }
        }
        if (launched) {
            // Collect the G answer words; meanwhile - cross-device early exit - watch the pinned mirrors: the first device that
            // reports a match has its epoch stored into every other device's flag through the BAR, so that THEIR grids stop
            // scanning too (a match in shard 0 of a 64 GiB haystack over eight devices otherwise costs the full 1.2 ms scan of
            // the seven others).  The flag only ever means "found somewhere": the OR of the answers is unchanged.
            const auto t0 = std::chrono::steady_clock::now();
            const auto budget = std::chrono::microseconds((long long)(2.0 * estimate) + 300);
            uint64_t got = 0;
            bool relayed = !set->relay_ok || G < 2;
            spun = true;
            for (unsigned spins = 0; got != (G >= 64 ? ~0ull : (1ull << G) - 1); ++spins) {
                for (int g = 0; g < G; ++g) {
                    if ((got >> g) & 1) continue;
                    const long long v = __atomic_load_n(set->h_words + g, __ATOMIC_ACQUIRE);
                    if (((uint32_t)v >> 1) == (uint32_t)epoch) {
                        any_word |= (int)(v & 1);
                        got |= 1ull << g;
                    }
                }
                if (!relayed) {
                    for (int g = 0; g < G; ++g) {
                        if (__atomic_load_n(set->h_flag[g], __ATOMIC_ACQUIRE) != epoch) continue;
                        for (int o = 0; o < G; ++o) {
                            if (o == g || ((got >> o) & 1) || shard_lens[o] < s->n) continue;
                            *reinterpret_cast<volatile int *>(set->d_flag[o]) = epoch;
                        }
                        _mm_sfence();
                        for (int o = 0; o < G; ++o)
                            if (o != g && set->hdp_flush[o]) __atomic_store_n(set->hdp_flush[o], 1u, __ATOMIC_RELAXED);
                        relayed = true;
                        break;
                    }
                }
                cpu_relax();
                if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > budget) { spun = false; break; }
            }
            if (spun && (epoch & 255) != 0) {
                *found = any_word;
                return SS_OK;
            }
        }
    }
    for (int g = 0; g < G; ++g) {
        hipError_t e = hipSetDevice(set->devs[g]);
        if (e == hipSuccess) e = hipStreamSynchronize(set->streams[g]);
        if (e != hipSuccess && rc == SS_OK) rc = fail(SS_ERR_HIP, "stream wait on device %d: %s", set->devs[g], hipGetErrorString(e));
    }
    if (rc != SS_OK) return rc;
    if (spun) {
        *found = any_word;
        return SS_OK;
    }
    int any = 0;
    if (set->combine == SS_COMBINE_RCCL) {
        any = *set->h_recv == epoch;
    } else {
        for (int g = 0; g < G; ++g) any |= __atomic_load_n(set->h_flag[g], __ATOMIC_ACQUIRE) == epoch;
    }
    *found = any;
    return SS_OK;
}

int ss_find_sharded_all(const ss_searcher *s, const void *const *d_shards, const size_t *shard_lens,
                        const uint64_t *shard_begins, ss_comm_set *set, uint64_t *position)
{
    if (!s || !d_shards || !shard_lens || !shard_begins || !set || !position) return fail(SS_ERR_ARGUMENT, "NULL argument");
    Rccl *r = set->combine == SS_COMBINE_RCCL ? rccl() : nullptr;
    if (set->combine == SS_COMBINE_RCCL && !r) return fail(SS_ERR_RCCL, "librccl not loaded");
    const int G = set->ndev;
    for (int g = 0; g < G; ++g)
        if (shard_lens[g] && !d_shards[g]) return fail(SS_ERR_ARGUMENT, "shard %d is NULL", g);
    BusyGuard busy(&set->busy);
    if (!busy.mine) return fail(SS_ERR_ARGUMENT, "this communicator set is in use by another search (one search at a time per set)");
    SearchGate gate(s);
    DeviceGuard guard;
    int rc = SS_OK;
    for (int g = 0; g < G && rc == SS_OK; ++g) {
        hipError_t e = hipSetDevice(set->devs[g]);
        if (e == hipSuccess) e = hipMemsetAsync(set->d_best[g], 0xFF, sizeof(uint64_t), set->streams[g]);
        if (e != hipSuccess) { rc = fail(SS_ERR_HIP, "device %d: %s", set->devs[g], hipGetErrorString(e)); break; }
        rc = ss_find_device_async(s, d_shards[g], shard_lens[g], shard_begins[g], set->streams[g], set->d_best[g]);
    }
    if (rc == SS_OK && set->combine == SS_COMBINE_RCCL) {
        int nrc = r->GroupStart();
        for (int g = 0; g < G && nrc == 0; ++g)
            nrc = r->AllReduce(set->d_best[g], set->d_best_recv[g], 1, kNcclUint64, kNcclMin, set->comms[g], set->streams[g]);
        const int erc = r->GroupEnd();
        if (nrc == 0) nrc = erc;
        if (nrc != 0) rc = rccl_fail(r, nrc, "grouped ncclAllReduce");
    }
    if (rc == SS_OK) {                                       // read-back: the reduced value from device 0, or all G minima
        const int nread = set->combine == SS_COMBINE_RCCL ? 1 : G;
        for (int g = 0; g < nread && rc == SS_OK; ++g) {
            hipError_t e = hipSetDevice(set->devs[g]);
            if (e == hipSuccess)
                e = hipMemcpyAsync(set->h_best + g, set->combine == SS_COMBINE_RCCL ? set->d_best_recv[g] : set->d_best[g],
                                   sizeof(uint64_t), hipMemcpyDeviceToHost, set->streams[g]);
            if (e != hipSuccess) rc = fail(SS_ERR_HIP, "offset read-back: %s", hipGetErrorString(e));
        }
    }
    for (int g = 0; g < G; ++g) {
        hipError_t e = hipSetDevice(set->devs[g]);
        if (e == hipSuccess) e = hipStreamSynchronize(set->streams[g]);
        if (e != hipSuccess && rc == SS_OK) rc = fail(SS_ERR_HIP, "stream wait on device %d: %s", set->devs[g], hipGetErrorString(e));
    }
    if (rc != SS_OK) return rc;
    uint64_t best = set->h_best[0];
    if (set->combine != SS_COMBINE_RCCL)
        for (int g = 1; g < G; ++g) best = set->h_best[g] < best ? set->h_best[g] : best;
    *position = best;
    return SS_OK;
}

}  // extern "C"
