// ss_internal.hpp - what the host-side translation units of libsliceslice_hip share (namespace ssh; nothing here is exported:
// the library is built with -fvisibility=hidden and only the SS_API entry points of include/sliceslice_hip*.h are visible).
//
//   ss_core.hip     errors, device info, control-block pools, ss_searcher (constructors, filter-byte choice, accessors)
//   ss_scan.hip     kernel selection, the Problem of a (searcher, haystack), enqueue_scan, ss_search_device / ss_find_device
//   ss_census.hip   what a searcher learns about a haystack by asking it: candidate census, byte histogram -> launch hints
//   ss_host.hip     host-slice and host-file front ends, the byte histogram (rows f2, f3 of SURVEY.md 8f)
//   ss_batched.hip  batched search / find, batch plans, short-haystack pairs (config 5, row f4)
//   ss_service.hip  the resident search service - NOT in libsliceslice_hip.so: libsliceslice_hip_service.so and the hooks builds
//   ss_comm.hip     RCCL, the range-sharded searches (one process per GPU and all GPUs from one process), ss_shard_range
//   ss_tools.hip    benchmark / tuning helpers and the test hooks (sliceslice_hip_tuning.h)
// The scan kernels live in scan_filters.hpp / scan_kernels.hpp (instantiated in scan_inst_*.hip), the others next to their users.
// There is no CPU search path in any of them.
#pragma once
#include <hip/hip_runtime.h>
#include <emmintrin.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/sliceslice_hip.h"
#include "../../include/sliceslice_hip_service.h"      // (declarations only: ss_service.hip is linked into libsliceslice_hip_service.so and the hooks builds)
#include "../../include/sliceslice_hip_tuning.h"
#include "scan_filters.hpp"

namespace ssh {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const char *last_error();               // the calling thread's message buffer (what ss_last_error returns)

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return ssh::fail(e_ == hipErrorNoDevice ? SS_ERR_NO_DEVICE : SS_ERR_HIP, "%s: %s (%s:%d)",  \
                             #expr, hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

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

struct DeviceInfo {
    int cus = 0;
    bool ok = false;
    bool gfx950 = false;
    bool large_bar = false;                     // every byte of the device's memory is CPU-visible through the PCIe BAR
    volatile uint32_t *hdp_flush = nullptr;     // the device's HDP_MEM_COHERENCY_FLUSH_CNTL register (CPU-visible), or null
};
int device_info(int dev, DeviceInfo *out);

constexpr int kSlots = 64;
constexpr int kMaxDevices = 64;
constexpr uint32_t kFindTagMax = (1u << (64 - ss::kFindOffsetBits)) - 2;   // keys tag << kFindOffsetBits stay below all-ones
constexpr uint32_t kDoneLowMax = 0x7FFF0000u;                               // start over before the low half could carry
constexpr uint64_t kShiftMaxD = 62;      // cross-lane kernels: d + 1 halo chunks must fit one piece (tools/tune.py: wins up to d = 62)

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
    // Workgroups per CU (ss_scan.hip): what the candidate census said about the haystacks this searcher has been used on, on this
    // device - launch tuning only; no search result depends on it.  A small table (the searcher's latest haystacks), the one census
    // that may be in flight, and the latest launch's shape for ss_searcher_last_launch.  census_lock is a try-lock: a thread that
    // finds it taken launches with what it has.
    // One (searcher, haystack) pair's census state (ss_census.hip): the counts, and the filter bytes IN FORCE on this haystack -
    // the searcher's own triple at first, then whatever a few rounds of measured, one-byte-at-a-time improvement arrive at.
    struct Census {
        const void *hay = nullptr;
        size_t len = 0;
        uint32_t gen = 0;           // the searcher's filter generation the entry belongs to
        uint32_t state = 0;         // 0 = empty, 1 = a census is in flight and no counts of cur[] are in yet, 2 = counts of cur[] are in
        uint32_t tag = 0;           // tag of the census in flight (inflight != 0)
        uint32_t uses = 0;          // scans that went by these counts (everything is looked at again every 256: a buffer may be refilled in place)
        uint64_t sums = 0;          // counts of cur[]: tiles3 | tiles2 << 11 | match tiles << 22 | candidate lanes << 33 (aux_kernels.hpp)
        size_t cur[3] = {0, 0, 0};  // the first-phase bytes in force on this haystack (needle indices; cur[0] the smallest)
        bool adopted = false;       // cur[] differs from the searcher's own triple
        // the census in flight: of cur[] (inflight == 1) or of a proposal on TRIAL (inflight == 2: prop[]); `roles`: which coordinate the
        // per-position counts are gathered FOR - the kernel's "pair" is the other two bytes, so pair_match[k] is what a first phase
        // with k in that coordinate's place would let through
        uint32_t inflight = 0;
        int roles = 2;
        size_t prop[3] = {0, 0, 0};
        uint32_t prop_kind = 0;     // 1 = from the haystack's histogram, 2 = one coordinate moved by measurement, 3 = the near form of a far pair,
                                    // 4 = a jump to the byte that kills most of today's candidates, 5 = the compact form of a filter that
                                    // meets no candidates (all three bytes within eight)
        uint32_t trials = 0, accepted = 0;
        // the descent: coordinates the library may move (bit j), the next one to look at, coordinates looked at since the last improvement
        uint32_t free_mask = 0, coord = 2, stale = 0, rounds = 0;
        bool settled = false, hist_tried = false, near_tried = false, jump_tried = false, compact_tried = false;
        // what the latest census of cur[] MEASURED about survival: how many of the sampled pair / triple candidates match the needle at
        // position k (< 64); stats_roles = the coordinate they were gathered for (-1: none at hand)
        int stats_roles = -1;
        uint16_t pair_match[64] = {0}, triple_match[64] = {0};
        uint32_t pair_lanes = 0, triple_lanes = 0;
        uint32_t deep_lanes = 0;    // of cur[]: sampled candidates that only the compare in memory can tell from a match (aux_kernels.hpp)
        // ... and the second level's schedule ordered by it (enqueue_scan takes it instead of the static rarity order)
        bool have_order = false;
        uint32_t norder = 0;
        uint64_t order_idx[2] = {0, 0}, order_val[2] = {0, 0};
        uint64_t stamp = 0;
    } census[4];
    uint32_t census_lock = 0, census_tag = 0;
    int census_pending = -1;
    uint64_t census_clock = 0;
    unsigned long long *d_census = nullptr;     // the census kernel's accumulator word
    unsigned long long *h_census = nullptr;     // pinned: [0] sums, [1] tag of the launch they belong to
    uint32_t *d_stats = nullptr;                // the census kernel's per-position match counters (ss::kCensusStatWords)
    uint32_t *h_stats = nullptr;                // pinned: the counters of the launch h_census[1] names
    int last_occ = 0, last_found = 0;           // workgroups per CU of the latest launch; the latest synchronous search found the needle
    unsigned last_grid = 0;
    int last_mode = 0;                          // kernel family of the latest launch: 0 single stream, 2 / 3 cross-lane with / without the third byte
    uint32_t done_low[64] = {0}, done_hi[64] = {0};
    uint32_t find_tag[64] = {0};            // next key of the slot; counts down from kFindTagMax
    uint64_t free_mask = 0;
    int epoch[64] = {0};        // per slot: the "found" value of the slot's latest call
    uint64_t upload_ticket = 0; // g_upload_ticket when d_needle had been written (a resident service acquires what is newer)
    uint32_t block = ~0u;       // control block of the device's pool (BlockPool): everything above points into it ...
    bool needle_own = false;    // ... except a needle too long for the block, which has an allocation of its own
};
extern std::atomic<uint64_t> g_upload_ticket;

// exit() has begun: thread-local HIP objects and pool blocks are leaked instead of released (see ss_core.hip, ExitMark)
bool process_exiting();

// `bytes` (a multiple of 16) from host memory into device memory through the BAR, complete before anything the caller does next
// can reach the device (ss_core.hip).
void bar_write(uint8_t *d_dst, const uint8_t *src, size_t bytes, volatile uint32_t *hdp_flush);
bool bar_writes_allowed();              // SLICESLICE_NO_BAR_WRITES != 1

}  // namespace ssh

struct ss_searcher {
    uint64_t uid = 0;         // unique for the life of the process (never 0): what timing records and communicator sets remember
                              // a searcher by - an address may be handed out again after ss_searcher_free
    std::vector<uint8_t> needle;
    size_t n = 0;
    size_t position = 0;      // the API position (what ss_searcher_position reports; x86.rs:468)
    // The needle bytes the filter is SAID to test (ss_searcher_filter3): needle[fa], needle[fb], fa <= fb (fa == fb == 0 for
    // one-byte needles), and - only when fb - fa <= 15 - a third one, fa < fc <= fa + 15 (== fb: none).  with_position callers
    // get their byte plus partners (filter_for_position), `new` callers a triple chosen by choose_filter_triple,
    // ss_searcher_set_filter3 any pair verbatim.  The result of a search never depends on them (src/lib.rs:375-378 asserts that
    // for every position).  What the DEVICE tests is this triple, except for a pair too far apart for any kernel (see
    // fill_problem in ss_scan.hip): first byte + two partners close behind it, the caller's far byte checked in memory.
    size_t fa = 0, fb = 0, fc = 0;
    size_t da = 0, db = 0, dc = 0;  // the triple the device tests (derive_device_filter): == fa, fb, fc unless the pair is too far apart
    size_t far = 0;                 // ... then: the caller's far byte (== fb), tested first when a candidate reaches memory; else 0
    uint32_t filter_gen = 0;        // bumped by every rewrite of the triple: census counts taken with an older triple are stale
    bool auto_filter = false;       // built by ss_searcher_new and not touched since: the library chose the triple and may choose again per haystack
    bool anchor_owned = false;      // ss_searcher_with_position with position >= 16: the caller's byte has a partner close in front of it that
                                    // the LIBRARY chose (choose_anchor) - the census may choose it again
    bool third_owned = false;       // the THIRD first-phase byte is the library's choice (every constructor; ss_searcher_set_filter3 given a
                                    // plain pair): the census may move it to the needle position that lets the fewest candidates through
    int variant = 0;          // tuning builds only (ss_searcher_set_variant / _set_grid); 0 = automatic
    int grid = 0;
    bool timing = false;
    // Searches in flight (>= 0), or -1 while ss_searcher_set_filter3 rewrites fa / fb / fc: the setter refuses
    // (SS_ERR_ARGUMENT) while a search runs instead of letting it read a half-written triple.
    mutable std::atomic<int> gate{0};
#ifdef SS_TEST_HOOKS
    mutable std::atomic<int> debug_fail_scans{0};       // the next k enqueue_scan calls fail (ss_debug_fail_next_scans)
#endif
    mutable std::atomic<bool> used_async{false};        // an *_async entry point may have left work behind (ss_searcher_free waits)
    mutable std::mutex mu;
    mutable std::condition_variable slot_cv;    // signalled when a flag slot is released
    mutable std::deque<ssh::PerDevice> per;     // deque: PerDevice pointers handed out stay valid as devices are added
};

namespace ssh {

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

// ---- ss_core.hip ------------------------------------------------------------------------------------------------------
int get_per_device(const ss_searcher *s, PerDevice **out);      // the searcher's state on the CURRENT device (uploads on first use)
int acquire_slot(const ss_searcher *s, PerDevice *p);
void release_slot(const ss_searcher *s, PerDevice *p, int k);
int next_epoch(PerDevice *p, int k);
void start_over(PerDevice *p, int k);

// Cost of one filter byte: a static, corpus-free rarity class, or - with a byte histogram of (a sample of) the haystack -
// 8 * log2(count + 1): summing costs then compares PRODUCTS of frequencies, which is what the candidate rate of a multi-byte
// filter is (bytes taken as independent).
struct ByteCost {
    int cost[256];
    explicit ByteCost(const uint64_t *hist);
    int operator()(uint8_t b) const { return cost[b]; }
};
size_t choose_third(const uint8_t *needle, size_t n, size_t fa, size_t fb, const ByteCost &cost, size_t other = ~(size_t)0);

// ---- ss_census.hip ----------------------------------------------------------------------------------------------------
// What the haystack has told this searcher so far (launch tuning only).  Scans of less than kCensusMinBytes are not asked.
constexpr size_t kCensusMinBytes = (size_t)256 << 20;
struct LaunchHints {
    bool have_counts;           // the census of (searcher, haystack) is in:
    int workgroups_per_cu;      //   four, five or six
    int tiles_per_workgroup;    //   0: the launch's own choice (one below 2 GiB, two from there); 1 or 2: by the candidate density
    bool sparse_pair;           //   the first two filter bytes alone rarely match (cross-lane kernels: no third byte needed)
    bool have_triple;           // filter bytes chosen for this haystack (from its histogram, or from the census's own match counts):
    size_t tri[3];              //   first <= second, third, all within 15 of the first
    bool have_order;            // the second level's schedule ordered by what the census measured (for the triple in force):
    uint32_t norder;            //   as Problem::norder / order_idx / order_val
    uint64_t order_idx[2], order_val[2];
};
// Looks up - and, when nothing is known and the stream is not being captured, starts - the census and the histogram sampling in
// front of the caller's scan on `st`.  Never waits.
void launch_hints(const ss_searcher *s, PerDevice *pd, const void *d_hay, size_t len, hipStream_t st, LaunchHints *out);

// ---- ss_scan.hip ------------------------------------------------------------------------------------------------------
// What a launch needs to know about a Problem besides the Problem itself.
struct ProblemShape {
    size_t position, position3;     // the second / third filter byte, relative to the first (ordered by dword: q3 <= Q)
    size_t fa;                      // index of the first filter byte
    bool one_byte;
};
// The Problem of (searcher, haystack): everything but the sink-side fields (epoch, host_flag, completion word), which the caller
// sets.  Preconditions: 1 <= n <= len.
// `triple` != nullptr: filter bytes chosen for this haystack instead of the searcher's own (all within 16 bytes of the first).
void fill_problem(const ss_searcher *s, const uint8_t *d_needle, const void *d_hay, size_t len, uint64_t find_base, ss::Problem *out,
                  ProblemShape *shape, const size_t *triple = nullptr);
bool autotune_enabled();          // ss_set_autotune / SLICESLICE_AUTOTUNE (ss_census.hip): off = static choices only, no sampling kernels
// Builds the Problem for (hay, len) and enqueues the scan.  find == false: *d_sink is an int flag, set to `epoch` by the wave that
// finds the needle, never cleared.  find == true: *d_sink is a uint64, atomicMin'ed with find_base + offset of every match the grid
// sees (the leftmost one survives).  Preconditions: 1 <= n <= len.  done_slot >= 0: the call owns flag slot `done_slot` and would
// like to wait on the slot's completion word instead of the stream; granted (*used_done = true) for small grids.
int enqueue_scan(const ss_searcher *s, PerDevice *pd, const void *d_hay, size_t len, hipStream_t st, void *d_sink, bool find = false,
                 uint64_t find_base = 0, int *host_flag = nullptr, int epoch = 1, int done_slot = -1, bool *used_done = nullptr);
void timer_forget(const ss_searcher *s);                         // ss_searcher_free: the calling thread's timing record
int thread_last_kernel_ms(uint64_t uid, int dev, float *ms);     // the calling thread's latest timed scan of searcher `uid` on `dev` (< 0: its latest)

// Scans too large for the workgroup count of the completion word still answer through a pinned word when the scan is short enough
// to be waited for by spinning: a one-lane kernel behind the scan (behind the all-reduce, for a sharded search) stores the word.
constexpr double kSpinMaxEstimateUs = 20000.0;
constexpr double kSpinMinEstimateUs = 0.5;
inline double scan_estimate_us(size_t len) { return (double)len / 7.0e6; }          // 7 TB/s
bool spin_wait_enabled();                                                           // SLICESLICE_SPIN_WAIT != 0
bool spin_for_word(const long long *word, int epoch, double estimate_us, int *found);
bool spin_for_shard_word(const long long *word, int epoch, double estimate_us, int *found, int *failed);
// one-lane kernels behind a scan / an all-reduce (aux_kernels.hpp): answer word epoch << 1 | found (pair: epoch << 2 | failed << 1 |
// found), and find()'s minimum into its pinned mirror (pair: {offset, status}; else the device word is re-armed)
hipError_t launch_signal_flag(hipStream_t st, const int *d_flag, int epoch, long long *h_word, int pair);
hipError_t launch_publish_best(hipStream_t st, uint64_t *d_best, uint64_t *h_best, int pair);

#ifdef SS_TEST_HOOKS
bool cross_exit_enabled();        // SLICESLICE_CROSS_EXIT != 0 (hooks builds: the relay's effect is measured by a test)
#else
inline bool cross_exit_enabled() { return true; }
#endif

}  // namespace ssh
