// ss_batched.hip - many (needle, haystack) problems per launch.
//   ss_search_batched / ss_find_batched   BASELINE.json config 5 (4,096 needles x 1 MiB haystacks as one grid), and the loop of
//                                         /root/reference/bench/benches/i386.rs:252-256 (4,585 needles, one text) as one call
//   ss_batch_plan_*                       the same with the per-problem set-up done ONCE (the reference builds its searchers
//                                         once, i386.rs:246-250, and times only the searches)
//   ss_search_pairs                       row f4 of SURVEY.md 8f: the short-haystack loop of i386.rs:118-129, one lane per problem
// Semantics per problem are those of ss_search_device / ss_find_device.  There is no CPU search path in this file.
#include "ss_internal.hpp"

#include "batched_kernels.hpp"

namespace ssh {
namespace {

// Total workgroups aimed at, per CU.  Measured in one process on one buffer (tools/batch_tune.py, profiles/r03/batch_tune_*.jsonl;
// kernel time: tools/shape_trace.py under rocprofv3): 96 per CU is best or within 1 % of the best on every shape at 1 GiB in
// total (1 / 64 / 256 / 1,024 problems: 149-151 us = 7.1-7.2 TB/s of kernel time; 160 per CU 150-153 us, 256 per CU 161-163 us,
// 512 per CU 190 us - surplus workgroups cost 0.3-0.4 us of a slot each) and for the i386 loop (0.151 ms; 160: 0.17, 256: 0.77);
// at 4 GiB in 4,096 problems 256 per CU is 2 % faster (583 vs 597 us), which is not worth the rest.
constexpr unsigned kPlanWgsPerCu = 96;
constexpr uint32_t kPlanMinTiles = 2;       // shortest slice worth a workgroup, in 16 KiB tiles
// ... and in a plan, where a problem scanned by ONE workgroup is published by that workgroup (no state word, no second kernel), and
// where the lengths are known when the grid is sized (ss_batch_plan_create reads them back from the plan kernel): about kPlanTilesPerWg tiles per workgroup - 6,500 workgroups
// for 1 GiB, 26,000 for 4 GiB - was best or within 1 % of the best on every cut (tools/batch_probe.py under SLICESLICE_BATCH_WGS x
// SLICESLICE_BATCH_MIN_TILES, profiles/r04/batch_plan_sweep.jsonl).
constexpr uint32_t kPlanMinTilesCounted = 8;
constexpr uint32_t kPlanTilesPerWg = 10;
constexpr uint32_t kPlanTilesPerWgLong = 4, kPlanMinTilesLong = 4;   // problems of more than 80 tiles (1.25 MiB): see ss_batch_plan_create

// Descriptor scratch of the unplanned calls: one grow-only device buffer per (device, stream), kept for the life of the process.
// Launches on one stream execute in order, so a buffer that belongs to the stream can be reused by the next call on that
// stream without any wait; the entry's mutex keeps the two launches of one call adjacent when several threads share a
// stream.  (hipMallocAsync / hipFreeAsync per call did the same job at 5-10 us of extra latency per call.)  At most
// kPlanScratchEntries streams per device are remembered; beyond that the least recently used entry is freed (hipFree waits
// for the device, so nothing that still reads the buffer can be running).  None of this can be captured into a hipGraph - a
// graph would bake in a buffer that a later call frees - so a capturing stream is refused; graphs take an ss_batch_plan.
constexpr int kPlanScratchEntries = 32;
struct PlanScratch {
    hipStream_t stream = nullptr;
    bool used = false;
    ss::BatchDesc *buf = nullptr;
    size_t cap = 0;             // descriptors (and as many cold records behind them)
    uint64_t stamp = 0;
    std::mutex mu;              // held across the plan + scan launches of one call
};
struct PlanTable {
    std::mutex mu;
    PlanScratch entry[kPlanScratchEntries];
    uint64_t clock = 0;
};
PlanTable *plan_tables()        // never destroyed (a call may come from a thread that outlives main)
{
    static PlanTable *const t = new PlanTable[kMaxDevices];
    return t;
}

// Returns the stream's entry with its mutex LOCKED and room for `count` descriptors, or nullptr (no memory).
PlanScratch *plan_scratch_acquire(int dev, hipStream_t st, size_t count)
{
    if (dev < 0 || dev >= kMaxDevices) return nullptr;
    PlanTable &tab = plan_tables()[dev];
    std::unique_lock<std::mutex> table(tab.mu);
    PlanScratch *e = nullptr, *victim = nullptr;
    for (auto &c : tab.entry) {
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
    e->stamp = ++tab.clock;
    table.unlock();
    if (e->cap < count) {
        const size_t want = count < 4096 ? 4096 : count + count / 2;
        if (e->buf) (void)hipFree(e->buf);
        e->buf = nullptr;
        e->cap = 0;
        if (hipMalloc((void **)&e->buf, want * (sizeof(ss::BatchDesc) + sizeof(ss::BatchCold))) != hipSuccess) {
            (void)hipGetLastError();
            e->buf = nullptr;
            e->mu.unlock();
            return nullptr;
        }
        e->cap = want;
    }
    return e;
}

// Unused dynamic LDS that leaves room for exactly `occ` workgroups of kBlock threads per CU (160 KiB of LDS; the kernels'
// needle slices are static LDS).  The batched kernels (73-80 VGPRs) are held to four workgroups per CU like the single-problem
// scan on random bytes (ss_scan.hip, pick_variant).
uint32_t batch_lds_pad()
{
    int occ = 4;
#ifdef SS_TEST_HOOKS
    if (const char *e = getenv("SLICESLICE_BATCH_OCC")) { const int v = atoi(e); if (v >= 1 && v <= 8) occ = v; }
#endif
    const uint32_t per = (160u * 1024u) / (uint32_t)occ, fixed = ss::kWavesPerBlock * ss::kNeedleLds;
    uint32_t pad = per > fixed + 2048 ? ((per - fixed - 1024) & ~1023u) : 0;
    if (pad > 64u * 1024u - fixed) pad = 64u * 1024u - fixed;
    return pad;
}

int fill_batch_args(ss::BatchArgs *a, const void *d_haystacks, const uint64_t *d_hay_begin, const uint64_t *d_hay_end,
                    const void *d_needles, const uint64_t *d_needle_begin, const uint64_t *d_needle_end, const uint64_t *d_position)
{
    if (!d_hay_begin || !d_hay_end || !d_needle_begin || !d_needle_end) return fail(SS_ERR_ARGUMENT, "NULL argument");
    a->haystacks = static_cast<const uint8_t *>(d_haystacks);
    a->hay_begin = d_hay_begin;
    a->hay_end = d_hay_end;
    a->needles = static_cast<const uint8_t *>(d_needles);
    a->needle_begin = d_needle_begin;
    a->needle_end = d_needle_end;
    a->position = d_position;
    a->found = nullptr;
    a->best = nullptr;
    return SS_OK;
}

// The grid of a batch: the haystack lengths live on the device, so it is sized from the problem COUNT - kPlanWgsPerCu
// workgroups per CU in total, i.e. `slices` workgroups per problem.  The price of being wrong is small: a surplus slice costs
// one scalar round trip, and a slice as short as kPlanMinTiles tiles is worth a workgroup because nothing but that load stands
// in front of its first haystack byte.
struct BatchShape {
    uint32_t slices, min_tiles;
};
int batch_shape(int dev, size_t count, BatchShape *out, bool counted = false)
{
    DeviceInfo di;
    if (int rc = device_info(dev, &di)) return rc;
    if (count > 0x3fffffffull) return fail(SS_ERR_ARGUMENT, "too many problems");
    uint64_t wg_target = (uint64_t)di.cus * kPlanWgsPerCu;
    uint32_t min_tiles = counted ? kPlanMinTilesCounted : kPlanMinTiles;
#ifdef SS_TEST_HOOKS
    if (const char *e = getenv("SLICESLICE_BATCH_WGS")) { const long v = atol(e); if (v > 0) wg_target = (uint64_t)v; }
    if (const char *e = getenv("SLICESLICE_BATCH_MIN_TILES")) { const long v = atol(e); if (v > 0) min_tiles = (uint32_t)v; }
#endif
    uint64_t slices = (wg_target + count - 1) / count;
    if (slices < 1) slices = 1;
    while (slices > 1 && (uint64_t)count * slices > 0x7fffffffull) --slices;   // gridDim.x
    out->slices = (uint32_t)slices;
    out->min_tiles = min_tiles;
    return SS_OK;
}

// `cls` (may be null): the 256 rarity classes batch_sample_kernel derived from the haystacks' own bytes - the filter bytes are
// chosen by them instead of the static, corpus-free table.
hipError_t launch_plan_kernel(const ss::BatchArgs &a, size_t count, ss::BatchDesc *descs, const BatchShape &sh, hipStream_t st,
                              ss::PlanStats *stats = nullptr, const uint8_t *cls = nullptr, ss::BatchCold *colds = nullptr)
{
    const uint64_t pblocks = ((uint64_t)count + ss::kBlock - 1) / ss::kBlock;
    ss::batch_plan_kernel<<<dim3((unsigned)pblocks), dim3(ss::kBlock), 0, st>>>(a, (uint64_t)count, descs, sh.slices, sh.min_tiles,
                                                                               ss::kWavesPerBlock * 4, stats, cls, colds);
    return hipGetLastError();
}

// Row f3 of SURVEY.md 8f for the UNPLANNED calls, without a wait and without a cost for callers that come once: per device a few
// remembered batches (haystack blob, range array, count).  The first call that names a batch only leaves its name; the second has
// batch_sample_kernel launched in front of it, on its stream (4 MiB read at most); calls after that use the classes once the
// kernel's tag has arrived in pinned memory - the host never waits for it.  Sampled again every kClassRefreshEvery uses (a blob
// may be refilled in place).  An entry that is evicted while a plan kernel of another stream still reads its classes hands that
// kernel another batch's classes: a slower choice of filter bytes at worst, never another answer.
constexpr int kClassEntries = 4;
constexpr uint32_t kClassRefreshEvery = 1024;
struct ClassEntry {
    const void *hay = nullptr, *begin = nullptr;
    size_t count = 0;
    uint32_t state = 0;             // 0 = empty, 1 = named once, 2 = sampling launched (tag), 3 = classes in
    uint32_t uses = 0;
    unsigned long long tag = 0;
    uint64_t stamp = 0;
    ss::BatchClasses *mem = nullptr;
};
struct ClassTable {
    std::mutex mu;
    ClassEntry e[kClassEntries];
    int pending = -1;
    unsigned long long tag = 0;
    uint64_t clock = 0;
    unsigned long long *h_tag = nullptr;    // pinned
    bool broken = false;
};
ClassTable *class_tables()          // never destroyed (a call may come from a thread that outlives main)
{
    static ClassTable *const t = new ClassTable[kMaxDevices];
    return t;
}

// The classes of this batch if they are in; launches the sampling in front of the caller's kernels when the batch has been seen
// before and nothing is in flight on the device.
const uint8_t *batch_classes(int dev, const ss::BatchArgs &a, size_t count, hipStream_t st)
{
    if (dev < 0 || dev >= kMaxDevices || !autotune_enabled()) return nullptr;      // (ss_set_autotune(0): the static classes, no sampling)
#ifdef SS_TEST_HOOKS
    if (const char *v = getenv("SLICESLICE_BATCH_STATIC_CLASSES")) { if (atoi(v) != 0) return nullptr; }
#endif
    ClassTable &t = class_tables()[dev];
    std::unique_lock<std::mutex> lock(t.mu, std::try_to_lock);
    if (!lock.owns_lock() || t.broken) return nullptr;
    if (t.pending >= 0 && __atomic_load_n(t.h_tag, __ATOMIC_ACQUIRE) == t.e[t.pending].tag) {
        t.e[t.pending].state = 3;
        t.pending = -1;
    }
    ClassEntry *hit = nullptr, *victim = nullptr;
    for (auto &c : t.e) {
        if (c.state != 0 && c.hay == a.haystacks && c.begin == a.hay_begin && c.count == count) hit = &c;
        if ((int)(&c - t.e) != t.pending && (!victim || c.stamp < victim->stamp)) victim = &c;     // (never the entry being sampled)
    }
    if (!hit) {
        victim->hay = a.haystacks;
        victim->begin = a.hay_begin;
        victim->count = count;
        victim->state = 1;
        victim->uses = 0;
        victim->stamp = ++t.clock;
        return nullptr;
    }
    hit->stamp = ++t.clock;
    const uint8_t *have = hit->state == 3 ? hit->mem->cls : nullptr;
    if (hit->state == 2) return nullptr;
    if (hit->state == 3 && ++hit->uses % kClassRefreshEvery != 0) return have;
    if (t.pending >= 0) return have;
    if (!t.h_tag) {
        if (hipHostMalloc((void **)&t.h_tag, sizeof(unsigned long long), hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            t.broken = true;
            return have;
        }
        *t.h_tag = 0;
    }
    if (!hit->mem) {
        // (zeroed on the call's stream, in front of the sampling kernel: no null-stream work inside a search call)
        hipError_t e = hipMalloc((void **)&hit->mem, sizeof(ss::BatchClasses));
        if (e == hipSuccess) e = hipMemsetAsync(hit->mem, 0, sizeof(ss::BatchClasses), st);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            hit->mem = nullptr;
            t.broken = true;
            return have;
        }
    }
    if (++t.tag == 0) t.tag = 1;
    ss::batch_sample_kernel<<<dim3(ss::kPlanSampleBlocks), dim3(ss::kBlock), 0, st>>>(a.haystacks, a.hay_begin, a.hay_end, (uint64_t)count, hit->mem,
                                                                                  t.h_tag, t.tag);
    if (hipGetLastError() != hipSuccess) return have;
    // (a refresh overwrites the classes in place while this very call's plan kernel - behind it on the same stream - reads them:
    // stream order makes that the NEW classes; a call on another stream may see a mix of old and new: any classes are valid)
    hit->tag = t.tag;
    if (hit->state != 3) hit->state = 2;                        // (a refresh leaves the old classes in use until the new ones are in)
    t.pending = (int)(hit - t.e);
    return have;
}

// batch_plan_kernel turns the range arrays into one 64-byte descriptor per problem and writes the initial outputs (no memset
// launch), the scan grid's workgroups then start with one scalar load.
int launch_batched(const ss::BatchArgs &a, size_t count, hipStream_t st)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(SS_ERR_ARGUMENT, "ss_search_batched / ss_find_batched keep per-stream scratch and cannot be captured into a hipGraph: "
                                     "build an ss_batch_plan outside the capture and capture ss_batch_plan_run");
    (void)hipGetLastError();
    BatchShape sh;
    if (int rc = batch_shape(dev, count, &sh)) return rc;
    PlanScratch *ps = plan_scratch_acquire(dev, st, count);
    if (!ps) return fail(SS_ERR_NOMEM, "no device memory for %zu problem descriptors", count);
    ss::BatchDesc *descs = ps->buf;
    ss::BatchCold *colds = reinterpret_cast<ss::BatchCold *>(ps->buf + ps->cap);        // (behind the descriptors: ColdInCall)
    const uint8_t *cls = batch_classes(dev, a, count, st);
    hipError_t e = launch_plan_kernel(a, count, descs, sh, st, nullptr, cls, colds);
    if (e == hipSuccess) {
        const dim3 grid((unsigned)((uint64_t)count * sh.slices));
        if (a.best)
            ss::scan_batched_plan_kernel<4, true, false><<<grid, dim3(ss::kBlock), batch_lds_pad(), st>>>(a, descs, (uint32_t)count, sh.slices, colds,
                                                                                                          nullptr, nullptr, nullptr, 0u);
        else
            ss::scan_batched_plan_kernel<4, false, false><<<grid, dim3(ss::kBlock), batch_lds_pad(), st>>>(a, descs, (uint32_t)count, sh.slices, colds,
                                                                                                           nullptr, nullptr, nullptr, 0u);
        e = hipGetLastError();
    }
    ps->mu.unlock();
    if (e != hipSuccess) return fail(SS_ERR_HIP, "batched launch: %s", hipGetErrorString(e));
    return SS_OK;
}

}  // namespace
}  // namespace ssh

using namespace ssh;

// The plan's own memory: descriptors | cold parts (64 bytes each, like the descriptors) | the problems' state words (ss::PlanState, 64
// bytes each: one word per run parity) | the plan kernel's PlanStats (64 bytes) | the control word of the runs (ss::PlanCtl, 64 bytes:
// the latest run's identity and parity, the tallies) | the sampling's counters and the rarity classes of the haystacks' bytes
// (ss::BatchClasses).
struct ss_batch_plan {
    int dev = 0;
    size_t count = 0;
    bool find = false;
    ss::BatchArgs args;
    BatchShape shape = {1, 1};
    // Plans of LONG problems hold a second layout.  Their workgroups scan a problem round robin, side by side - the fastest way
    // through haystacks that do not hold the needle (consecutive addresses in flight), and one that gains nothing when they do:
    // four tiles per workgroup leave no room for an early exit.  Eight contiguous runs per problem (the slice-major layout) are 6 %
    // slower without matches and 2 to 25 times faster with them (256 x 4 MiB, every needle present at the start / in the middle:
    // 0.014 / 0.087 ms against 0.196 / 0.160; profiles/r05/plan_layouts.jsonl) - the later runs of a found problem leave at their
    // entry poll.  Which of the two a plan's problems want is not known when it is made, and is known after its first run: the
    // first finder of every problem counts it into the run's tally, the first workgroup of the NEXT run stores that count to pinned
    // memory, and a later run - which never waits for it - takes the contiguous runs when at least an eighth of the problems were
    // found.  Deterministic for given inputs; no result depends on it.  (A run captured into a hipGraph keeps the layout - and the run
    // number - it was captured with: the choice is the host's, made when the launch is issued.)
    bool has_alt = false;
    BatchShape shape_alt = {1, 1};
    unsigned long long *h_tally = nullptr;          // pinned: run << 32 | found problems of the run before it
    mutable uint32_t runs = 0;
    ss::BatchDesc *mem_alt = nullptr;               // the second layout's descriptors (plans that hold two)
    uint8_t *mem = nullptr;
    static constexpr size_t kPerProblem = sizeof(ss::BatchDesc) + sizeof(ss::BatchCold) + sizeof(ss::PlanState);
    ss::BatchDesc *descs() const { return reinterpret_cast<ss::BatchDesc *>(mem); }
    ss::BatchDesc *descs_alt() const { return mem_alt; }
    ss::BatchCold *colds() const { return reinterpret_cast<ss::BatchCold *>(mem + count * sizeof(ss::BatchDesc)); }
    ss::PlanState *states() const { return reinterpret_cast<ss::PlanState *>(mem + count * (sizeof(ss::BatchDesc) + sizeof(ss::BatchCold))); }
    ss::PlanStats *stats() const { return reinterpret_cast<ss::PlanStats *>(mem + count * kPerProblem); }
    ss::PlanCtl *ctl() const { return reinterpret_cast<ss::PlanCtl *>(mem + count * kPerProblem + 64); }
    ss::BatchClasses *classes() const { return reinterpret_cast<ss::BatchClasses *>(mem + count * kPerProblem + 128); }
};

extern "C" {

int ss_search_batched(const void *d_haystacks, const uint64_t *d_hay_begin, const uint64_t *d_hay_end,
                      const void *d_needles, const uint64_t *d_needle_begin, const uint64_t *d_needle_end,
                      const uint64_t *d_position, size_t count, void *hip_stream, int *d_found)
{
    if (count == 0) return SS_OK;
    if (!d_found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    ss::BatchArgs a;
    if (int rc = fill_batch_args(&a, d_haystacks, d_hay_begin, d_hay_end, d_needles, d_needle_begin, d_needle_end, d_position)) return rc;
    a.found = d_found;
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
    if (int rc = fill_batch_args(&a, d_haystacks, d_hay_begin, d_hay_end, d_needles, d_needle_begin, d_needle_end, nullptr)) return rc;
    a.best = d_position;
    return launch_batched(a, count, static_cast<hipStream_t>(hip_stream));
}

int ss_batch_plan_create(const void *d_haystacks, const uint64_t *d_hay_begin, const uint64_t *d_hay_end, const void *d_needles,
                         const uint64_t *d_needle_begin, const uint64_t *d_needle_end, const uint64_t *d_position, size_t count,
                         int find, void *hip_stream, ss_batch_plan **out)
{
    if (!out) return fail(SS_ERR_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (count == 0) return fail(SS_ERR_ARGUMENT, "a plan needs at least one problem");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(SS_ERR_ARGUMENT, "ss_batch_plan_create allocates and waits: call it outside the stream capture");
    (void)hipGetLastError();
    ss_batch_plan *p = new (std::nothrow) ss_batch_plan;
    if (!p) return fail(SS_ERR_NOMEM, "out of memory");
    p->count = count;
    p->find = find != 0;
    int rc = fill_batch_args(&p->args, d_haystacks, d_hay_begin, d_hay_end, d_needles, d_needle_begin, d_needle_end, find ? nullptr : d_position);
    hipError_t e = hipSuccess;
    if (rc == SS_OK && (e = hipGetDevice(&p->dev)) != hipSuccess) rc = fail(SS_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    if (rc == SS_OK) rc = batch_shape(p->dev, count, &p->shape, true);
    if (rc == SS_OK) {
        const size_t bytes = count * ss_batch_plan::kPerProblem + 128 + sizeof(ss::BatchClasses);
        if ((e = hipMalloc((void **)&p->mem, bytes)) != hipSuccess)
            rc = fail(e == hipErrorOutOfMemory ? SS_ERR_NOMEM : SS_ERR_HIP, "plan memory (%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    if (rc == SS_OK) {
        // The descriptors (the plan kernel writes no outputs here: args.found and
        // args.best are both null).  The lengths live on the device, so the first pass runs with the grid guessed from the problem
        // count and reports what it saw; the host then sizes the slices for about kPlanTilesPerWg tiles per workgroup (bounded:
        // one huge haystack among many short ones must not multiply everybody's surplus slices) and, if that changes anything,
        // has the descriptors rebuilt.  The runs launch as many slices per problem as the busiest problem uses.
        // Which needle bytes the scan filters on is decided by how rare they are IN THESE HAYSTACKS: a sampled histogram first
        // (4 MiB read at most; batched_kernels.hpp, batch_sample_kernel).
        ss::PlanStats seen = {0, 0, 0};
        e = hipSuccess;
        const uint8_t *cls = autotune_enabled() ? p->classes()->cls : nullptr;      // (ss_set_autotune(0): the static classes)
#ifdef SS_TEST_HOOKS
        if (const char *v = getenv("SLICESLICE_BATCH_STATIC_CLASSES")) { if (atoi(v) != 0) cls = nullptr; }   // A/B: the static table
#endif
        if (e == hipSuccess && cls) {
            e = hipMemsetAsync(p->classes(), 0, sizeof(ss::BatchClasses), st);
            if (e == hipSuccess) {
                ss::batch_sample_kernel<<<dim3(ss::kPlanSampleBlocks), dim3(ss::kBlock), 0, st>>>(p->args.haystacks, p->args.hay_begin, p->args.hay_end,
                                                                                              (uint64_t)count, p->classes(), nullptr, 0ull);
                e = hipGetLastError();
            }
        }
        for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {
            e = hipMemsetAsync(p->stats(), 0, 64, st);
            if (e == hipSuccess) e = launch_plan_kernel(p->args, count, p->descs(), p->shape, st, p->stats(), cls);
            if (e == hipSuccess) e = hipMemcpyAsync(&seen, p->stats(), sizeof(seen), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess || pass == 1 || seen.max_slices == 0) break;
#ifdef SS_TEST_HOOKS
            if (getenv("SLICESLICE_BATCH_WGS")) break;                         // tuning: the grid is what the variable says
#endif
            // problems short enough for the slice-major layout (contiguous runs): about kPlanTilesPerWg tiles per workgroup; longer
            // ones are scanned round robin by workgroups side by side, like ONE haystack by the single-problem kernel, and like
            // there short workgroups win: kPlanTilesPerWgLong
            const bool longp = seen.max_tiles > ss::kPlanSliceMajorMax * kPlanTilesPerWg;
            const uint32_t per = longp ? kPlanTilesPerWgLong : kPlanTilesPerWg, min_tiles = longp ? kPlanMinTilesLong : kPlanMinTilesCounted;
            const uint64_t want = ((uint64_t)seen.max_tiles + per - 1) / per;
            const uint64_t fair = ((seen.total_tiles + per - 1) / per + count - 1) / count;
            uint64_t slices = std::min<uint64_t>(want, 4 * fair);
            slices = std::max<uint64_t>(1, slices);
            while (slices > 1 && (uint64_t)count * slices > 0x7fffffffull) --slices;
            if (slices == p->shape.slices && min_tiles == p->shape.min_tiles) break;
            p->shape.slices = (uint32_t)slices;
            p->shape.min_tiles = min_tiles;
        }
        if (e == hipSuccess) {
            const uint32_t most = seen.max_slices;
            p->shape.slices = most < 1 ? 1 : (most < p->shape.slices ? most : p->shape.slices);
            // the cold part of every problem (second-level schedule, the needle's dwords): once, here, instead of by every wave that
            // meets a candidate (batched_kernels.hpp, BatchCold)
            ss::batch_cold_kernel<<<dim3((unsigned)((count + ss::kBlock - 1) / ss::kBlock)), dim3(ss::kBlock), 0, st>>>(p->args, p->descs(), (uint64_t)count,
                                                                                                                        p->colds(), cls, p->find ? 1 : 0);
            e = hipGetLastError();
            // the runs' state: every problem's two state words idle (bool: 0, find: all ones), the control word naming no run
            if (e == hipSuccess) e = hipMemsetAsync(p->states(), p->find ? 0xFF : 0, count * sizeof(ss::PlanState), st);
            if (e == hipSuccess) e = hipMemsetAsync(p->ctl(), 0, sizeof(ss::PlanCtl), st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            // the second layout (see ss_batch_plan): for plans whose problems are scanned round robin and are numerous enough for
            // eight runs each to fill the device
            bool alt_ok = p->shape.slices > ss::kPlanSliceMajorMax && (uint64_t)count * ss::kPlanSliceMajorMax >= 1024 && autotune_enabled();
#ifdef SS_TEST_HOOKS
            if (getenv("SLICESLICE_BATCH_WGS")) alt_ok = false;
            if (const char *v = getenv("SLICESLICE_PLAN_ONE_LAYOUT")) { if (atoi(v) != 0) alt_ok = false; }
#endif
            if (e == hipSuccess && alt_ok && hipMalloc((void **)&p->mem_alt, count * sizeof(ss::BatchDesc)) != hipSuccess) {
                (void)hipGetLastError();
                p->mem_alt = nullptr;
                alt_ok = false;                                    // (no memory for it: one layout)
            }
            if (e == hipSuccess && alt_ok) {
                ss::PlanStats alt = {0, 0, 0};
                p->shape_alt.slices = ss::kPlanSliceMajorMax;
                p->shape_alt.min_tiles = kPlanMinTilesCounted;
                e = hipMemsetAsync(p->stats(), 0, 64, st);
                if (e == hipSuccess) e = launch_plan_kernel(p->args, count, p->descs_alt(), p->shape_alt, st, p->stats(), cls);
                if (e == hipSuccess) e = hipMemcpyAsync(&alt, p->stats(), sizeof(alt), hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                if (e == hipSuccess && alt.max_slices > 1 &&
                    hipHostMalloc((void **)&p->h_tally, sizeof(unsigned long long), hipHostMallocPortable) == hipSuccess) {
                    *p->h_tally = 0;
                    p->shape_alt.slices = alt.max_slices < p->shape_alt.slices ? alt.max_slices : p->shape_alt.slices;
                    p->has_alt = true;
                } else {
                    (void)hipGetLastError();
                }
            }
        }
        if (e != hipSuccess)
            rc = fail(e == hipErrorOutOfMemory ? SS_ERR_NOMEM : SS_ERR_HIP, "plan set-up: %s", hipGetErrorString(e));
    }
    if (rc != SS_OK) {
        (void)hipGetLastError();
        (void)hipFree(p->mem);
        (void)hipFree(p->mem_alt);
        if (p->h_tally) (void)hipHostFree(p->h_tally);
        delete p;
        return rc;
    }
    *out = p;
    return SS_OK;
}

int ss_batch_plan_run(const ss_batch_plan *p, void *hip_stream, void *d_out)
{
    if (!p || !d_out) return fail(SS_ERR_ARGUMENT, "NULL argument");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != p->dev) return fail(SS_ERR_ARGUMENT, "the plan was made on device %d, the current device is %d", p->dev, dev);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    ss::BatchArgs a = p->args;
    // which layout: the contiguous runs when the latest tally that has arrived says an eighth of the problems were found (see
    // ss_batch_plan; runs of one plan are ordered one behind the other by contract, so `runs` needs no atomics)
    bool alt = false;
    if (p->has_alt) {
        const unsigned long long t = __atomic_load_n(p->h_tally, __ATOMIC_RELAXED);
        alt = (t >> 32) != 0 && (uint64_t)(uint32_t)t * 8 >= (uint64_t)p->count;
    }
    const ss::BatchDesc *descs = alt ? p->descs_alt() : p->descs();
    const BatchShape &sh = alt ? p->shape_alt : p->shape;
    const dim3 grid((unsigned)((uint64_t)p->count * sh.slices));
    // ONE launch: problems scanned by a single workgroup are published by it, problems scanned by several raise the plan's state
    // word of this run's parity and write the caller's output behind it (batched_kernels.hpp, PlanCtl) - no publish kernel, nothing
    // initialised by the host, so a run can be captured into a hipGraph and replayed.
    const uint32_t run = ++p->runs == 0 ? ++p->runs : p->runs;
    unsigned long long *h_tally = p->has_alt ? p->h_tally : nullptr;
    // (a plan whose every problem is scanned by one workgroup - one slice - launches the instantiation without the run machinery)
    const bool multi = sh.slices > 1;
    const dim3 block(ss::kBlock);
    const uint32_t n = (uint32_t)p->count, pad = batch_lds_pad();
    if (p->find) {
        a.best = static_cast<uint64_t *>(d_out);
        if (multi) ss::scan_batched_plan_kernel<4, true, true, true><<<grid, block, pad, st>>>(a, descs, n, sh.slices, p->colds(), p->ctl(), p->states(), h_tally, run);
        else ss::scan_batched_plan_kernel<4, true, true, false><<<grid, block, pad, st>>>(a, descs, n, sh.slices, p->colds(), nullptr, nullptr, nullptr, run);
    } else {
        a.found = static_cast<int *>(d_out);
        if (multi) ss::scan_batched_plan_kernel<4, false, true, true><<<grid, block, pad, st>>>(a, descs, n, sh.slices, p->colds(), p->ctl(), p->states(), h_tally, run);
        else ss::scan_batched_plan_kernel<4, false, true, false><<<grid, block, pad, st>>>(a, descs, n, sh.slices, p->colds(), nullptr, nullptr, nullptr, run);
    }
    HIP_TRY(hipGetLastError());
    return SS_OK;
}

#ifdef SS_TEST_HOOKS
int ss_debug_batch_classes(const void *d_haystacks, const uint64_t *d_hay_begin, size_t count, uint32_t *state, uint8_t cls[256])
{
    if (!state || !cls) return fail(SS_ERR_ARGUMENT, "NULL argument");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    *state = 0;
    ClassTable &t = class_tables()[dev];
    std::lock_guard<std::mutex> lock(t.mu);
    if (t.pending >= 0 && __atomic_load_n(t.h_tag, __ATOMIC_ACQUIRE) == t.e[t.pending].tag) {
        t.e[t.pending].state = 3;
        t.pending = -1;
    }
    for (auto &c : t.e) {
        if (c.state == 0 || c.hay != d_haystacks || c.begin != d_hay_begin || c.count != count) continue;
        *state = c.state;
        if (c.state == 3) HIP_TRY(hipMemcpy(cls, c.mem->cls, 256, hipMemcpyDeviceToHost));
    }
    return SS_OK;
}

int ss_debug_plan_layout(const ss_batch_plan *p, uint32_t out[5])
{
    if (!p || !out) return fail(SS_ERR_ARGUMENT, "NULL argument");
    const unsigned long long t = p->has_alt ? __atomic_load_n(p->h_tally, __ATOMIC_RELAXED) : 0ull;
    out[0] = p->has_alt ? 1u : 0u;
    out[1] = p->shape.slices;
    out[2] = p->has_alt ? p->shape_alt.slices : 0u;
    out[3] = (uint32_t)t;                                            // problems found in the latest tallied run
    out[4] = p->has_alt && (t >> 32) != 0 && (uint64_t)(uint32_t)t * 8 >= (uint64_t)p->count ? 1u : 0u;   // the next run takes the second layout
    return SS_OK;
}

int ss_debug_plan_cold(const ss_batch_plan *p, size_t problem, uint32_t out[14])
{
    if (!p || !out || problem >= p->count) return fail(SS_ERR_ARGUMENT, "bad argument");
    ss::BatchCold c;
    HIP_TRY(hipMemcpy(&c, p->colds() + problem, sizeof c, hipMemcpyDeviceToHost));
    out[0] = c.norder;
    out[1] = c.exact_len;
    for (int t = 0; t < 2; ++t) {
        out[2 + 2 * t] = (uint32_t)c.order_idx[t];
        out[3 + 2 * t] = (uint32_t)(c.order_idx[t] >> 32);
        out[6 + 2 * t] = (uint32_t)c.order_val[t];
        out[7 + 2 * t] = (uint32_t)(c.order_val[t] >> 32);
    }
    for (int j = 0; j < 4; ++j) out[10 + j] = c.tail16[j];
    return SS_OK;
}

int ss_debug_plan_filter(const ss_batch_plan *p, size_t problem, uint32_t out[5])
{
    if (!p || !out || problem >= p->count) return fail(SS_ERR_ARGUMENT, "bad argument");
    ss::BatchDesc d;
    HIP_TRY(hipMemcpy(&d, p->descs() + problem, sizeof d, hipMemcpyDeviceToHost));
    const uint32_t r = (d.shifts >> 4) & 3, q = (d.shifts >> 6) & 3, r3 = (d.shifts >> 8) & 3, q3 = (d.shifts >> 10) & 3;
    const bool scanned = (d.per >> 32) != 0;
    out[0] = scanned ? (uint32_t)d.anchor : 0;
    out[1] = scanned ? (uint32_t)d.anchor + 4 * q + r : 0;
    out[2] = scanned ? (uint32_t)d.anchor + 4 * q3 + r3 : 0;
    out[3] = d.bytes;
    out[4] = (uint32_t)(d.per >> 32);
    return SS_OK;
}
#endif

void ss_batch_plan_free(ss_batch_plan *p)
{
    if (!p) return;
    (void)hipFree(p->mem);          // (waits for the device: a run the caller forgot about cannot read freed memory)
    (void)hipFree(p->mem_alt);
    if (p->h_tally) (void)hipHostFree(p->h_tally);
    delete p;
}

int ss_search_pairs(const void *d_haystacks, const uint64_t *d_hay_begin, const uint64_t *d_hay_end,
                    const void *d_needles, const uint64_t *d_needle_begin, const uint64_t *d_needle_end,
                    const uint64_t *d_position, size_t count, void *hip_stream, int *d_found)
{
    if (count == 0) return SS_OK;
    if (!d_found) return fail(SS_ERR_ARGUMENT, "NULL argument");
    ss::BatchArgs a;
    if (int rc = fill_batch_args(&a, d_haystacks, d_hay_begin, d_hay_end, d_needles, d_needle_begin, d_needle_end, d_position)) return rc;
    a.found = d_found;
    const uint64_t blocks = ((uint64_t)count + ss::kBlock - 1) / ss::kBlock;
    if (blocks > 0x7fffffffull) return fail(SS_ERR_ARGUMENT, "too many problems");
    ss::scan_pairs_kernel<<<dim3((unsigned)blocks), dim3(ss::kBlock), 0, static_cast<hipStream_t>(hip_stream)>>>(a, count);
    HIP_TRY(hipGetLastError());
    return SS_OK;
}

}  // extern "C"
