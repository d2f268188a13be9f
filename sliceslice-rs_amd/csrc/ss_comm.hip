// ss_comm.hip - the range-sharded searches: rank r of G scans bytes [r*S, (r+1)*S + n-1) of the logical haystack,
// the found flags are combined by ONE ncclAllReduce(MAX) per search (OR over {0,1}; RCCL has no OR), leftmost offsets by ONE
// ncclAllReduce(MIN).  Two forms: one process per GPU (ss_comm_init_rank / ss_search_sharded / ss_find_sharded) and all GPUs
// of a node from one process (ss_comm_init_all / ss_search_sharded_all / ss_find_sharded_all: the form a drop-in
// `search_in(&self, &[u8]) -> bool`, /root/reference/src/x86.rs:523, needs - no launcher, no rendezvous).  librccl is
// dlopen()ed on first use.  The scans are the same kernels (enqueue_scan, ss_scan.hip); there is no CPU search path here.
#include "ss_internal.hpp"

#include <dlfcn.h>

using namespace ssh;

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
    int (*GetVersion)(int *) = nullptr;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        // SLICESLICE_RCCL_LIB names the library to use (a site's own build; the shared-memory stand-in of tests/native/fake_rccl.c,
        // which lets several ranks share one GPU): RTLD_LOCAL, so that its nccl* symbols never interpose on a librccl that is
        // already in the process (torch's)
        if (const char *path = getenv("SLICESLICE_RCCL_LIB")) {
            if (path[0]) r.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        } else {
            const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char *nm : names) {
                r.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
                if (r.h) break;
            }
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
        r.GetVersion = (decltype(r.GetVersion))dlsym(r.h, "ncclGetVersion");
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
struct SetWorker;
struct ss_comm_set {
    int ndev = 0;
    int combine = 0;            // SS_COMBINE_RCCL / SS_COMBINE_HOST
    int issue = 0;              // SS_ISSUE_THREADS / SS_ISSUE_SERIAL
    int epoch = 0;
    std::vector<int> devs;
    std::vector<void *> comms;
    std::vector<hipStream_t> streams;
    // per device: the flag PAIR {found, a device failed its local part} (both epoch-valued, never cleared), the all-reduce(MAX)
    // result, the pinned mirror the finding wave writes, the pinned source of the copy that raises the pair's second word
    std::vector<int *> d_flag, d_recv, h_flag, h_err;
    std::vector<uint64_t *> d_best, d_best_recv;
    bool no_rccl = false;                               // librccl could not be loaded: no communicators, host combine only
    int *h_recv = nullptr;                              // pinned int[2]: device 0's all-reduce result
    long long *h_words = nullptr;                       // pinned: ndev answer words of signal_flag_kernel (spinning read-back)
    uint64_t *h_best = nullptr;                         // pinned: ndev offsets (host combine) / [0] = all-reduce result
    std::atomic<bool> busy{false};                      // one search at a time per set: a second concurrent call is refused
    // Cross-device early exit of ss_search_sharded_all: the host, which waits for the answer words anyway, watches the pinned
    // mirrors the finding waves write and stores the epoch into every OTHER device's flag through that device's PCIe BAR;
    // their workgroups see it at their next poll and leave.  Possible when every device's memory is CPU-visible.
    bool relay_ok = false;
    std::vector<volatile uint32_t *> hdp_flush;         // per device: HDP flush register (pushes the store out of the host data path)
    // One issue thread per device (SS_ISSUE_THREADS): see SetWorker below.  Created with the set, joined by ss_comm_set_free.
    std::vector<SetWorker *> workers;
    uint64_t timed = 0;                                 // uid of the searcher of the latest search (ss_comm_set_last_kernel_ms); 0: none
    float issue_us[4] = {0, 0, 0, 0};                   // the latest search's host time: scans, collective, answer words, all of it
};

// ---- per-device issue threads -----------------------------------------------------------------------------------------
// From ONE thread a search over G devices is 3 G runtime calls in a row - G scan launches, G all-reduces (one group), G
// answer-word kernels, a hipSetDevice in front of each - and device G-1 starts its scan G-1 launches after device 0: at eight
// devices and a 1.2 ms shard scan that skew is a few per cent of the step (profiles/r05/native_set8.json has the host times).
// So a set keeps one thread per device, parked on the device for good: ss_search_sharded_all hands every thread its shard
// (one store per thread), each enqueues scan -> all-reduce -> answer word on its device's stream - the G chains are issued side by
// side, and RCCL's "one thread per communicator" form needs no group - and reports back.  A thread spins for kWorkerSpinUs after
// its latest job (back-to-back searches find it awake), then sleeps on a condition variable.
struct SetJob {
    int kind = 0;                       // 1 = one shard of a search, 2 = read the thread's kernel time, 3 = one shard of a find
    uint64_t begin = 0;                 // find: the shard's global offset
    uint64_t uid = 0;                   // kind 2: the searcher whose time is asked for (its uid: it may have been freed since)
    const ss_searcher *s = nullptr;
    const void *shard = nullptr;
    size_t len = 0;
    int epoch = 0;
    bool rccl = false, signal = false;
};
struct SetWorker {
    ss_comm_set *set = nullptr;
    int g = 0;
    std::thread th;
    std::atomic<uint32_t> posted{0}, done{0};
    std::atomic<bool> asleep{false}, quit{false}, caller_asleep{false};
    std::mutex mu;
    std::condition_variable cv, done_cv;
    SetJob job;
    int rc = SS_OK;
    char msg[256] = "";
    float kernel_ms = 0;
    float issue_us[3] = {0, 0, 0};
};
constexpr long long kWorkerSpinUs = 500;
constexpr long long kCallerSpinUs = 200;     // how long the caller spins for a worker's report before it sleeps (wait_job)

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

void stop_workers(ss_comm_set *set)
{
    for (SetWorker *w : set->workers) {
        w->quit.store(true, std::memory_order_seq_cst);
        {
            std::lock_guard<std::mutex> lock(w->mu);
            w->cv.notify_one();
        }
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    set->workers.clear();
}

void free_comm_set(ss_comm_set *set)
{
    stop_workers(set);
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
        if (g < (int)set->h_err.size()) (void)hipHostFree(set->h_err[g]);
        if (g < (int)set->d_best.size()) (void)hipFree(set->d_best[g]);
        if (g < (int)set->d_best_recv.size()) (void)hipFree(set->d_best_recv[g]);
    }
    (void)hipHostFree(set->h_recv);
    (void)hipHostFree(set->h_words);
    (void)hipHostFree(set->h_best);
    delete set;
}

double us_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

// One device's chain of a search: scan -> all-reduce -> answer word (or the stream wait), enqueued on the device's stream by
// whoever runs this - the device's issue thread, or the caller's thread for every device in turn (SS_ISSUE_SERIAL; then the
// all-reduces are issued separately, as one group).  The current device is the chain's.  A chain whose scan cannot be enqueued
// still goes on: it raises the pair's second word and ENTERS THE COLLECTIVE, so that no other device's all-reduce waits for it.
int issue_chain(ss_comm_set *set, int g, const SetJob &job, bool collective_here, float issue_us[3])
{
    const ss_searcher *s = job.s;
    hipStream_t st = set->streams[g];
    int rc = SS_OK;
    auto t0 = std::chrono::steady_clock::now();
    if (job.len >= s->n) {                                   // a shard shorter than the needle holds no candidate: the flag stays old
        PerDevice *pd = nullptr;
        rc = get_per_device(s, &pd);
        if (rc == SS_OK) rc = enqueue_scan(s, pd, job.shard, job.len, st, set->d_flag[g], false, 0, set->h_flag[g], job.epoch);
        if (rc != SS_OK) {
            *set->h_err[g] = job.epoch;
            (void)hipMemcpyAsync(set->d_flag[g] + 1, set->h_err[g], sizeof(int), hipMemcpyHostToDevice, st);
        }
    }
    issue_us[0] = (float)us_since(t0);
    t0 = std::chrono::steady_clock::now();
    if (job.rccl && collective_here) {
        Rccl *r = rccl();
        const int nrc = r->AllReduce(set->d_flag[g], set->d_recv[g], 2, kNcclInt32, kNcclMax, set->comms[g], st);
        if (nrc != 0 && rc == SS_OK) rc = rccl_fail(r, nrc, "ncclAllReduce");
    }
    issue_us[1] = (float)us_since(t0);
    return rc;
}

// ... and what follows the all-reduce on the device's stream: the answer word for a host that spins, or the read-back + stream wait.
int issue_tail(ss_comm_set *set, int g, const SetJob &job, bool wait_here, float issue_us[3])
{
    hipStream_t st = set->streams[g];
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipSuccess;
    if (job.signal) {
        e = launch_signal_flag(st, job.rccl ? set->d_recv[g] : set->d_flag[g], job.epoch, set->h_words + g, 1);
    } else {
        if (job.rccl && g == 0) e = hipMemcpyAsync(set->h_recv, set->d_recv[0], 2 * sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && wait_here) e = hipStreamSynchronize(st);
    }
    issue_us[2] = (float)us_since(t0);
    if (e != hipSuccess) return fail(SS_ERR_HIP, "device %d: %s", set->devs[g], hipGetErrorString(e));
    return SS_OK;
}

void run_job(SetWorker *w)
{
    ss_comm_set *set = w->set;
    w->rc = SS_OK;
    w->msg[0] = 0;
    if (w->job.kind == 2) {
        w->rc = thread_last_kernel_ms(w->job.uid, set->devs[w->g], &w->kernel_ms);
    } else if (w->job.kind == 3) {
        // one device's chain of a find: minimum reset -> scan (atomicMin of begin + offset) -> all-reduce(MIN) -> read-back -> wait.
        // A chain whose scan cannot be enqueued still enters the collective (its minimum stays all ones); the caller sees its error.
        const int g = w->g;
        hipStream_t st = set->streams[g];
        hipError_t e = hipMemsetAsync(set->d_best[g], 0xFF, sizeof(uint64_t), st);
        if (e != hipSuccess) w->rc = fail(SS_ERR_HIP, "device %d: %s", set->devs[g], hipGetErrorString(e));
        else w->rc = ss_find_device_async(w->job.s, w->job.shard, w->job.len, w->job.begin, st, set->d_best[g]);
        if (w->rc != SS_OK) snprintf(w->msg, sizeof w->msg, "%s", last_error());
        if (w->job.rccl) {
            Rccl *r = rccl();
            const int nrc = r->AllReduce(set->d_best[g], set->d_best_recv[g], 1, kNcclUint64, kNcclMin, set->comms[g], st);
            if (nrc != 0 && w->rc == SS_OK) {
                w->rc = rccl_fail(r, nrc, "ncclAllReduce");
                snprintf(w->msg, sizeof w->msg, "%s", last_error());
            }
        }
        if (!w->job.rccl || g == 0)
            e = hipMemcpyAsync(set->h_best + g, w->job.rccl ? set->d_best_recv[g] : set->d_best[g], sizeof(uint64_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess && w->rc == SS_OK) {
            w->rc = fail(SS_ERR_HIP, "device %d: %s", set->devs[g], hipGetErrorString(e));
            snprintf(w->msg, sizeof w->msg, "%s", last_error());
        }
    } else {
        w->rc = issue_chain(set, w->g, w->job, true, w->issue_us);
        if (w->rc != SS_OK) snprintf(w->msg, sizeof w->msg, "%s", last_error());
        const int trc = issue_tail(set, w->g, w->job, true, w->issue_us);
        if (trc != SS_OK && w->rc == SS_OK) {
            w->rc = trc;
            snprintf(w->msg, sizeof w->msg, "%s", last_error());
        }
    }
    if (w->rc != SS_OK && !w->msg[0]) snprintf(w->msg, sizeof w->msg, "%s", last_error());
}

void worker_main(SetWorker *w)
{
    (void)hipSetDevice(w->set->devs[w->g]);
    uint32_t seen = 0;
    auto idle_since = std::chrono::steady_clock::now();
    for (;;) {
        uint32_t p = seen;
        for (unsigned spins = 0;; ++spins) {
            p = w->posted.load(std::memory_order_acquire);
            if (p != seen || w->quit.load(std::memory_order_acquire)) break;
            cpu_relax();
            if ((spins & 1023) == 1023 && us_since(idle_since) > (double)kWorkerSpinUs) {
                std::unique_lock<std::mutex> lock(w->mu);
                w->asleep.store(true, std::memory_order_seq_cst);
                w->cv.wait(lock, [&]() { return w->posted.load(std::memory_order_seq_cst) != seen || w->quit.load(std::memory_order_seq_cst); });
                w->asleep.store(false, std::memory_order_seq_cst);
            }
        }
        if (p == seen) return;                                // quit
        run_job(w);
        seen = p;
        w->done.store(p, std::memory_order_seq_cst);
        if (w->caller_asleep.load(std::memory_order_seq_cst)) {      // (the caller gave up spinning: see wait_job)
            std::lock_guard<std::mutex> lock(w->mu);
            w->done_cv.notify_all();
        }
        idle_since = std::chrono::steady_clock::now();
    }
}

uint32_t post_job(SetWorker *w)
{
    const uint32_t p = w->posted.load(std::memory_order_relaxed) + 1;
    w->posted.store(p, std::memory_order_seq_cst);
    if (w->asleep.load(std::memory_order_seq_cst)) {
        std::lock_guard<std::mutex> lock(w->mu);
        w->cv.notify_one();
    }
    return p;
}

// The caller's side of a job: a job that only enqueues is done in microseconds, so spin first; a worker that waits for its stream
// inside the job (long shards, every find) - or sits in a collective another rank has not entered yet - is waited for on the
// worker's condition variable instead of a burning core.
void wait_job(SetWorker *w, uint32_t p)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; w->done.load(std::memory_order_acquire) != p; ++spins) {
        cpu_relax();
        if ((spins & 255) == 255 && us_since(t0) > (double)kCallerSpinUs) {
            std::unique_lock<std::mutex> lock(w->mu);
            w->caller_asleep.store(true, std::memory_order_seq_cst);
            w->done_cv.wait(lock, [&]() { return w->done.load(std::memory_order_seq_cst) == p; });
            w->caller_asleep.store(false, std::memory_order_seq_cst);
            return;
        }
    }
}

// std::thread's constructor throws std::system_error when the system is out of threads: not through an extern "C" boundary
bool start_worker(SetWorker *w)
{
    try {
        w->th = std::thread(worker_main, w);
    } catch (...) {
        return false;
    }
    return true;
}

bool threads_wanted()
{
    static const bool on = []() { const char *v = getenv("SLICESLICE_SET_THREADS"); return !(v && v[0] == '0'); }();
    return on;
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

#ifdef SS_TEST_HOOKS
int ss_debug_set_comm_epoch(ss_comm *c, ss_comm_set *set, int value)
{
    if (c) c->epoch = value;
    if (set) set->epoch = value;
    return SS_OK;
}
// how many searches of this process collected their answer words only AFTER the spin had run out of its budget (the drain-and-
// re-read path of ss_search_sharded_all / the stream wait of the pair form): what a test with a slow collective must have taken
static std::atomic<unsigned long long> g_late_answers{0};
uint64_t ss_debug_late_answers(void) { return (uint64_t)g_late_answers.load(std::memory_order_relaxed); }
#define SS_COUNT_LATE_ANSWER() g_late_answers.fetch_add(1, std::memory_order_relaxed)
#else
#define SS_COUNT_LATE_ANSWER() ((void)0)
#endif

// Which collective library the communicators of this process are made of: the file the resolved ncclAllReduce lives in (dladdr) and
// what its ncclGetVersion says.  A process that has torch in it holds torch's bundled librccl as well as the system's; which one
// "librccl.so.1" resolved to is otherwise invisible from a benchmark line.
int ss_comm_rccl_info(char *path, size_t path_cap, int *version)
{
    Rccl *r = rccl();
    if (!r) return fail(SS_ERR_RCCL, "librccl could not be loaded: %s", dlerror());
    if (path && path_cap) {
        Dl_info info;
        path[0] = 0;
        if (dladdr(reinterpret_cast<void *>(r->AllReduce), &info) && info.dli_fname) snprintf(path, path_cap, "%s", info.dli_fname);
    }
    if (version) {
        *version = 0;
        if (r->GetVersion) (void)r->GetVersion(version);
    }
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
    char local_msg[512] = "";
    if (shard_len >= s->n) {                                // a shard shorter than the needle holds no candidate
        PerDevice *pd = nullptr;
        local_rc = get_per_device(s, &pd);
        if (local_rc == SS_OK) local_rc = enqueue_scan(s, pd, d_shard, shard_len, st, c->d_flag, false, 0, nullptr, epoch);
        if (local_rc != SS_OK) {
            snprintf(local_msg, sizeof local_msg, "%s", last_error());
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
    const bool spin_ok = spin_wait_enabled();
    const double estimate = scan_estimate_us(shard_len) + 100.0;      // + the collective
    if (spin_ok && estimate <= kSpinMaxEstimateUs && scan_estimate_us(shard_len) >= kSpinMinEstimateUs) {
        // the answer word behind the all-reduce, and a bounded spin on it (see spin_for_word); ranks that arrive late in
        // the collective make the others' spins run out, which costs those nothing but the stream wait they had before
        __atomic_store_n(c->h_word, 0ll, __ATOMIC_RELAXED);
        HIP_TRY(launch_signal_flag(st, c->d_recv, epoch, c->h_word, 1));
        int failed = 0;
        if (spin_for_shard_word(c->h_word, epoch, estimate, found, &failed)) {
            if ((epoch & 255) == 0) HIP_TRY(hipStreamSynchronize(st));
            return done(failed);
        }
        SS_COUNT_LATE_ANSWER();
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
    char local_msg[512] = "";
    hipError_t e0 = hipMemsetAsync(c->d_best, 0xFF, 2 * sizeof(uint64_t), st);
    if (e0 != hipSuccess) local_rc = fail(SS_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e0));
    if (local_rc == SS_OK) local_rc = ss_find_device_async(s, d_shard, shard_len, shard_begin, hip_stream, c->d_best);
    if (local_rc != SS_OK) {                                // still enter the collective: see ss_search_sharded
        snprintf(local_msg, sizeof local_msg, "%s", last_error());
        (void)hipMemsetAsync(c->d_best + 1, 0, sizeof(uint64_t), st);
    }
    auto done = [&](uint64_t status) {
        if (local_rc != SS_OK) return fail(local_rc, "%s", local_msg);
        if (status != ~0ull) return fail(SS_ERR_PEER, "another rank failed the local part of this sharded find; no answer");
        return (int)SS_OK;
    };
    if (int rc = r->AllReduce(c->d_best, c->d_best, 2, kNcclUint64, kNcclMin, c->comm, st)) return rccl_fail(r, rc, "ncclAllReduce");
    const bool spin_ok = spin_wait_enabled();
    const double estimate = scan_estimate_us(shard_len) + 100.0;
    if (spin_ok && estimate <= kSpinMaxEstimateUs && scan_estimate_us(shard_len) >= kSpinMinEstimateUs) {
        // as ss_search_sharded: the pinned mirror starts as "pending", a one-lane kernel behind the all-reduce stores the
        // pair (and re-arms nothing: d_best is this communicator's scratch, set to all ones at the top of every call)
        constexpr uint64_t kPending = ~0ull - 1;
        __atomic_store_n(c->h_best, kPending, __ATOMIC_RELAXED);
        HIP_TRY(launch_publish_best(st, c->d_best, c->h_best, 1));
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
    // Without a loadable librccl the set still works: no communicators are created and only the host combine is offered (all
    // ranks live in this process; the collective is an option here, not a need).  Whether a device may be listed twice is RCCL's
    // call (ncclCommInitAll refuses; the shared-memory stand-in of the tests, tests/native/fake_rccl.c, allows it).
    Rccl *r = rccl();
    const bool no_rccl = r == nullptr;
    if (ndev < 1 || ndev > kMaxDevices) return fail(SS_ERR_ARGUMENT, "%d devices requested (1 .. %d)", ndev, kMaxDevices);
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
        set->devs.push_back(d);
    }
    set->comms.assign(ndev, nullptr);
    set->streams.assign(ndev, nullptr);
    set->d_flag.assign(ndev, nullptr);
    set->d_recv.assign(ndev, nullptr);
    set->h_flag.assign(ndev, nullptr);
    set->h_err.assign(ndev, nullptr);
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
        if (e == hipSuccess) e = hipMalloc((void **)&set->d_recv[g], 2 * sizeof(int));
        if (e == hipSuccess) e = hipMemset(set->d_recv[g], 0, 2 * sizeof(int));
        if (e == hipSuccess) e = hipHostMalloc((void **)&set->h_flag[g], sizeof(int), hipHostMallocPortable | hipHostMallocMapped);
        if (e == hipSuccess) *set->h_flag[g] = 0;
        if (e == hipSuccess) e = hipHostMalloc((void **)&set->h_err[g], sizeof(int), hipHostMallocPortable);
        if (e == hipSuccess) e = hipMalloc((void **)&set->d_best[g], sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void **)&set->d_best_recv[g], sizeof(uint64_t));
    }
    if (e == hipSuccess) e = hipHostMalloc((void **)&set->h_recv, 2 * sizeof(int), hipHostMallocPortable | hipHostMallocMapped);
    if (e == hipSuccess) set->h_recv[0] = set->h_recv[1] = 0;
    if (e == hipSuccess) e = hipHostMalloc((void **)&set->h_words, (size_t)ndev * sizeof(long long), hipHostMallocPortable | hipHostMallocMapped);
    if (e == hipSuccess) memset(set->h_words, 0, (size_t)ndev * sizeof(long long));
    if (e == hipSuccess) e = hipHostMalloc((void **)&set->h_best, (size_t)ndev * sizeof(uint64_t), hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) {
        free_comm_set(set);
        return fail(SS_ERR_HIP, "communicator set scratch: %s", hipGetErrorString(e));
    }
    {
        set->relay_ok = bar_writes_allowed();
        set->hdp_flush.assign((size_t)ndev, nullptr);
        for (int g = 0; g < ndev; ++g) {
            DeviceInfo di;
            if (device_info(set->devs[g], &di) != SS_OK || !di.large_bar) set->relay_ok = false;
            else set->hdp_flush[g] = di.hdp_flush;
            (void)hipSetDevice(set->devs[g]);
            (void)hipDeviceSynchronize();               // the memsets above are asynchronous to the host: done before anyone stores there
        }
    }
    // one issue thread per device (a set of one device has nothing to issue side by side)
    set->issue = ndev >= 2 && threads_wanted() ? SS_ISSUE_THREADS : SS_ISSUE_SERIAL;
    if (set->issue == SS_ISSUE_THREADS) {
        for (int g = 0; g < ndev; ++g) {
            SetWorker *w = new (std::nothrow) SetWorker;
            if (!w) {
                free_comm_set(set);
                return fail(SS_ERR_NOMEM, "out of memory");
            }
            w->set = set;
            w->g = g;
            set->workers.push_back(w);
            if (!start_worker(w)) {
                free_comm_set(set);
                return fail(SS_ERR_NOMEM, "an issue thread could not be started");
            }
        }
    }
    *out = set;
    return SS_OK;
}

void ss_comm_set_free(ss_comm_set *set)
{
    if (set) free_comm_set(set);
}

int ss_comm_set_combine(ss_comm_set *set, int combine)
{
    if (!set || (combine != SS_COMBINE_RCCL && combine != SS_COMBINE_HOST)) return fail(SS_ERR_ARGUMENT, "bad combine mode");
    if (set->no_rccl && combine == SS_COMBINE_RCCL) return fail(SS_ERR_RCCL, "this set was created without communicators (librccl could not be loaded)");
    set->combine = combine;
    return SS_OK;
}

int ss_comm_set_issue(ss_comm_set *set, int issue)
{
    if (!set || (issue != SS_ISSUE_THREADS && issue != SS_ISSUE_SERIAL)) return fail(SS_ERR_ARGUMENT, "bad issue mode");
    BusyGuard busy(&set->busy);
    if (!busy.mine) return fail(SS_ERR_ARGUMENT, "this communicator set is in use by a search");
    if (issue == SS_ISSUE_THREADS && set->workers.empty()) {
        if (set->ndev < 2) return SS_OK;                      // one device: the caller's thread is the issue thread
        for (int g = 0; g < set->ndev; ++g) {
            SetWorker *w = new (std::nothrow) SetWorker;
            if (!w) return fail(SS_ERR_NOMEM, "out of memory");
            w->set = set;
            w->g = g;
            set->workers.push_back(w);
            if (!start_worker(w)) {
                stop_workers(set);                                // all or none: the set stays on the one-thread form
                set->issue = SS_ISSUE_SERIAL;
                return fail(SS_ERR_NOMEM, "an issue thread could not be started");
            }
        }
    }
    set->issue = issue;
    return SS_OK;
}

int ss_comm_set_count(const ss_comm_set *set, int *nranks)
{
    if (!set || !nranks) return fail(SS_ERR_ARGUMENT, "NULL argument");
    if (set->no_rccl) return fail(SS_ERR_RCCL, "this set was created without communicators (librccl could not be loaded)");
    Rccl *r = rccl();
    if (!r) return fail(SS_ERR_RCCL, "librccl not loaded");
    int agreed = -1;
    for (int g = 0; g < set->ndev; ++g) {                    // what RCCL itself says, of EVERY communicator of the set
        int n = 0;
        if (int rc = r->CommCount(set->comms[g], &n)) return rccl_fail(r, rc, "ncclCommCount");
        if (agreed >= 0 && n != agreed) return fail(SS_ERR_RCCL, "communicator %d of the set reports %d ranks, communicator 0 %d", g, n, agreed);
        agreed = n;
    }
    *nranks = agreed;
    return SS_OK;
}

int ss_comm_set_last_kernel_ms(ss_comm_set *set, float *ms, int count)
{
    if (!set || !ms || count < set->ndev) return fail(SS_ERR_ARGUMENT, "bad argument");
    BusyGuard busy(&set->busy);
    if (!busy.mine) return fail(SS_ERR_ARGUMENT, "this communicator set is in use by a search");
    if (!set->timed) return fail(SS_ERR_ARGUMENT, "no search has been made through this set yet");
    if (set->issue == SS_ISSUE_THREADS && !set->workers.empty()) {
        std::vector<uint32_t> posted(set->ndev);
        for (int g = 0; g < set->ndev; ++g) {
            set->workers[g]->job.kind = 2;
            set->workers[g]->job.s = nullptr;
            set->workers[g]->job.uid = set->timed;
            posted[g] = post_job(set->workers[g]);
        }
        int rc = SS_OK;
        for (int g = 0; g < set->ndev; ++g) {
            wait_job(set->workers[g], posted[g]);
            if (set->workers[g]->rc != SS_OK && rc == SS_OK) rc = fail(set->workers[g]->rc, "%s", set->workers[g]->msg);
            ms[g] = set->workers[g]->kernel_ms;
        }
        return rc;
    }
    for (int g = 0; g < set->ndev; ++g)
        if (int rc = thread_last_kernel_ms(set->timed, set->devs[g], ms + g)) return rc;
    return SS_OK;
}

int ss_comm_set_last_issue_us(const ss_comm_set *set, float us[4])
{
    if (!set || !us) return fail(SS_ERR_ARGUMENT, "NULL argument");
    for (int k = 0; k < 4; ++k) us[k] = set->issue_us[k];
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
    set->timed = s->uid;
    // Scans short enough to be waited for by spinning (spin_for_word) end with a one-lane kernel per device that stores the
    // device's answer word - behind the scan and the all-reduce, so a word that has arrived says its stream is done - and the
    // host collects the G words; longer ones are waited for on their streams.
    const bool spin_ok = spin_wait_enabled();
    size_t longest = 0;
    for (int g = 0; g < G; ++g) longest = shard_lens[g] > longest ? shard_lens[g] : longest;
    const double estimate = scan_estimate_us(longest) + 100.0;
    const bool signal = spin_ok && estimate <= kSpinMaxEstimateUs && scan_estimate_us(longest) >= kSpinMinEstimateUs;
    const bool rccl_on = set->combine == SS_COMBINE_RCCL;
    const bool threads = set->issue == SS_ISSUE_THREADS && (int)set->workers.size() == G;
    if (signal)
        for (int g = 0; g < G; ++g) __atomic_store_n(set->h_words + g, 0ll, __ATOMIC_RELAXED);
    int rc = SS_OK;
    char msg[256] = "";
    bool tails_waited = false;                              // every stream has been waited for already
    const auto t_issue = std::chrono::steady_clock::now();
    float us[3] = {0, 0, 0};
    if (threads) {
        // 1t. every device's chain - scan, all-reduce(MAX) of the flag pair, answer word - from that device's own thread
        std::vector<uint32_t> posted(G);
        for (int g = 0; g < G; ++g) {
            SetJob &j = set->workers[g]->job;
            j.kind = 1;
            j.s = s;
            j.shard = d_shards[g];
            j.len = shard_lens[g];
            j.epoch = epoch;
            j.rccl = rccl_on;
            j.signal = signal;
            posted[g] = post_job(set->workers[g]);
        }
        for (int g = 0; g < G; ++g) {
            SetWorker *w = set->workers[g];
            wait_job(w, posted[g]);
            if (w->rc != SS_OK && rc == SS_OK) {
                rc = w->rc;
                snprintf(msg, sizeof msg, "%s", w->msg);
            }
            for (int k = 0; k < 3; ++k) us[k] = std::max(us[k], w->issue_us[k]);
        }
        tails_waited = !signal;
    } else {
        // 1s. one scan per device, each on its device's stream, from this thread; the finding wave also writes the epoch to
        //     that device's pinned-host mirror
        SetJob j;
        j.kind = 1;
        j.s = s;
        j.epoch = epoch;
        j.rccl = rccl_on;
        j.signal = signal;
        float one[3];
        for (int g = 0; g < G && rc == SS_OK; ++g) {
            if (hipSetDevice(set->devs[g]) != hipSuccess) { rc = fail(SS_ERR_HIP, "hipSetDevice(%d) failed", set->devs[g]); break; }
            j.shard = d_shards[g];
            j.len = shard_lens[g];
            rc = issue_chain(set, g, j, false, one);
            us[0] += one[0];
        }
        if (rc != SS_OK) snprintf(msg, sizeof msg, "%s", last_error());
        // 2s. combine: G all-reduce(MAX) calls as ONE group, or no collective at all - the host ORs the G pinned mirrors
        //     (possible only because all ranks live in this process).  Nothing has entered a collective yet, so a failed enqueue
        //     simply ends the call.
        if (rc == SS_OK && rccl_on) {
            const auto t0 = std::chrono::steady_clock::now();
            int nrc = r->GroupStart();
            for (int g = 0; g < G && nrc == 0; ++g)
                nrc = r->AllReduce(set->d_flag[g], set->d_recv[g], 2, kNcclInt32, kNcclMax, set->comms[g], set->streams[g]);
            const int erc = r->GroupEnd();
            if (nrc == 0) nrc = erc;
            if (nrc != 0) {
                rc = rccl_fail(r, nrc, "grouped ncclAllReduce");
                snprintf(msg, sizeof msg, "%s", last_error());
            }
            us[1] = (float)us_since(t0);
        }
        // 3s. the answer words / the read-back
        for (int g = 0; g < G && rc == SS_OK; ++g) {
            if (hipSetDevice(set->devs[g]) != hipSuccess) { rc = fail(SS_ERR_HIP, "hipSetDevice(%d) failed", set->devs[g]); break; }
            rc = issue_tail(set, g, j, false, one);
            us[2] += one[2];
            if (rc != SS_OK) snprintf(msg, sizeof msg, "%s", last_error());
        }
    }
    set->issue_us[0] = us[0];
    set->issue_us[1] = us[1];
    set->issue_us[2] = us[2];
    set->issue_us[3] = (float)us_since(t_issue);
    bool spun = false;
    int any_word = 0, any_failed = 0;
    if (rc == SS_OK && signal) {
        // Collect the G answer words (epoch << 2 | a device failed << 1 | found); meanwhile - cross-device early exit - watch the
        // pinned mirrors: the first device that reports a match has its epoch stored into every other device's flag through the
        // BAR, so that THEIR grids stop scanning too (a match in shard 0 of a 64 GiB haystack over eight devices otherwise costs
        // the full 1.2 ms scan of the seven others).  The flag only ever means "found somewhere": the OR of the answers is unchanged.
        const auto t0 = std::chrono::steady_clock::now();
        const auto budget = std::chrono::microseconds((long long)(2.0 * estimate) + 300);
        uint64_t got = 0;
        bool relayed = !set->relay_ok || G < 2 || !cross_exit_enabled();
        spun = true;
        for (unsigned spins = 0; got != (G >= 64 ? ~0ull : (1ull << G) - 1); ++spins) {
            for (int g = 0; g < G; ++g) {
                if ((got >> g) & 1) continue;
                const unsigned long long v = (unsigned long long)__atomic_load_n(set->h_words + g, __ATOMIC_ACQUIRE);
                if ((v >> 2) == (unsigned long long)(uint32_t)epoch) {
                    any_word |= (int)(v & 1);
                    any_failed |= (int)((v >> 1) & 1);
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
                    // push the stores out of each device's host data path and WAIT for them (read the register back, as
                    // bar_write does): a posted write still in flight when this call returns could land after the NEXT
                    // search on the set has moved that flag on to its own epoch, and put it back
                    for (int o = 0; o < G; ++o) {
                        if (o == g || ((got >> o) & 1) || shard_lens[o] < s->n) continue;
                        if (set->hdp_flush[o]) {
                            __atomic_store_n(set->hdp_flush[o], 1u, __ATOMIC_RELAXED);
                            (void)__atomic_load_n(set->hdp_flush[o], __ATOMIC_RELAXED);
                        } else {
                            (void)*reinterpret_cast<volatile int *>(set->d_flag[o]);
                        }
                    }
                    relayed = true;
                    break;
                }
            }
            cpu_relax();
            if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > budget) { spun = false; break; }
        }
        if (spun && (epoch & 255) != 0) {
            if (any_failed) return fail(SS_ERR_HIP, "a device of the set failed its part of this search; no answer");
            *found = any_word;
            return SS_OK;
        }
    }
    // every stream is drained before the call returns: the haystacks are only borrowed for the call
    if (!tails_waited || rc != SS_OK) {
        for (int g = 0; g < G; ++g) {
            hipError_t e = hipSetDevice(set->devs[g]);
            if (e == hipSuccess) e = hipStreamSynchronize(set->streams[g]);
            if (e != hipSuccess && rc == SS_OK) {
                rc = fail(SS_ERR_HIP, "stream wait on device %d: %s", set->devs[g], hipGetErrorString(e));
                snprintf(msg, sizeof msg, "%s", last_error());
            }
        }
    }
    if (rc != SS_OK) return fail(rc, "%s", msg);
    if (spun) {
        if (any_failed) return fail(SS_ERR_HIP, "a device of the set failed its part of this search; no answer");
        *found = any_word;
        return SS_OK;
    }
    int any = 0;
    if (signal) {
        // The spin ran out of its budget (a first ncclAllReduce that connects lazily, a device busy with other work) and the streams
        // have been drained since: the G answer words ARE written now - every chain ends with its answer-word kernel, and only the
        // non-signal chains enqueue the read-back into h_recv.  A word that is still missing is an error, never "not found".
        SS_COUNT_LATE_ANSWER();
        for (int g = 0; g < G; ++g) {
            const unsigned long long v = (unsigned long long)__atomic_load_n(set->h_words + g, __ATOMIC_ACQUIRE);
            if ((v >> 2) != (unsigned long long)(uint32_t)epoch)
                return fail(SS_ERR_HIP, "device %d of the set did not write its answer word", set->devs[g]);
            any |= (int)(v & 1);
            any_failed |= (int)((v >> 1) & 1);
        }
        if (any_failed) return fail(SS_ERR_HIP, "a device of the set failed its part of this search; no answer");
    } else if (rccl_on) {
        any = set->h_recv[0] == epoch;
        if (set->h_recv[1] == epoch) return fail(SS_ERR_HIP, "a device of the set failed its part of this search; no answer");
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
    if (set->issue == SS_ISSUE_THREADS && (int)set->workers.size() == G) {
        // every device's chain from that device's issue thread (see SetWorker); the threads wait for their streams themselves
        std::vector<uint32_t> posted(G);
        for (int g = 0; g < G; ++g) {
            SetJob &j = set->workers[g]->job;
            j.kind = 3;
            j.s = s;
            j.shard = d_shards[g];
            j.len = shard_lens[g];
            j.begin = shard_begins[g];
            j.rccl = set->combine == SS_COMBINE_RCCL;
            posted[g] = post_job(set->workers[g]);
        }
        char msg[256] = "";
        for (int g = 0; g < G; ++g) {
            wait_job(set->workers[g], posted[g]);
            if (set->workers[g]->rc != SS_OK && rc == SS_OK) {
                rc = set->workers[g]->rc;
                snprintf(msg, sizeof msg, "%s", set->workers[g]->msg);
            }
        }
        if (rc != SS_OK) return fail(rc, "%s", msg);
        uint64_t best = set->h_best[0];
        if (set->combine != SS_COMBINE_RCCL)
            for (int g = 1; g < G; ++g) best = set->h_best[g] < best ? set->h_best[g] : best;
        *position = best;
        return SS_OK;
    }
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

}  // extern "C"
