// ss_service.hip - host side of the resident search service (ss::service_kernel, service_kernels.hpp has the protocol): a kernel
// that stays on the device and takes searches from a mailbox in device memory instead of being launched per search - what is
// left of a per-call search_in (/root/reference/src/x86.rs:523) when the launch is taken out: one PCIe round trip.  One request
// at a time per service (a mutex); the kernel is (re)started on demand - at the first request, after every lease that ran out,
// after every residency that reached its cap.  There is no CPU search path in this file.
#include "ss_internal.hpp"

#include <algorithm>

#include "service_kernels.hpp"

using namespace ssh;

struct ss_service {
    int dev = 0;
    int workgroups = 0;
    unsigned long long idle_ticks = 0, residency_ticks = 0;     // s_memrealtime ticks (100 MHz): the lease, and a residency's cap
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
    // ss_service_stop may free the service only when nobody is inside it: calls count themselves in before they take the
    // mutex, and a call that gets the mutex after the stop request finds `stopped` set and leaves.
    std::atomic<int> users{0};
    bool stopped = false;                   // under mu
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
// A residency ends after this many leases (and at least kServiceMinResidencyMs) even when requests never stop: every request
// renews the lease, so continuous traffic would otherwise keep the kernel resident for good - and anything that waits for the
// whole device (hipDeviceSynchronize in ss_searcher_free, hipMalloc / hipFree anywhere in the process) with it.  The host starts
// the next residency with the request that found the kernel gone, at the price of one launch.
constexpr double kServiceResidencyLeases = 16.0, kServiceMinResidencyMs = 250.0;

struct ServiceUser {                        // one call's presence in the service (see ss_service::users)
    ss_service *sv;
    explicit ServiceUser(ss_service *s) : sv(s) { sv->users.fetch_add(1, std::memory_order_acq_rel); }
    ~ServiceUser() { sv->users.fetch_sub(1, std::memory_order_acq_rel); }
    ServiceUser(const ServiceUser &) = delete;
    ServiceUser &operator=(const ServiceUser &) = delete;
};

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
#ifdef SS_TEST_HOOKS
    static const bool dbg = getenv("SLICESLICE_SERVICE_DEBUG") != nullptr;
    const auto w0 = dbg ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
#endif
    for (int k = 0; k < 16; ++k)
        _mm_store_si128(reinterpret_cast<__m128i *>(const_cast<uint32_t *>(m)) + k, _mm_load_si128(reinterpret_cast<const __m128i *>(img) + k));
    _mm_sfence();
    for (int line = 0; line < 4; ++line) m[line * 16 + 15] = seq;
    _mm_sfence();
    if (sv->hdp_flush) __atomic_store_n(sv->hdp_flush, 1u, __ATOMIC_RELAXED);   // (no read-back: the kernel polls, nothing is ordered behind this)
#ifdef SS_TEST_HOOKS
    if (dbg) {
        static double total_us = 0;
        static unsigned long n = 0;
        total_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        if (++n == 0x4000) {            // the average of the last 16,384 requests, then start over
            fprintf(stderr, "[service] mailbox write: %.3f us average over %lu requests\n", total_us / n, n);
            total_us = 0;
            n = 0;
        }
    }
#endif
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
        const_cast<unsigned long long *>(sv->answer()), sv->d_stop(), sv->d_done(), sv->d_found(), first_seq, sv->idle_ticks, sv->residency_ticks);
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
            // The keeper has left.  It may have left right BEHIND this request (a residency that reached its cap ends between
            // two requests, not after an idle lease): the workgroup that completes the count may still be about to store the
            // answer.  Wait for the kernel to be gone - whatever it was going to store has been stored then - and look again
            // before posting the request a second time: a second residency answering the same request would leave the host's
            // copy of the counter's found half (taken from the FIRST answer) out of step with the device's.
            HIP_TRY(hipStreamSynchronize(sv->stream));
            const unsigned long long late = __atomic_load_n(sv->answer(), __ATOMIC_ACQUIRE);
            if ((uint32_t)((late >> 1) & 0x7FFFFFFFu) == seq) {
                *answer = late;
                return SS_OK;
            }
            // the lease ran out (or the cap was reached) before all of the kernel saw this request: a new residency starts with it
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
    if (!di.large_bar || !bar_writes_allowed())
        return fail(SS_ERR_NO_DEVICE, "device %d does not expose its memory to the CPU (no large BAR, or SLICESLICE_NO_BAR_WRITES=1): no search service", dev);
    ss_service *sv = new (std::nothrow) ss_service;
    if (!sv) return fail(SS_ERR_NOMEM, "out of memory");
    sv->dev = dev;
    sv->workgroups = workgroups;
    sv->hdp_reg = sv->hdp_flush = di.hdp_flush;
#ifdef SS_TEST_HOOKS
    if (const char *v = getenv("SLICESLICE_SERVICE_HDP_FLUSH")) { if (v[0] == '0') sv->hdp_flush = nullptr; }   // measurement: requests without the flush
#endif
    sv->idle_ticks = (unsigned long long)(lease_ms * 1e5);             // s_memrealtime: 100 MHz
    sv->residency_ticks = (unsigned long long)(std::max(kServiceResidencyLeases * lease_ms, kServiceMinResidencyMs) * 1e5);
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
#ifdef SS_TEST_HOOKS
    static const bool dbg = getenv("SLICESLICE_SERVICE_DEBUG") != nullptr;
    const auto c0 = dbg ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
#endif
    ServiceUser user(sv);
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
    rq.one_byte = ps.one_byte ? 1u : 0u;
    std::lock_guard<std::mutex> lock(sv->mu);
    if (sv->stopped) return fail(SS_ERR_ARGUMENT, "this search service has been stopped");
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
#ifdef SS_TEST_HOOKS
    const auto c1 = dbg ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
#endif
#ifdef SS_TEST_HOOKS
    const uint64_t launches_before = sv->launches;
    const uint32_t hi_before = sv->done_hi, low_before = sv->done_low;
#endif
    if (int rc = service_post(sv, rq, seq, &a)) return rc;
#ifdef SS_TEST_HOOKS
    if (dbg && sv->launches != launches_before)
        fprintf(stderr, "[service] request %u met %llu relaunch(es): answer %016llx, target %u (active %u), host counter before %u/%u\n", seq,
                (unsigned long long)(sv->launches - launches_before), a, rq.pr.done_target, rq.active, hi_before, low_before);
#endif
#ifdef SS_TEST_HOOKS
    if (dbg) {
        static double prep_us = 0, post_us = 0;
        static unsigned long n = 0;
        const auto c2 = std::chrono::steady_clock::now();
        prep_us += std::chrono::duration<double, std::micro>(c1 - c0).count();
        post_us += std::chrono::duration<double, std::micro>(c2 - c1).count();
        if (++n == 0x4000) {
            fprintf(stderr, "[service] per request: %.3f us before the post, %.3f us post + wait (%lu requests)\n", prep_us / n, post_us / n, n);
            prep_us = post_us = 0;
            n = 0;
        }
    }
#endif
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
    ServiceUser user(sv);
    std::lock_guard<std::mutex> lock(sv->mu);
    if (sv->stopped) return fail(SS_ERR_ARGUMENT, "this search service has been stopped");
    sv->bound_lo = len ? static_cast<const uint8_t *>(d_haystack) : nullptr;
    sv->bound_hi = sv->bound_lo ? sv->bound_lo + len : nullptr;
    sv->bound_settled = false;                          // the next request inside the range acquires it, the ones after that do not
    return SS_OK;
}

#ifdef SS_TEST_HOOKS
int ss_service_counters(ss_service *sv, uint64_t *requests, uint64_t *kernel_launches, uint64_t *settled)
{
    if (!sv) return fail(SS_ERR_ARGUMENT, "service is NULL");
    ServiceUser user(sv);
    std::lock_guard<std::mutex> lock(sv->mu);
    if (requests) *requests = sv->requests;
    if (kernel_launches) *kernel_launches = sv->launches;
    if (settled) *settled = sv->settled_requests;
    return SS_OK;
}
#endif

void ss_service_stop(ss_service *sv)
{
    if (!sv) return;
    DeviceGuard guard;
    (void)hipSetDevice(sv->dev);
    {
        std::lock_guard<std::mutex> lock(sv->mu);               // (a search still in its wait loop finishes first)
        sv->stopped = true;                                     // ... and whoever gets the mutex after us leaves at once
        ss::ServiceRequest bye;
        memset(&bye, 0, sizeof bye);
        bye.stop = 1;
        unsigned long long ignored = 0;
        (void)service_post(sv, bye, ++sv->seq, &ignored);      // (a kernel that does not answer leaves when its lease runs out)
    }
    // calls that had entered before the stop (blocked on the mutex, or on their way out) are gone before the memory is
    while (sv->users.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    service_free(sv);
}

}  // extern "C"
