// service_kernels.hpp - the resident search service's kernel (ss_service_*).  Included by ss_service.hip only.
#pragma once
#include "scan_kernels.hpp"

namespace ss {

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
// counter and posts the request again to a new residency.  Requests renew the lease, so the keeper also ends a residency that
// has lasted `residency_ticks` whatever the traffic (same protocol; the request that meets the leaving kernel starts the next
// one): nothing that waits for the whole device - hipDeviceSynchronize, hipFree - waits longer than that.  Every spin in here
// is bounded.
struct ServiceRequest {
    Problem pr;
    uint32_t one_byte;
    uint32_t stop;         // != 0: no search - the service ends
    uint32_t settled;      // != 0: every byte this request reads was last written before an earlier request's acquire (or the
                           // kernel's start) - a bound haystack (ss_service_bind), a needle uploaded earlier: no acquire
    uint32_t active;       // workgroups 0 .. active-1 scan (tiles b, b + active, ...) and count out; the others only watch
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
               int *d_found, uint32_t first_seq, unsigned long long idle_ticks, unsigned long long residency_ticks)
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
    const unsigned long long t_resident = __builtin_amdgcn_s_memrealtime();
    for (uint32_t next = first_seq;; ++next) {
        // the residency's cap (keeper only; the others follow the stop word): leave BETWEEN requests, like a lease that ran out
        if (keeper && next != first_seq && __builtin_amdgcn_s_memrealtime() - t_resident > residency_ticks) {
            if (lane == 0) {
                __hip_atomic_store(h_status, kSvcLeaving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(d_stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            break;
        }
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
        if (rq.one_byte)
            scan_tiles<0, 0, true, U, SS_SERVICE_NT, false, false, false>(rq.pr, cold, s_needle, blockIdx.x, rq.active, ntiles, d_found, &s_wg_found);
        else            // the second byte's window is run-time data (Problem::q)
            scan_tiles<kQDynamic, 0, false, U, SS_SERVICE_NT, false, false, false>(rq.pr, cold, s_needle, blockIdx.x, rq.active, ntiles, d_found, &s_wg_found);
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

}  // namespace ss
