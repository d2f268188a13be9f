// Can the host store straight into device memory (large BAR), and what does a host -> resident lane -> host round trip cost when the
// REQUEST word lives in device memory (the lane polls locally) instead of pinned host memory (the lane polls over PCIe)?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/vram_mailbox_probe tools/vram_mailbox_probe.hip && /tmp/vram_mailbox_probe
// Prints one JSON line.  The host-store test catches the fault of a part without CPU-visible VRAM.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <csetjmp>
#include <csignal>
#include <unistd.h>
#include <vector>
#include <x86intrin.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void echo_kernel(const unsigned long long *req, unsigned long long *resp, unsigned iters, unsigned long long max_ticks)
{
    for (unsigned i = 1; i <= iters; ++i) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < i)
            if (__builtin_amdgcn_s_memrealtime() - t0 > max_ticks) return;
        __hip_atomic_store(resp, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// many workgroups poll the same word (agent scope), the LAST to count out answers: the service's shape without a leader hop
__global__ void echo_grid_kernel(const unsigned long long *req, unsigned long long *resp, unsigned long long *done, unsigned iters,
                                 unsigned long long max_ticks, int scope_system)
{
    if (threadIdx.x != 0) return;
    for (unsigned i = 1; i <= iters; ++i) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const unsigned long long v = scope_system ? __hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                                      : __hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v >= i) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > max_ticks) return;
            __builtin_amdgcn_s_sleep(2);
        }
        const unsigned long long n = __hip_atomic_fetch_add(done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        if (n == (unsigned long long)i * gridDim.x) __hip_atomic_store(resp, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the same, every workgroup answers into a slot of its own in pinned memory (no counter hop)
__global__ void echo_slots_kernel(const unsigned long long *req, unsigned *resp_slots, unsigned iters, unsigned long long max_ticks)
{
    if (threadIdx.x != 0) return;
    for (unsigned i = 1; i <= iters; ++i) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            if (__hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= i) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > max_ticks) return;
            __builtin_amdgcn_s_sleep(2);
        }
        __hip_atomic_store(resp_slots + blockIdx.x, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

// (not in a forked child: the driver's mappings are not inherited across fork(), a child faults whatever the parent may do)
static sigjmp_buf g_jmp;
static void on_fault(int) { siglongjmp(g_jmp, 1); }
static bool host_can_store(void *p)
{
    struct sigaction sa, old_segv, old_bus;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_fault;
    sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGBUS, &sa, &old_bus);
    bool ok = false;
    if (sigsetjmp(g_jmp, 1) == 0) {
        volatile unsigned *q = (volatile unsigned *)p;
        q[0] = 0xABCD1234u;
        _mm_sfence();
        ok = q[0] == 0xABCD1234u;
    }
    sigaction(SIGSEGV, &old_segv, nullptr);
    sigaction(SIGBUS, &old_bus, nullptr);
    return ok;
}

int main()
{
    const unsigned iters = 2000;
    const unsigned long long patience = 200000000ull;      // 2 s of s_memrealtime
    unsigned long long *h_req = nullptr, *h_resp = nullptr, *d_req = nullptr, *d_fine = nullptr, *d_done = nullptr;
    unsigned *h_slots = nullptr;
    CHECK(hipHostMalloc((void **)&h_req, 64, hipHostMallocMapped));
    CHECK(hipHostMalloc((void **)&h_resp, 64, hipHostMallocMapped));
    CHECK(hipHostMalloc((void **)&h_slots, 1024, hipHostMallocMapped));
    CHECK(hipMalloc((void **)&d_req, 4096));
    CHECK(hipMalloc((void **)&d_done, 64));
    const bool fine_ok = hipExtMallocWithFlags((void **)&d_fine, 4096, hipDeviceMallocFinegrained) == hipSuccess;
    (void)hipGetLastError();
    const bool coarse_host = host_can_store(d_req);
    const bool fine_host = fine_ok && host_can_store(d_fine);
    printf("{\"host_store_to_hipMalloc\": %s, \"host_store_to_finegrained\": %s", coarse_host ? "true" : "false",
           fine_ok ? (fine_host ? "true" : "false") : "null");
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto run_single = [&](unsigned long long *req_dev_view, volatile unsigned long long *req_host_view, const char *name) {
        *req_host_view = 0;
        _mm_sfence();
        *(volatile unsigned long long *)h_resp = 0;
        CHECK(hipDeviceSynchronize());
        echo_kernel<<<1, 1, 0, st>>>(req_dev_view, h_resp, iters, patience);
        std::vector<double> us;
        for (unsigned i = 1; i <= iters; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            *req_host_view = i;
            _mm_sfence();
            while (__atomic_load_n(h_resp, __ATOMIC_ACQUIRE) < i) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3)) { printf(", \"%s\": \"timeout at %u\"", name, i); CHECK(hipStreamSynchronize(st)); return; }
            }
            us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        CHECK(hipStreamSynchronize(st));
        printf(", \"%s_us\": %.2f", name, median(us));
    };
    auto run_grid = [&](int wgs, unsigned long long *req_dev_view, volatile unsigned long long *req_host_view, int scope_system, bool slots, const char *name) {
        *req_host_view = 0;
        _mm_sfence();
        *(volatile unsigned long long *)h_resp = 0;
        memset(h_slots, 0, 1024);
        CHECK(hipMemset(d_done, 0, 64));
        CHECK(hipDeviceSynchronize());
        if (slots) echo_slots_kernel<<<wgs, 64, 0, st>>>(req_dev_view, h_slots, iters, patience);
        else echo_grid_kernel<<<wgs, 64, 0, st>>>(req_dev_view, h_resp, d_done, iters, patience, scope_system);
        std::vector<double> us;
        for (unsigned i = 1; i <= iters; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            *req_host_view = i;
            _mm_sfence();
            for (;;) {
                bool ok;
                if (slots) {
                    ok = true;
                    for (int w = 0; w < wgs; ++w) ok &= __atomic_load_n(h_slots + w, __ATOMIC_ACQUIRE) >= i;
                } else ok = __atomic_load_n(h_resp, __ATOMIC_ACQUIRE) >= i;
                if (ok) break;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3)) { printf(", \"%s\": \"timeout at %u\"", name, i); CHECK(hipStreamSynchronize(st)); return; }
            }
            us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        CHECK(hipStreamSynchronize(st));
        printf(", \"%s_us\": %.2f", name, median(us));
    };
    run_single(h_req, h_req, "one_lane_pinned_request");
    if (coarse_host) run_single(d_req, d_req, "one_lane_vram_request");
    if (fine_host) run_single(d_fine, d_fine, "one_lane_finegrained_vram_request");
    for (int wgs : {32, 64}) {
        char nm[96];
        snprintf(nm, sizeof nm, "grid%d_pinned_request_counter", wgs);
        run_grid(wgs, h_req, h_req, 1, false, nm);
        if (coarse_host) {
            snprintf(nm, sizeof nm, "grid%d_vram_request_counter", wgs);
            run_grid(wgs, d_req, d_req, 0, false, nm);
            snprintf(nm, sizeof nm, "grid%d_vram_request_slots", wgs);
            run_grid(wgs, d_req, d_req, 0, true, nm);
        }
        if (fine_host) {
            snprintf(nm, sizeof nm, "grid%d_finegrained_request_counter", wgs);
            run_grid(wgs, d_fine, d_fine, 0, false, nm);
        }
    }
    printf("}\n");
    return 0;
}
