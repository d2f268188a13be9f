// Stress of the HOST side of the C-ABI library: the flag-slot pool under more concurrent callers than slots,
// the epoch wrap, the staging leases of the host / file front ends under contention, the communicator
// objects (one rank; all visible devices in one process) and the error paths.  Built twice by
// tests/test_gpu_native.py: plainly, and with -fsanitize=address,undefined against the sanitized build of the
// library - the counterpart of the reference's ASAN job (.github/workflows/check.yml:42-58) for the part of
// this build that is ordinary C++.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "sliceslice_hip.h"
#include "sliceslice_hip_tuning.h"      // built against a hooks build (ASan / TSan): fault injection, epoch setters, counters

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, ss_last_error()); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static std::atomic<int> g_failures{0};
#define TCHECK(cond)                                                                 \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, ss_last_error()); \
            ++g_failures;                                                            \
            return;                                                                  \
        }                                                                            \
    } while (0)

int main(int argc, char **argv)
{
    std::setvbuf(stdout, nullptr, _IONBF, 0);           // (a sanitizer abort must not swallow the progress lines)
    const int nthreads = argc > 1 ? std::atoi(argv[1]) : 128;
    const int iterations = argc > 2 ? std::atoi(argv[2]) : 16;
    const size_t len = 2u << 20;
    const uint8_t needle[7] = {9, 8, 7, 6, 5, 4, 3};
    std::vector<uint8_t> h_no(len, 0), h_yes(len, 0);
    std::memcpy(h_yes.data() + len - 7, needle, 7);
    uint8_t *d_no = nullptr, *d_yes = nullptr;
    CHECK(hipMalloc((void **)&d_no, len) == hipSuccess && hipMalloc((void **)&d_yes, len) == hipSuccess);
    CHECK(hipMemcpy(d_no, h_no.data(), len, hipMemcpyHostToDevice) == hipSuccess);
    CHECK(hipMemcpy(d_yes, h_yes.data(), len, hipMemcpyHostToDevice) == hipSuccess);
    const std::string path = "/tmp/host_stress_haystack.bin";
    {
        std::FILE *fh = std::fopen(path.c_str(), "wb");
        CHECK(fh != nullptr);
        CHECK(std::fwrite(h_yes.data(), 1, len, fh) == len);
        std::fclose(fh);
    }

    // ---- error paths ----
    ss_searcher *bad = nullptr;
    CHECK(ss_searcher_with_position(needle, 7, 7, &bad) == SS_ERR_POSITION && bad == nullptr);
    CHECK(ss_searcher_with_position(needle, 1, 1, &bad) == SS_ERR_POSITION);
    CHECK(ss_searcher_new(nullptr, 3, &bad) == SS_ERR_ARGUMENT);
    CHECK(ss_searcher_new(needle, 7, nullptr) == SS_ERR_ARGUMENT);
    ss_comm *badc = nullptr;
    uint8_t id[SS_UNIQUE_ID_BYTES] = {0};
    CHECK(ss_comm_init_rank(id, 0, 0, &badc) == SS_ERR_ARGUMENT && badc == nullptr);
    CHECK(ss_comm_init_rank(id, 2, 2, &badc) == SS_ERR_ARGUMENT);
    ss_comm_set *bads = nullptr;
    CHECK(ss_comm_init_all(0, nullptr, &bads) == SS_ERR_ARGUMENT);
    CHECK(ss_comm_init_all(4096, nullptr, &bads) == SS_ERR_ARGUMENT);
    ss_searcher_free(nullptr);
    ss_comm_free(nullptr);
    ss_comm_set_free(nullptr);

    ss_searcher *s = nullptr;
    CHECK(ss_searcher_new(needle, 7, &s) == SS_OK);
    CHECK(ss_searcher_set_filter3(s, 3, 2, 2) == SS_ERR_POSITION && ss_searcher_set_filter3(s, 0, 7, 7) == SS_ERR_POSITION);
    size_t fa = 9, fb = 9;
    { size_t fc = 0; CHECK(ss_searcher_filter3(s, &fa, &fb, &fc) == SS_OK && fa <= fb && fb < 7); }
    CHECK(ss_searcher_set_timing(s, 1) == SS_OK);

    // ---- more concurrent callers than flag slots, all entry points mixed ----
    std::vector<std::thread> pool;
    for (int k = 0; k < nthreads; ++k)
        pool.emplace_back([&, k]() {
            hipStream_t st = nullptr;
            TCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
            for (int it = 0; it < iterations; ++it) {
                int found = -1;
                uint64_t pos = 1;
                switch ((k + it) % 6) {
                case 0:
                    TCHECK(ss_search_device(s, d_yes, len, st, &found) == SS_OK && found == 1);
                    break;
                case 1: {
                    TCHECK(ss_search_device(s, d_no, len, st, &found) == SS_OK && found == 0);
                    float ms = -1;
                    TCHECK(ss_searcher_last_kernel_ms(s, &ms) == SS_OK && ms >= 0);
                    break;
                }
                case 2:
                    TCHECK(ss_find_device(s, d_yes, len, st, &pos) == SS_OK && pos == len - 7);
                    TCHECK(ss_find_device(s, d_no, len, st, &pos) == SS_OK && pos == SS_NPOS);
                    break;
                case 3:
                    TCHECK(ss_search_host(s, h_yes.data(), len, &found) == SS_OK && found == 1);
                    TCHECK(ss_search_host(s, h_no.data(), len, &found) == SS_OK && found == 0);
                    break;
                case 4:
                    TCHECK(ss_find_host(s, h_yes.data(), len, &pos) == SS_OK && pos == len - 7);
                    break;
                default:
                    TCHECK(ss_search_file(s, path.c_str(), &found) == SS_OK && found == 1);
                    break;
                }
            }
            (void)hipStreamDestroy(st);
        });
    for (auto &t : pool) t.join();
    CHECK(g_failures == 0);

    // ---- epoch wrap of the flag slots ----
    CHECK(ss_debug_set_epochs(s, INT_MAX - 3) == SS_OK);
    for (int it = 0; it < 8; ++it) {
        int found = -1;
        CHECK(ss_search_device(s, d_yes, len, nullptr, &found) == SS_OK && found == 1);
        CHECK(ss_search_device(s, d_no, len, nullptr, &found) == SS_OK && found == 0);
        CHECK(ss_search_host(s, h_yes.data(), len, &found) == SS_OK && found == 1);
    }

    // ---- communicators: one rank; every visible device from this process ----
    {
        ss_comm *c = nullptr;
        CHECK(ss_comm_unique_id(id) == SS_OK);
        CHECK(ss_comm_init_rank(id, 1, 0, &c) == SS_OK);
        int nr = 0;
        CHECK(ss_comm_count(c, &nr) == SS_OK && nr == 1);
        CHECK(ss_debug_set_comm_epoch(c, nullptr, INT_MAX - 2) == SS_OK);
        for (int it = 0; it < 5; ++it) {
            int found = -1;
            uint64_t pos = 1;
            CHECK(ss_search_sharded(s, d_yes, len, c, nullptr, &found) == SS_OK && found == 1);
            CHECK(ss_search_sharded(s, d_no, len, c, nullptr, &found) == SS_OK && found == 0);
            CHECK(ss_find_sharded(s, d_yes, len, 1000, c, nullptr, &pos) == SS_OK && pos == 1000 + len - 7);
        }
        // a rank-local failure (injected: the scan is never enqueued) still goes through the collective, comes back as
        // THIS rank's error - not a hang, not SS_ERR_PEER - and leaves the communicator usable
        for (int it = 0; it < 3; ++it) {
            int found = -1;
            uint64_t pos = 1;
            CHECK(ss_debug_fail_next_scans(s, 1) == SS_OK);
            CHECK(ss_search_sharded(s, d_yes, len, c, nullptr, &found) == SS_ERR_HIP);
            CHECK(std::strstr(ss_last_error(), "injected") != nullptr);
            CHECK(ss_search_sharded(s, d_yes, len, c, nullptr, &found) == SS_OK && found == 1);
            CHECK(ss_debug_fail_next_scans(s, 1) == SS_OK);
            CHECK(ss_find_sharded(s, d_yes, len, 1000, c, nullptr, &pos) == SS_ERR_HIP);
            CHECK(ss_find_sharded(s, d_yes, len, 1000, c, nullptr, &pos) == SS_OK && pos == 1000 + len - 7);
            CHECK(ss_search_sharded(s, d_no, len, c, nullptr, &found) == SS_OK && found == 0);
        }
        // one search at a time per communicator: two threads on ONE communicator get an answer or a refusal, never a
        // corrupted answer
        {
            std::atomic<int> ok{0}, refused{0};
            std::vector<std::thread> two;
            for (int t = 0; t < 2; ++t)
                two.emplace_back([&, t]() {
                    hipStream_t st = nullptr;
                    TCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
                    for (int it = 0; it < 200; ++it) {
                        int found = -1;
                        const int rc = ss_search_sharded(s, t ? d_yes : d_no, len, c, st, &found);
                        if (rc == SS_OK) { TCHECK(found == t); ++ok; }
                        else { TCHECK(rc == SS_ERR_ARGUMENT); ++refused; }
                    }
                    (void)hipStreamDestroy(st);
                });
            for (auto &t : two) t.join();
            CHECK(g_failures == 0 && ok > 0);
            std::printf("communicator shared by two threads: %d answered, %d refused\n", ok.load(), refused.load());
        }
        ss_comm_free(c);
    }
    {
        int ndev = 0;
        CHECK(hipGetDeviceCount(&ndev) == hipSuccess && ndev >= 1);
        if (ndev > 8) ndev = 8;
        ss_comm_set *set = nullptr;
        CHECK(ss_comm_init_all(ndev, nullptr, &set) == SS_OK);
        std::vector<uint8_t *> bufs(ndev, nullptr);
        std::vector<const void *> shards(ndev);
        std::vector<size_t> lens(ndev, len);
        std::vector<uint64_t> begins(ndev);
        for (int g = 0; g < ndev; ++g) {
            CHECK(hipSetDevice(g) == hipSuccess && hipMalloc((void **)&bufs[g], len) == hipSuccess);
            CHECK(hipMemcpy(bufs[g], g == ndev - 1 ? h_yes.data() : h_no.data(), len, hipMemcpyHostToDevice) == hipSuccess);
            shards[g] = bufs[g];
            begins[g] = (uint64_t)g * len;
        }
        CHECK(hipSetDevice(0) == hipSuccess);
        for (int mode : {SS_COMBINE_RCCL, SS_COMBINE_HOST}) {
            CHECK(ss_comm_set_combine(set, mode) == SS_OK);
            for (int it = 0; it < 4; ++it) {
                int found = -1;
                uint64_t pos = 1;
                CHECK(ss_search_sharded_all(s, shards.data(), lens.data(), set, &found) == SS_OK && found == 1);
                CHECK(ss_find_sharded_all(s, shards.data(), lens.data(), begins.data(), set, &pos) == SS_OK &&
                      pos == (uint64_t)(ndev - 1) * len + len - 7);
                int cur = -1;
                CHECK(hipGetDevice(&cur) == hipSuccess && cur == 0);
            }
        }
        // one search at a time per set (include/sliceslice_hip.h): the second concurrent call is refused with
        // SS_ERR_ARGUMENT instead of corrupting both answers
        {
            std::atomic<int> ok{0}, refused{0};
            std::vector<std::thread> two;
            for (int t = 0; t < 2; ++t)
                two.emplace_back([&, t]() {
                    for (int it = 0; it < 200; ++it) {
                        int found = -1;
                        uint64_t pos = 1;
                        const int rc = (it & 1) ? ss_find_sharded_all(s, shards.data(), lens.data(), begins.data(), set, &pos)
                                                : ss_search_sharded_all(s, shards.data(), lens.data(), set, &found);
                        if (rc == SS_OK) {
                            if (it & 1) TCHECK(pos == (uint64_t)(ndev - 1) * len + len - 7);
                            else TCHECK(found == 1);
                            ++ok;
                        } else {
                            TCHECK(rc == SS_ERR_ARGUMENT);
                            ++refused;
                        }
                    }
                });
            for (auto &t : two) t.join();
            CHECK(g_failures == 0 && ok > 0);
            std::printf("communicator set shared by two threads: %d answered, %d refused\n", ok.load(), refused.load());
        }
        for (int g = 0; g < ndev; ++g) {
            CHECK(hipSetDevice(g) == hipSuccess);
            (void)hipFree(bufs[g]);
        }
        CHECK(hipSetDevice(0) == hipSuccess);
        ss_comm_set_free(set);
    }

    // ---- ss_searcher_set_filter* against running searches: refused while any search is in flight, accepted in between;
    //      whatever triple a search ends up with, its answer is right (src/lib.rs:375-378) ----
    {
        std::atomic<bool> stop{false};
        std::atomic<int> accepted{0}, refused{0}, searched{0};
        std::vector<std::thread> searchers;
        for (int t = 0; t < 4; ++t)
            searchers.emplace_back([&, t]() {
                hipStream_t st = nullptr;
                TCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
                while (!stop.load()) {
                    int found = -1;
                    TCHECK(ss_search_device(s, (t & 1) ? d_yes : d_no, len, st, &found) == SS_OK && found == (t & 1));
                    ++searched;
                }
                (void)hipStreamDestroy(st);
            });
        const size_t pairs[4][3] = {{0, 6, 5}, {1, 5, 3}, {0, 1, 2}, {2, 6, 4}};
        while (searched.load() < 64 && g_failures == 0) std::this_thread::yield();      // all four are searching by now
        for (int it = 0; it < 20000 || (refused == 0 && it < 2000000); ++it) {
            const size_t *q = pairs[it & 3];
            const int rc = (it & 4) ? ss_searcher_set_filter3(s, q[0], q[1], q[1]) : ss_searcher_set_filter3(s, q[0], q[1], q[2]);
            if (rc == SS_OK) ++accepted;
            else { CHECK(rc == SS_ERR_ARGUMENT); ++refused; }
        }
        stop.store(true);
        for (auto &t : searchers) t.join();
        CHECK(g_failures == 0);
        CHECK(refused > 0);                       // four threads searching back to back: most attempts meet one in flight
        size_t a = 9, b = 9, c3 = 9;
        CHECK(ss_searcher_set_filter3(s, 0, 6, 5) == SS_OK);          // nothing in flight now
        CHECK(ss_searcher_filter3(s, &a, &b, &c3) == SS_OK && a == 0 && b == 6 && c3 == 5);
        std::printf("set_filter against 4 searching threads: %d accepted, %d refused\n", accepted.load(), refused.load());
    }

    // The resident search service from several threads (requests queue on its mutex), searchers built and dropped while it is
    // resident (control blocks from the slab pool, written through the BAR), bind / unbind, a lease short enough to end the
    // residency now and then - and the stop: the service's memory, its mutex included, is freed OUTSIDE its own lock.
    if (!std::getenv("HST_SKIP_SERVICE")) {
        ss_service *sv = nullptr;
        const double lease = std::getenv("HST_SVC_LEASE_MS") ? std::atof(std::getenv("HST_SVC_LEASE_MS")) : 1.0;
        const bool build_inside = !std::getenv("HST_SVC_NO_NEW");
        CHECK(ss_service_start(32, lease, &sv) == SS_OK);
        CHECK(ss_service_bind(sv, d_yes, len) == SS_OK);
        std::vector<std::thread> callers;
        for (int t = 0; t < 4; ++t)
            callers.emplace_back([&, t]() {
                for (int it = 0; it < 400; ++it) {
                    int found = -1;
                    TCHECK(ss_service_search(sv, s, (it & 1) ? d_yes : d_no, len, &found) == SS_OK && found == (it & 1));
                    if (build_inside && (it & 63) == 17) {
                        const uint8_t other[5] = {9, 9, 8, 8, (uint8_t)t};
                        ss_searcher *tmp = nullptr;
                        TCHECK(ss_searcher_new(other, 5, &tmp) == SS_OK);
                        TCHECK(ss_service_search(sv, tmp, d_yes, len, &found) == SS_OK && found == 0);
                        ss_searcher_free(tmp);
                    }
                    if ((it & 127) == 100) std::this_thread::sleep_for(std::chrono::milliseconds(2));    // > lease
                }
            });
        for (auto &t : callers) t.join();
        std::puts("service callers joined");
        CHECK(g_failures == 0);
        uint64_t requests = 0, launches = 0, settled = 0;
        CHECK(ss_service_counters(sv, &requests, &launches, &settled) == SS_OK && requests >= 1600 && launches >= 1 && settled > 0);
        CHECK(ss_service_bind(sv, nullptr, 0) == SS_OK);
        int found = -1;
        CHECK(ss_service_search(sv, s, d_yes, len, &found) == SS_OK && found == 1);
        ss_service_stop(sv);
        std::puts("service stopped");
        std::printf("service from 4 threads: %llu requests, %llu residencies, %llu without an acquire\n", (unsigned long long)requests,
                    (unsigned long long)launches, (unsigned long long)settled);
    }

    // ss_service_stop against callers that are INSIDE the service: one long request holds the service's mutex, three more calls
    // have entered and wait for it, then the stop arrives.  Whoever gets the mutex before the stop is answered, whoever gets it
    // after is turned away (SS_ERR_ARGUMENT), and the stop frees the service only when all four have left - a caller blocked on
    // the mutex at that moment used to wake into freed memory (ASan / TSan builds of this test are the proof).
    if (!std::getenv("HST_SKIP_SERVICE")) {
        const size_t big_len = (size_t)4 << 30;
        uint8_t *d_big = nullptr;
        CHECK(hipMalloc((void **)&d_big, big_len) == hipSuccess && hipMemset(d_big, 0, big_len) == hipSuccess);
        CHECK(hipDeviceSynchronize() == hipSuccess);
        for (int round = 0; round < 3; ++round) {
            ss_service *sv = nullptr;
            CHECK(ss_service_start(8, 5.0, &sv) == SS_OK);
            int warm = -1;
            CHECK(ss_service_search(sv, s, d_no, len, &warm) == SS_OK && warm == 0);
            std::atomic<int> entering{0}, answered{0}, refused{0};
            std::vector<std::thread> callers;
            for (int t = 0; t < 4; ++t)
                callers.emplace_back([&, t]() {
                    (void)hipSetDevice(0);
                    int found = -1;
                    entering.fetch_add(1);
                    const int rc = t == 0 ? ss_service_search(sv, s, d_big, big_len, &found)        // ~ tens of ms on 8 workgroups
                                          : ss_service_search(sv, s, (t & 1) ? d_yes : d_no, len, &found);
                    if (rc == SS_OK) {
                        TCHECK(found == (t == 0 ? 0 : (t & 1)));
                        answered.fetch_add(1);
                    } else {
                        TCHECK(rc == SS_ERR_ARGUMENT);
                        refused.fetch_add(1);
                    }
                });
            while (entering.load() < 4) std::this_thread::yield();
            std::this_thread::sleep_for(std::chrono::milliseconds(3));          // all four are inside ss_service_search by now
            ss_service_stop(sv);
            for (auto &t : callers) t.join();
            CHECK(answered.load() + refused.load() == 4 && answered.load() >= 1);
            std::printf("stop against 4 callers inside the service: %d answered, %d turned away\n", answered.load(), refused.load());
        }
        CHECK(g_failures == 0);
        (void)hipFree(d_big);
    }

    // Batched calls and batch plans from several threads: the unplanned calls share per-stream scratch (same stream here: the
    // entry's mutex keeps each call's two launches adjacent), every thread owns a bool plan and a find plan and runs them on its
    // own stream (one run at a time per plan is the caller's side of the contract), plans are created and freed while other
    // threads' runs are in flight, and a plan made for another device's number is refused.  Two problems per call: the needle in
    // d_yes (present at len - 7) and in d_no (absent), addressed as ranges of ONE base pointer.
    {
        const uint8_t *lo = d_yes < d_no ? d_yes : d_no;
        const uint64_t off_yes = (uint64_t)(d_yes - lo), off_no = (uint64_t)(d_no - lo);
        const uint64_t h_hb[2] = {off_yes, off_no}, h_he[2] = {off_yes + len, off_no + len}, h_nb[2] = {0, 0}, h_ne[2] = {7, 7};
        uint64_t *d_rng = nullptr;
        uint8_t *d_nd = nullptr;
        CHECK(hipMalloc((void **)&d_rng, 8 * sizeof(uint64_t)) == hipSuccess && hipMalloc((void **)&d_nd, 16) == hipSuccess);
        CHECK(hipMemcpy(d_rng, h_hb, 16, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(d_rng + 2, h_he, 16, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(hipMemcpy(d_rng + 4, h_nb, 16, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(d_rng + 6, h_ne, 16, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(hipMemcpy(d_nd, needle, 7, hipMemcpyHostToDevice) == hipSuccess);
        std::vector<std::thread> callers;
        for (int t = 0; t < 4; ++t)
            callers.emplace_back([&, t]() {
                (void)hipSetDevice(0);
                hipStream_t st = nullptr;
                TCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
                int *d_found = nullptr;
                uint64_t *d_pos = nullptr;
                TCHECK(hipMalloc((void **)&d_found, 8) == hipSuccess && hipMalloc((void **)&d_pos, 16) == hipSuccess);
                for (int round = 0; round < 6; ++round) {
                    ss_batch_plan *pb = nullptr, *pf = nullptr;
                    TCHECK(ss_batch_plan_create(lo, d_rng, d_rng + 2, d_nd, d_rng + 4, d_rng + 6, nullptr, 2, 0, st, &pb) == SS_OK);
                    TCHECK(ss_batch_plan_create(lo, d_rng, d_rng + 2, d_nd, d_rng + 4, d_rng + 6, nullptr, 2, 1, st, &pf) == SS_OK);
                    for (int it = 0; it < 25; ++it) {
                        int h_found[2] = {-1, -1};
                        uint64_t h_pos[2] = {1, 1};
                        TCHECK(hipMemsetAsync(d_found, 0x5a, 8, st) == hipSuccess && hipMemsetAsync(d_pos, 0x5a, 16, st) == hipSuccess);
                        TCHECK(ss_batch_plan_run(pb, st, d_found) == SS_OK && ss_batch_plan_run(pf, st, d_pos) == SS_OK);
                        TCHECK(hipMemcpyAsync(h_found, d_found, 8, hipMemcpyDeviceToHost, st) == hipSuccess);
                        TCHECK(hipMemcpyAsync(h_pos, d_pos, 16, hipMemcpyDeviceToHost, st) == hipSuccess);
                        TCHECK(hipStreamSynchronize(st) == hipSuccess);
                        TCHECK(h_found[0] == 1 && h_found[1] == 0 && h_pos[0] == len - 7 && h_pos[1] == SS_NPOS);
                        // the unplanned forms on the SHARED null stream (per-stream scratch, one entry for all four threads)
                        if ((it & 3) == t) {
                            TCHECK(ss_search_batched(lo, d_rng, d_rng + 2, d_nd, d_rng + 4, d_rng + 6, nullptr, 2, nullptr, d_found) == SS_OK);
                            TCHECK(hipMemcpy(h_found, d_found, 8, hipMemcpyDeviceToHost) == hipSuccess && h_found[0] == 1 && h_found[1] == 0);
                        }
                    }
                    ss_batch_plan_free(pf);
                    ss_batch_plan_free(pb);
                }
                (void)hipFree(d_pos);
                (void)hipFree(d_found);
                (void)hipStreamDestroy(st);
            });
        for (auto &t : callers) t.join();
        CHECK(g_failures == 0);
        ss_batch_plan *none = nullptr;
        CHECK(ss_batch_plan_create(lo, d_rng, d_rng + 2, d_nd, d_rng + 4, d_rng + 6, nullptr, 0, 0, nullptr, &none) == SS_ERR_ARGUMENT && none == nullptr);
        CHECK(ss_batch_plan_run(nullptr, nullptr, d_rng) == SS_ERR_ARGUMENT);
        ss_batch_plan_free(nullptr);
        std::puts("batch plans from 4 threads ok");
        (void)hipFree(d_nd);
        (void)hipFree(d_rng);
    }

    // ---- what a searcher learns about a large haystack (ss_census.hip): the candidate census and the per-device byte histogram,
    // looked up and started by many threads on ONE handle and by several handles on one haystack at once; a haystack whose most
    // frequent bytes look rare to the static ranking, so that the histogram's triple is adopted while the others keep searching
    {
        const size_t big = (260u << 20) + 17;
        uint8_t *d_big = nullptr;
        CHECK(hipMalloc((void **)&d_big, big) == hipSuccess);
        std::vector<uint8_t> h_big(big);
        for (size_t i = 0; i < big; ++i) h_big[i] = (i & 1) ? (uint8_t)(0x80 + ((i * 2654435761u) >> 27)) : (uint8_t)(0xD0 + ((i >> 1) & 1));
        const uint8_t word[] = {0xD0, 0x81, 0xD1, 0x9C, 0xD0, 0xBE, 0xD1, 0x83, 0xD0, 0x80, 0xD1, 0x8F, 0xD0, 0x99, 0xD1, 0x82, 'e', 0xD0, 0x85};
        std::memcpy(h_big.data() + big - sizeof word, word, sizeof word);
        CHECK(hipMemcpy(d_big, h_big.data(), big, hipMemcpyHostToDevice) == hipSuccess);
        ss_searcher *shared = nullptr;
        CHECK(ss_searcher_new(word, sizeof word, &shared) == SS_OK);
        std::vector<std::thread> many;
        for (int t = 0; t < 12; ++t)
            many.emplace_back([&, t]() {
                ss_searcher *mine = nullptr;
                uint8_t w2[sizeof word];
                std::memcpy(w2, word, sizeof word);
                w2[1] = (uint8_t)(0x80 + t);                    // twelve more needles on the same haystack: one histogram serves all
                TCHECK(ss_searcher_new(w2, sizeof w2, &mine) == SS_OK);
                for (int it = 0; it < 6; ++it) {
                    int found = -1;
                    uint64_t pos = 1;
                    TCHECK(ss_search_device(shared, d_big, big, nullptr, &found) == SS_OK && found == 1);
                    TCHECK(ss_find_device(shared, d_big, big, nullptr, &pos) == SS_OK && pos == big - sizeof word);
                    TCHECK(ss_search_device(shared, d_big, big - 1, nullptr, &found) == SS_OK && found == 0);      // (another length: another census)
                    TCHECK(ss_search_device(mine, d_big, big, nullptr, &found) == SS_OK && found == (t == 1));
                }
                ss_searcher_free(mine);
            });
        for (auto &t : many) t.join();
        CHECK(g_failures == 0);
        int wg = 0;
        unsigned grid = 0;
        CHECK(ss_searcher_last_launch(shared, &wg, &grid) == SS_OK && (wg == 4 || wg == 6) && grid > 0);
        ss_searcher_free(shared);
        (void)hipFree(d_big);
        std::puts("census and histogram from 12 threads ok");
    }

    ss_searcher_free(s);
    (void)hipFree(d_no);
    (void)hipFree(d_yes);
    std::remove(path.c_str());
    std::puts("host_stress_test ok");
    return 0;
}
