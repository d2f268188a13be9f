// set_threads_test.cpp - the per-device issue threads of a communicator set (ss_comm.hip: SetWorker) under the sanitizer builds:
// G members on device 0 (the shared-memory RCCL stand-in allows that: SLICESLICE_RCCL_LIB=tests/native/libfake_rccl.so), searches
// with one issue thread per member and with everything issued from the caller, both combines, a second thread contending for the
// set (refused, never corrupting), injected scan failures that must get through the collective, set create / free in a loop
// (threads started and joined).  Test infrastructure; links against a hooks build of the library.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "sliceslice_hip.h"
#include "sliceslice_hip_tuning.h"

static std::atomic<int> g_failures{0};
#define CHECK(x)                                                                                   \
    do {                                                                                           \
        if (!(x)) {                                                                                \
            std::fprintf(stderr, "%s:%d: CHECK(%s) failed: %s\n", __FILE__, __LINE__, #x, ss_last_error()); \
            std::exit(1);                                                                          \
        }                                                                                          \
    } while (0)
#define TCHECK(x)                                                                                  \
    do {                                                                                           \
        if (!(x)) {                                                                                \
            std::fprintf(stderr, "%s:%d: TCHECK(%s) failed\n", __FILE__, __LINE__, #x);          \
            ++g_failures;                                                                          \
        }                                                                                          \
    } while (0)

int main(int argc, char **argv)
{
    const int G = argc > 1 ? std::atoi(argv[1]) : 3;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 60;
    CHECK(getenv("SLICESLICE_RCCL_LIB") != nullptr);
    const size_t len = (6u << 20) + 123;
    const uint8_t needle[] = {9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 1, 2, 3, 4, 5, 6, 7};
    std::vector<uint8_t> h_no(len, 0x55), h_yes(len, 0x55);
    std::memcpy(h_yes.data() + len - sizeof needle, needle, sizeof needle);
    ss_searcher *s = nullptr;
    CHECK(ss_searcher_new(needle, sizeof needle, &s) == SS_OK);
    std::vector<int> devs(G, 0);
    std::vector<uint8_t *> bufs(G, nullptr);
    std::vector<const void *> shards(G);
    std::vector<size_t> lens(G, len);
    std::vector<uint64_t> begins(G);
    for (int g = 0; g < G; ++g) begins[g] = (uint64_t)g * len;
    for (int g = 0; g < G; ++g) {
        CHECK(hipMalloc((void **)&bufs[g], len) == hipSuccess);
        CHECK(hipMemcpy(bufs[g], h_no.data(), len, hipMemcpyHostToDevice) == hipSuccess);
        shards[g] = bufs[g];
    }
    for (int life = 0; life < 3; ++life) {                   // issue threads are started with the set and joined with it
        ss_comm_set *set = nullptr;
        CHECK(ss_comm_init_all(G, devs.data(), &set) == SS_OK);
        int counted = 0;
        CHECK(ss_comm_set_count(set, &counted) == SS_OK && counted == G);
        for (int combine : {SS_COMBINE_RCCL, SS_COMBINE_HOST})
            for (int issue : {SS_ISSUE_THREADS, SS_ISSUE_SERIAL}) {
                CHECK(ss_comm_set_combine(set, combine) == SS_OK && ss_comm_set_issue(set, issue) == SS_OK);
                for (int it = 0; it < rounds; ++it) {
                    const int where = it % G;                // the needle moves from member to member
                    if (it % 3 == 0) CHECK(hipMemcpy(bufs[where], h_yes.data(), len, hipMemcpyHostToDevice) == hipSuccess);
                    int found = -1;
                    CHECK(ss_search_sharded_all(s, shards.data(), lens.data(), set, &found) == SS_OK && found == (it % 3 == 0));
                    uint64_t pos = 1;
                    CHECK(ss_find_sharded_all(s, shards.data(), lens.data(), begins.data(), set, &pos) == SS_OK);
                    CHECK(pos == (it % 3 == 0 ? (uint64_t)where * len + len - sizeof needle : SS_NPOS));
                    if (it % 3 == 0) CHECK(hipMemcpy(bufs[where], h_no.data(), len, hipMemcpyHostToDevice) == hipSuccess);
                }
                // a member whose scan cannot be enqueued: everybody still gets through the collective, the call fails, the next one is in step
                CHECK(ss_debug_fail_next_scans(s, 1) == SS_OK);
                int found = -1;
                CHECK(ss_search_sharded_all(s, shards.data(), lens.data(), set, &found) == SS_ERR_HIP);
                CHECK(ss_debug_fail_next_scans(s, 0) == SS_OK);
                CHECK(ss_search_sharded_all(s, shards.data(), lens.data(), set, &found) == SS_OK && found == 0);
                // two threads on one set: one search at a time, the other call is refused
                std::atomic<int> ok{0}, refused{0};
                std::vector<std::thread> two;
                for (int t = 0; t < 2; ++t)
                    two.emplace_back([&]() {
                        for (int it = 0; it < rounds; ++it) {
                            int f = -1;
                            const int rc = ss_search_sharded_all(s, shards.data(), lens.data(), set, &f);
                            if (rc == SS_OK) { TCHECK(f == 0); ++ok; }
                            else { TCHECK(rc == SS_ERR_ARGUMENT); ++refused; }
                        }
                    });
                for (auto &t : two) t.join();
                CHECK(g_failures == 0 && ok > 0);
            }
        float us[4] = {0, 0, 0, 0};
        CHECK(ss_comm_set_last_issue_us(set, us) == SS_OK && us[3] > 0);
        ss_comm_set_free(set);
    }
    ss_searcher_free(s);
    for (int g = 0; g < G; ++g) (void)hipFree(bufs[g]);
    std::printf("set_threads_test ok: %d members\n", G);
    std::fflush(stdout);
    return 0;
}
