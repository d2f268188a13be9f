// native_bench.cpp - measurements with nothing but the C ABI and the HIP runtime in the loop (no Python, no
// torch).  Built by sliceslice-rs_amd/_build.py:build_native_bench() into tools/native_bench; bench.py runs the
// `latency` and `config1` modes and embeds their JSON in its line.
//
//   native_bench headline [GiB=64] [steps=20]     the headline measurement (synthetic haystack, 16-byte absent needle)
//   native_bench latency  [calls=2000]            per-call microseconds of ss_search_device / ss_find_device /
//                                                 ss_search_host on 1 KiB, 64 KiB, 1 MiB, 16 MiB haystacks
//   native_bench sharded  [GiB=8] [steps=50]      the multi-GPU entry points on every visible GPU: per-search overhead
//   native_bench ranks    <N> [GiB=8] [steps=200] N processes (this binary, re-executed) sharing device 0, one rank each of a native
//                                                 communicator: per-search overhead outside the kernel (launch skew + barrier + read-back).
//                                                 Needs an RCCL that allows several ranks per device: SLICESLICE_RCCL_LIB = tests/native/libfake_rccl.so
//   native_bench set      <G> [GiB=8] [steps=200] ONE process, a communicator set of G members (devices 0 .. G-1, or all on device 0 when fewer
//                                                 are visible - then under SLICESLICE_RCCL_LIB = the stand-in): what ss_search_sharded_all
//                                                 costs the host - issuing G scans + all-reduces + answer words - from per-device
//                                                 threads and from the calling thread, RCCL and host combine
//   native_bench soak     [calls=2000000]         small searches through every per-call entry point; RSS / VRAM before and after
//   native_bench config1  <i386.txt> <words.txt> [iters=5]
//   native_bench construct [searchers=2000]     what `new` costs (and with a search service resident)
//        BASELINE.json configs[0] on the GPU: one ss_search_device call per needle over the resident text - the
//        literal drop-in shape of bench/benches/i386.rs:246-256 - next to ONE ss_search_batched launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "sliceslice_hip.h"
#include "sliceslice_hip_service.h"     // the resident service is timed too: the tool links libsliceslice_hip_service.so (a superset of the product)
#include "sliceslice_hip_tuning.h"      // the synthetic haystack generator (libsliceslice_hip_tools.so)

#define CK(x)                                                                  \
    do {                                                                       \
        if ((x) != 0) {                                                        \
            std::fprintf(stderr, "%s failed: %s\n", #x, ss_last_error());      \
            return 1;                                                          \
        }                                                                      \
    } while (0)
#define HK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

using clk = std::chrono::steady_clock;
static double seconds_since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

static int headline(double gib, int steps)
{
    const size_t len = (size_t)(gib * (double)(1ull << 30));
    void *d_hay = nullptr;
    HK(hipMalloc(&d_hay, len));
    CK(ss_fill_random_device(d_hay, 0, len, 0x5EED0001ull, nullptr));
    uint8_t needle[16];
    CK(ss_fill_random_host(needle, 0, 16, 0x5EED0002ull));
    needle[8] = 0xFF;                                   // 0xFF never occurs in the haystack: absent
    ss_searcher *s = nullptr;
    CK(ss_searcher_new(needle, 16, &s));
    CK(ss_searcher_set_timing(s, 1));
    int found = 1;
    for (int w = 0; w < 5; ++w) CK(ss_search_device(s, d_hay, len, nullptr, &found));
    std::vector<float> kms;
    HK(hipDeviceSynchronize());
    const auto t0 = clk::now();
    for (int k = 0; k < steps; ++k) {
        CK(ss_search_device(s, d_hay, len, nullptr, &found));
        float ms = 0;
        CK(ss_searcher_last_kernel_ms(s, &ms));
        kms.push_back(ms);
    }
    const double wall = seconds_since(t0);
    double ksum = 0;
    for (float m : kms) ksum += m;
    char name[256];
    int cus = 0;
    size_t mem = 0;
    CK(ss_device_info(name, sizeof name, &cus, &mem));
    std::printf("{\"mode\": \"headline\", \"device\": \"%s\", \"haystack_bytes\": %zu, \"found\": %d, \"steps\": %d, "
                "\"ms_per_step\": %.4f, \"value_gbps\": %.1f, \"kernel_ms_avg\": %.4f, \"kernel_gbps\": %.1f, "
                "\"frac_of_8tbps\": %.4f}\n",
                name, len, found, steps, wall / steps * 1e3, (double)len * steps / wall / 1e9, ksum / steps,
                (double)len / (ksum / steps) / 1e6, (double)len / (ksum / steps) / 1e6 / 8000.0);
    ss_searcher_free(s);
    (void)hipFree(d_hay);
    return found != 0;
}

// median of per-call wall times, microseconds
template <class F>
static double median_us(int calls, F &&call)
{
    std::vector<double> us;
    us.reserve(calls);
    for (int k = 0; k < calls; ++k) {
        const auto t0 = clk::now();
        call();
        us.push_back(seconds_since(t0) * 1e6);
    }
    std::nth_element(us.begin(), us.begin() + us.size() / 2, us.end());
    return us[us.size() / 2];
}

__global__ void null_kernel(int *p)
{
    if (p && threadIdx.x == 9999) *p = 1;
}

static int latency(int calls)
{
    const size_t sizes[] = {1u << 10, 64u << 10, 1u << 20, 16u << 20};
    const size_t cap = 16u << 20;
    void *d_hay = nullptr;
    HK(hipMalloc(&d_hay, cap));
    CK(ss_fill_random_device(d_hay, 0, cap, 0x5EED0001ull, nullptr));
    std::vector<uint8_t> h_hay(cap);
    CK(ss_fill_random_host(h_hay.data(), 0, cap, 0x5EED0001ull));
    uint8_t needle[16];
    CK(ss_fill_random_host(needle, 0, 16, 0x5EED0002ull));
    needle[8] = 0xFF;
    ss_searcher *s = nullptr;
    CK(ss_searcher_new(needle, 16, &s));
    // a needle that IS there (the first 16 bytes of the haystack): the early-exit / found path
    ss_searcher *sp = nullptr;
    CK(ss_searcher_new(h_hay.data(), 16, &sp));
    hipStream_t st = nullptr;
    HK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int found = 0, rc = 0;
    uint64_t pos = 0;
    // the floor on this stack: an empty kernel + hipStreamSynchronize on the same stream
    for (int w = 0; w < 200; ++w) { null_kernel<<<1, 64, 0, st>>>(nullptr); (void)hipStreamSynchronize(st); }
    const double floor_us = median_us(calls, [&] { null_kernel<<<1, 64, 0, st>>>(nullptr); (void)hipStreamSynchronize(st); });
    const double launch_only_us = median_us(calls, [&] { null_kernel<<<1, 64, 0, st>>>(nullptr); });
    HK(hipStreamSynchronize(st));
    std::printf("{\"mode\": \"latency\", \"calls\": %d, \"needle_len\": 16, \"unit\": \"us per call (median)\", "
                "\"empty_kernel_plus_stream_sync\": %.2f, \"empty_kernel_launch_only\": %.2f, \"rows\": [", calls, floor_us, launch_only_us);
    bool first = true;
    ss_service *sv = nullptr;
    CK(ss_service_start(0, 0.0, &sv));
    for (size_t len : sizes) {
        for (int w = 0; w < 200; ++w) rc |= ss_search_device(s, d_hay, len, st, &found);
        const double dev_absent = median_us(calls, [&] { rc |= ss_search_device(s, d_hay, len, st, &found); });
        const int f0 = found;
        const double dev_present = median_us(calls, [&] { rc |= ss_search_device(sp, d_hay, len, st, &found); });
        const int f1 = found;
        const double find_absent = median_us(calls, [&] { rc |= ss_find_device(s, d_hay, len, st, &pos); });
        const uint64_t p0 = pos;
        const double find_present = median_us(calls, [&] { rc |= ss_find_device(sp, d_hay, len, st, &pos); });
        const uint64_t p1 = pos;
        for (int w = 0; w < 200; ++w) rc |= ss_service_search(sv, s, d_hay, len, &found);
        const double svc_absent = median_us(calls, [&] { rc |= ss_service_search(sv, s, d_hay, len, &found); });
        const int f2 = found;
        const double svc_present = median_us(calls, [&] { rc |= ss_service_search(sv, sp, d_hay, len, &found); });
        const int f3 = found;
        // the haystack bound (ss_service_bind: the caller vouches that it does not change): no acquire per request
        rc |= ss_service_bind(sv, d_hay, len);
        for (int w = 0; w < 50; ++w) rc |= ss_service_search(sv, s, d_hay, len, &found);
        const double bound_absent = median_us(calls, [&] { rc |= ss_service_search(sv, s, d_hay, len, &found); });
        const int f4 = found;
        const double bound_present = median_us(calls, [&] { rc |= ss_service_search(sv, sp, d_hay, len, &found); });
        const int f5 = found;
        rc |= ss_service_bind(sv, nullptr, 0);
        for (int w = 0; w < 50; ++w) rc |= ss_search_host(s, h_hay.data(), len, &found);
        const double host_absent = median_us(std::max(200, calls / 4), [&] { rc |= ss_search_host(s, h_hay.data(), len, &found); });
        if (rc != 0 || f0 != 0 || f1 != 1 || f2 != 0 || f3 != 1 || f4 != 0 || f5 != 1 || p0 != SS_NPOS || p1 != 0) {
            std::fprintf(stderr, "latency: wrong answer (rc %d, found %d/%d, pos %llu/%llu): %s\n", rc, f0, f1,
                         (unsigned long long)p0, (unsigned long long)p1, ss_last_error());
            return 1;
        }
        std::printf("%s{\"haystack_bytes\": %zu, \"search_device_absent\": %.2f, \"search_device_present_at_0\": %.2f, "
                    "\"find_device_absent\": %.2f, \"find_device_present_at_0\": %.2f, \"search_host_absent\": %.2f, "
                    "\"service_absent\": %.2f, \"service_present_at_0\": %.2f, \"service_bound_absent\": %.2f, "
                    "\"service_bound_present_at_0\": %.2f}",
                    first ? "" : ", ", len, dev_absent, dev_present, find_absent, find_present, host_absent, svc_absent, svc_present,
                    bound_absent, bound_present);
        first = false;
    }
    std::printf("]}\n");
    ss_service_stop(sv);
    ss_searcher_free(s);
    ss_searcher_free(sp);
    (void)hipStreamDestroy(st);
    (void)hipFree(d_hay);
    return 0;
}

static bool read_file(const char *path, std::vector<uint8_t> *out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    out->assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}

static int config1(const char *hay_path, const char *words_path, int iters)
{
    std::vector<uint8_t> hay, wordsblob;
    if (!read_file(hay_path, &hay) || !read_file(words_path, &wordsblob)) {
        std::fprintf(stderr, "cannot read %s / %s\n", hay_path, words_path);
        return 1;
    }
    std::vector<std::pair<size_t, size_t>> words;       // (begin, end) of every non-empty line
    for (size_t b = 0, i = 0; i <= wordsblob.size(); ++i)
        if (i == wordsblob.size() || wordsblob[i] == '\n') {
            if (i > b) words.emplace_back(b, i);
            b = i + 1;
        }
    void *d_hay = nullptr;
    HK(hipMalloc(&d_hay, hay.size()));
    HK(hipMemcpy(d_hay, hay.data(), hay.size(), hipMemcpyHostToDevice));
    std::vector<ss_searcher *> searchers(words.size());
    for (size_t w = 0; w < words.size(); ++w)           // searchers are prebuilt, as in bench/benches/i386.rs:247-250
        CK(ss_searcher_new(wordsblob.data() + words[w].first, words[w].second - words[w].first, &searchers[w]));
    hipStream_t st = nullptr;
    HK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    size_t hits = 0;
    int found = 0;
    for (ss_searcher *s : searchers) {                  // warm-up pass
        CK(ss_search_device(s, d_hay, hay.size(), st, &found));
        hits += found != 0;
    }
    const auto t0 = clk::now();
    for (int it = 0; it < iters; ++it)
        for (ss_searcher *s : searchers) CK(ss_search_device(s, d_hay, hay.size(), st, &found));
    const double per_call_ms = seconds_since(t0) / iters * 1e3;

    // the same per-needle loop through the resident search service: no launch per search (ss_service_search)
    ss_service *sv = nullptr;
    CK(ss_service_start(0, 0.0, &sv));
    size_t svc_hits = 0;
    for (ss_searcher *s : searchers) {                  // warm-up pass (starts the residency)
        CK(ss_service_search(sv, s, d_hay, hay.size(), &found));
        svc_hits += found != 0;
    }
    const auto ts = clk::now();
    for (int it = 0; it < iters; ++it)
        for (ss_searcher *s : searchers) CK(ss_service_search(sv, s, d_hay, hay.size(), &found));
    const double service_ms = seconds_since(ts) / iters * 1e3;
    // the reference's loop reads ONE text: bound to the service (the caller vouches that it does not change), no acquire per request
    CK(ss_service_bind(sv, d_hay, hay.size()));
    size_t bound_hits = 0;
    for (ss_searcher *s : searchers) {
        CK(ss_service_search(sv, s, d_hay, hay.size(), &found));
        bound_hits += found != 0;
    }
    const auto tb = clk::now();
    for (int it = 0; it < iters; ++it)
        for (ss_searcher *s : searchers) CK(ss_service_search(sv, s, d_hay, hay.size(), &found));
    const double bound_ms = seconds_since(tb) / iters * 1e3;
    if (std::getenv("NATIVE_BENCH_PER_NEEDLE")) {
        // where the bound loop's time goes: the fastest of `iters` timings per needle, against its length and the offset of
        // its first occurrence in the text (stderr; not part of the JSON line)
        std::vector<double> best(words.size(), 1e9);
        for (int it = 0; it < std::max(iters, 5); ++it)
            for (size_t w = 0; w < words.size(); ++w) {
                const auto a = clk::now();
                CK(ss_service_search(sv, searchers[w], d_hay, hay.size(), &found));
                best[w] = std::min(best[w], std::chrono::duration<double, std::micro>(clk::now() - a).count());
            }
        std::vector<size_t> idx(words.size());
        for (size_t w = 0; w < idx.size(); ++w) idx[w] = w;
        std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return best[a] < best[b]; });
        auto first_at = [&](size_t w) {
            const uint8_t *nd = wordsblob.data() + words[w].first;
            const size_t n = words[w].second - words[w].first;
            const void *p = memmem(hay.data(), hay.size(), nd, n);
            return p ? (size_t)((const uint8_t *)p - hay.data()) : hay.size();
        };
        std::fprintf(stderr, "per needle (bound service, best of %d): p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f us\n", std::max(iters, 5),
                     best[idx[idx.size() / 10]], best[idx[idx.size() / 2]], best[idx[idx.size() * 9 / 10]], best[idx[idx.size() * 99 / 100]],
                     best[idx.back()]);
        double by_len[4] = {0}, by_pos[4] = {0};
        size_t n_len[4] = {0}, n_pos[4] = {0};
        for (size_t w = 0; w < words.size(); ++w) {
            const size_t n = words[w].second - words[w].first, at = first_at(w);
            const int lb = n <= 4 ? 0 : n <= 8 ? 1 : n <= 16 ? 2 : 3, pb = at < 16384 ? 0 : at < 131072 ? 1 : at < 524288 ? 2 : 3;
            by_len[lb] += best[w]; ++n_len[lb];
            by_pos[pb] += best[w]; ++n_pos[pb];
        }
        std::fprintf(stderr, "mean us by needle length  <=4: %.2f (%zu)  5-8: %.2f (%zu)  9-16: %.2f (%zu)  >16: %.2f (%zu)\n",
                     by_len[0] / std::max<size_t>(n_len[0], 1), n_len[0], by_len[1] / std::max<size_t>(n_len[1], 1), n_len[1],
                     by_len[2] / std::max<size_t>(n_len[2], 1), n_len[2], by_len[3] / std::max<size_t>(n_len[3], 1), n_len[3]);
        std::fprintf(stderr, "mean us by first occurrence  <16K: %.2f (%zu)  <128K: %.2f (%zu)  <512K: %.2f (%zu)  later: %.2f (%zu)\n",
                     by_pos[0] / std::max<size_t>(n_pos[0], 1), n_pos[0], by_pos[1] / std::max<size_t>(n_pos[1], 1), n_pos[1],
                     by_pos[2] / std::max<size_t>(n_pos[2], 1), n_pos[2], by_pos[3] / std::max<size_t>(n_pos[3], 1), n_pos[3]);
        for (size_t k = 0; k < 8; ++k) {
            const size_t w = idx[idx.size() - 1 - k];
            std::fprintf(stderr, "  slow: %.2f us  '%.*s'  first at %zu\n", best[w], (int)(words[w].second - words[w].first),
                         (const char *)wordsblob.data() + words[w].first, first_at(w));
        }
        for (size_t k = 0; k < 4; ++k) {
            const size_t w = idx[k];
            std::fprintf(stderr, "  fast: %.2f us  '%.*s'  first at %zu\n", best[w], (int)(words[w].second - words[w].first),
                         (const char *)wordsblob.data() + words[w].first, first_at(w));
        }
    }
    ss_service_stop(sv);

    // the same loop as ONE launch: every needle range against the one haystack range
    const size_t W = words.size();
    std::vector<uint64_t> hb(W, 0), he(W, hay.size()), nb(W), ne(W);
    for (size_t w = 0; w < W; ++w) { nb[w] = words[w].first; ne[w] = words[w].second; }
    void *d_words = nullptr;
    uint64_t *d_rng = nullptr;
    int *d_found = nullptr;
    HK(hipMalloc(&d_words, wordsblob.size()));
    HK(hipMemcpy(d_words, wordsblob.data(), wordsblob.size(), hipMemcpyHostToDevice));
    HK(hipMalloc((void **)&d_rng, 4 * W * sizeof(uint64_t)));
    HK(hipMemcpy(d_rng, hb.data(), W * 8, hipMemcpyHostToDevice));
    HK(hipMemcpy(d_rng + W, he.data(), W * 8, hipMemcpyHostToDevice));
    HK(hipMemcpy(d_rng + 2 * W, nb.data(), W * 8, hipMemcpyHostToDevice));
    HK(hipMemcpy(d_rng + 3 * W, ne.data(), W * 8, hipMemcpyHostToDevice));
    HK(hipMalloc((void **)&d_found, W * sizeof(int)));
    std::vector<int> flags(W);
    auto batched = [&]() -> int {
        CK(ss_search_batched(d_hay, d_rng, d_rng + W, d_words, d_rng + 2 * W, d_rng + 3 * W, nullptr, W, st, d_found));
        HK(hipMemcpyAsync(flags.data(), d_found, W * sizeof(int), hipMemcpyDeviceToHost, st));
        HK(hipStreamSynchronize(st));
        return 0;
    };
    for (int w = 0; w < 3; ++w)
        if (batched()) return 1;
    const auto t1 = clk::now();
    for (int it = 0; it < iters * 4; ++it)
        if (batched()) return 1;
    const double batched_ms = seconds_since(t1) / (iters * 4) * 1e3;
    size_t bhits = 0;
    for (int f : flags) bhits += f == 1;
    // ... and with the per-needle set-up done ONCE, like the reference's prebuilt searchers (i386.rs:246-250): a batch plan, then
    // one scan launch + flag read-back per iteration
    ss_batch_plan *plan = nullptr;
    const auto tp0 = clk::now();
    CK(ss_batch_plan_create(d_hay, d_rng, d_rng + W, d_words, d_rng + 2 * W, d_rng + 3 * W, nullptr, W, 0, st, &plan));
    const double plan_create_ms = seconds_since(tp0) * 1e3;
    auto planned = [&]() -> int {
        CK(ss_batch_plan_run(plan, st, d_found));
        HK(hipMemcpyAsync(flags.data(), d_found, W * sizeof(int), hipMemcpyDeviceToHost, st));
        HK(hipStreamSynchronize(st));
        return 0;
    };
    for (int w = 0; w < 3; ++w)
        if (planned()) return 1;
    const auto t2 = clk::now();
    for (int it = 0; it < iters * 4; ++it)
        if (planned()) return 1;
    const double planned_ms = seconds_since(t2) / (iters * 4) * 1e3;
    size_t phits = 0;
    for (int f : flags) phits += f == 1;
    ss_batch_plan_free(plan);
    std::printf("{\"mode\": \"config1\", \"haystack_bytes\": %zu, \"needles\": %zu, \"hits\": %zu, \"batched_hits\": %zu, "
                "\"per_call_ms_per_iteration\": %.3f, \"per_call_us_per_search\": %.3f, "
                "\"service_ms_per_iteration\": %.3f, \"service_us_per_search\": %.3f, \"service_hits\": %zu, "
                "\"service_bound_ms_per_iteration\": %.3f, \"service_bound_us_per_search\": %.3f, \"service_bound_hits\": %zu, "
                "\"batched_ms_per_iteration\": %.4f, \"planned_ms_per_iteration\": %.4f, \"plan_create_ms\": %.3f, \"planned_hits\": %zu, "
                "\"reference_published_ms\": 35.181, "
                "\"note\": \"per-call = one ss_search_device (launch + completion word) per needle, natively; service = the same loop "
                "through the resident search service (ss_service_search: no launch per search), bound = the same with the text bound to the service "
                "(ss_service_bind: no cache acquire per request); batched = one ss_search_batched call + flag read-back for all needles; "
                "planned = ss_batch_plan_create once (the reference builds its searchers once too), then one ss_batch_plan_run + read-back "
                "per iteration\"}\n",
                hay.size(), W, hits, bhits, per_call_ms, per_call_ms * 1e3 / (double)W, service_ms, service_ms * 1e3 / (double)W, svc_hits,
                bound_ms, bound_ms * 1e3 / (double)W, bound_hits, batched_ms, planned_ms, plan_create_ms, phits);
    for (ss_searcher *s : searchers) ss_searcher_free(s);
    (void)hipFree(d_found);
    (void)hipFree(d_rng);
    (void)hipFree(d_words);
    (void)hipFree(d_hay);
    (void)hipStreamDestroy(st);
    if (svc_hits != W || bound_hits != W || phits != W) return 1;
    return (hits == W && bhits == W) ? 0 : 1;           // every word of words.txt occurs in i386.txt (tests/i386.rs:61-70)
}

// Per-search overhead of the multi-GPU entry points on however many GPUs are visible (one on the test box): the same
// shard through ss_search_device, ss_search_sharded (one process per GPU; here a communicator of ONE rank) and
// ss_search_sharded_all (one process, all visible devices; RCCL combine and host combine).  wall - kernel = what the
// launch, the all-reduce, the read-back and the stream waits cost per search.
static int sharded(double gib, int steps)
{
    int ndev = 0;
    HK(hipGetDeviceCount(&ndev));
    if (ndev > 8) ndev = 8;
    const size_t total = (size_t)(gib * (double)(1ull << 30));
    uint8_t needle[16];
    CK(ss_fill_random_host(needle, 0, 16, 0x5EED0002ull));
    needle[8] = 0xFF;
    ss_searcher *s = nullptr;
    CK(ss_searcher_new(needle, 16, &s));
    CK(ss_searcher_set_timing(s, 1));
    std::vector<void *> bufs(ndev);
    std::vector<const void *> shards(ndev);
    std::vector<size_t> lens(ndev);
    for (int g = 0; g < ndev; ++g) {
        size_t b = 0, e = 0;
        CK(ss_shard_range(total, 16, ndev, g, &b, &e));
        HK(hipSetDevice(g));
        HK(hipMalloc(&bufs[g], e - b));
        CK(ss_fill_random_device(bufs[g], b, e - b, 0x5EED0001ull, nullptr));
        HK(hipDeviceSynchronize());
        shards[g] = bufs[g];
        lens[g] = e - b;
    }
    HK(hipSetDevice(0));
    ss_comm_set *set = nullptr;
    CK(ss_comm_init_all(ndev, nullptr, &set));
    uint8_t id[SS_UNIQUE_ID_BYTES];
    ss_comm *c = nullptr;
    if (ndev == 1) {                                   // a one-rank communicator only makes sense on a one-GPU box
        CK(ss_comm_unique_id(id));
        CK(ss_comm_init_rank(id, 1, 0, &c));
    }
    int found = 1;
    auto timed = [&](auto &&call) -> double {
        for (int w = 0; w < 5; ++w) call();
        const auto t0 = clk::now();
        for (int k = 0; k < steps; ++k) call();
        return seconds_since(t0) / steps * 1e3;
    };
    int rc = 0;
    const double dev_ms = ndev == 1 ? timed([&] { rc |= ss_search_device(s, shards[0], lens[0], nullptr, &found); }) : 0.0;
    float kms = 0;
    if (ndev == 1) CK(ss_searcher_last_kernel_ms(s, &kms));
    const double one_rank_ms = c ? timed([&] { rc |= ss_search_sharded(s, shards[0], lens[0], c, nullptr, &found); }) : 0.0;
    CK(ss_comm_set_combine(set, SS_COMBINE_RCCL));
    const double all_rccl_ms = timed([&] { rc |= ss_search_sharded_all(s, shards.data(), lens.data(), set, &found); });
    CK(ss_comm_set_combine(set, SS_COMBINE_HOST));
    const double all_host_ms = timed([&] { rc |= ss_search_sharded_all(s, shards.data(), lens.data(), set, &found); });
    if (rc != 0 || found != 0) {
        std::fprintf(stderr, "sharded: rc %d found %d: %s\n", rc, found, ss_last_error());
        return 1;
    }
    std::printf("{\"mode\": \"sharded\", \"devices\": %d, \"haystack_bytes\": %zu, \"steps\": %d, \"kernel_ms_one_device\": %.4f, "
                "\"search_device_ms\": %.4f, \"search_sharded_one_rank_ms\": %.4f, \"search_sharded_all_rccl_ms\": %.4f, "
                "\"search_sharded_all_host_combine_ms\": %.4f, \"aggregate_gbps_all_rccl\": %.1f}\n",
                ndev, total, steps, kms, dev_ms, one_rank_ms, all_rccl_ms, all_host_ms, (double)total / all_rccl_ms / 1e6);
    if (c) ss_comm_free(c);
    ss_comm_set_free(set);
    ss_searcher_free(s);
    for (int g = 0; g < ndev; ++g) {
        (void)hipSetDevice(g);
        (void)hipFree(bufs[g]);
    }
    return 0;
}

// The single-process multi-GPU search: host time spent ISSUING the G chains (ss_comm_set_last_issue_us), the call's wall time and
// every member's kernel time, with the per-device issue threads and without, RCCL and host combine.  G members on G devices, or
// - fewer devices visible - all on device 0 (the collective then needs the stand-in: real RCCL refuses two ranks on one device).
static int set_mode(int G, double gib, int steps)
{
    int ndev = 0;
    HK(hipGetDeviceCount(&ndev));
    const bool shared = ndev < G;
    std::vector<int> devs(G);
    for (int g = 0; g < G; ++g) devs[g] = shared ? 0 : g;
    const size_t total = (size_t)(gib * (double)(1ull << 30));
    uint8_t needle[16];
    CK(ss_fill_random_host(needle, 0, 16, 0x5EED0002ull));
    needle[8] = 0xFF;
    ss_searcher *s = nullptr;
    CK(ss_searcher_new(needle, 16, &s));
    CK(ss_searcher_set_timing(s, 1));
    std::vector<void *> bufs(G);
    std::vector<const void *> shards(G);
    std::vector<size_t> lens(G);
    for (int g = 0; g < G; ++g) {
        size_t b = 0, e = 0;
        CK(ss_shard_range(total, 16, G, g, &b, &e));
        HK(hipSetDevice(devs[g]));
        HK(hipMalloc(&bufs[g], e - b));
        CK(ss_fill_random_device(bufs[g], b, e - b, 0x5EED0001ull, nullptr));
        HK(hipDeviceSynchronize());
        shards[g] = bufs[g];
        lens[g] = e - b;
    }
    HK(hipSetDevice(0));
    ss_comm_set *set = nullptr;
    CK(ss_comm_init_all(G, devs.data(), &set));
    int counted = -1;
    if (ss_comm_set_count(set, &counted) != 0) counted = -1;
    std::printf("{\"mode\": \"set\", \"members\": %d, \"devices_visible\": %d, \"members_share_device_0\": %s, \"haystack_bytes\": %zu, "
                "\"shard_bytes\": %zu, \"steps\": %d, \"comm_count\": %d, \"rccl\": \"%s\", \"rows\": [",
                G, ndev, shared ? "true" : "false", total, lens[0], steps, counted, getenv("SLICESLICE_RCCL_LIB") ? getenv("SLICESLICE_RCCL_LIB") : "librccl");
    int found = 1, rc = 0;
    bool first = true;
    for (int combine : {SS_COMBINE_RCCL, SS_COMBINE_HOST}) {
        if (ss_comm_set_combine(set, combine) != 0) continue;          // no RCCL at all: host combine only
        for (int issue : {SS_ISSUE_THREADS, SS_ISSUE_SERIAL}) {
            CK(ss_comm_set_issue(set, issue));
            for (int w = 0; w < 10; ++w) rc |= ss_search_sharded_all(s, shards.data(), lens.data(), set, &found);
            std::vector<double> wall(steps), iss[4];
            std::vector<float> kms(G), kmax(steps);
            for (int k = 0; k < steps; ++k) {
                const auto t0 = clk::now();
                rc |= ss_search_sharded_all(s, shards.data(), lens.data(), set, &found);
                wall[k] = seconds_since(t0) * 1e6;
                float us[4];
                CK(ss_comm_set_last_issue_us(set, us));
                for (int j = 0; j < 4; ++j) iss[j].push_back(us[j]);
                CK(ss_comm_set_last_kernel_ms(set, kms.data(), G));
                kmax[k] = *std::max_element(kms.begin(), kms.end());
            }
            auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            std::vector<float> km = kmax;
            std::sort(km.begin(), km.end());
            std::printf("%s{\"combine\": \"%s\", \"issue\": \"%s\", \"call_us\": %.1f, \"issue_scans_us\": %.1f, \"issue_collective_us\": %.1f, "
                        "\"issue_answer_words_us\": %.1f, \"issue_all_us\": %.1f, \"slowest_kernel_us\": %.1f, \"outside_kernel_us\": %.1f}",
                        first ? "" : ", ", combine == SS_COMBINE_RCCL ? "rccl" : "host", issue == SS_ISSUE_THREADS ? "threads" : "serial", med(wall),
                        med(iss[0]), med(iss[1]), med(iss[2]), med(iss[3]), km[km.size() / 2] * 1e3, med(wall) - km[km.size() / 2] * 1e3);
            first = false;
        }
    }
    std::printf("]}\n");
    if (rc != 0 || found != 0) {
        std::fprintf(stderr, "set: rc %d found %d: %s\n", rc, found, ss_last_error());
        return 1;
    }
    ss_comm_set_free(set);
    ss_searcher_free(s);
    for (int g = 0; g < G; ++g) {
        (void)hipSetDevice(devs[g]);
        (void)hipFree(bufs[g]);
    }
    return 0;
}

// Soak: millions of small searches through every per-call entry point, resident set size and free device memory before
// and after - the completion-word path returns without a stream wait, so this is where an unbounded backlog of
// un-retired commands or a leaked slot would show.
static long rss_kib()
{
    std::ifstream f("/proc/self/status");
    std::string line;
    while (std::getline(f, line))
        if (line.rfind("VmRSS:", 0) == 0) return std::atol(line.c_str() + 6);
    return -1;
}

static int soak(long calls)
{
    const size_t len = 64u << 10;
    void *d_hay = nullptr;
    HK(hipMalloc(&d_hay, len));
    CK(ss_fill_random_device(d_hay, 0, len, 0x5EED0001ull, nullptr));
    std::vector<uint8_t> h_hay(len);
    CK(ss_fill_random_host(h_hay.data(), 0, len, 0x5EED0001ull));
    uint8_t needle[16];
    CK(ss_fill_random_host(needle, 0, 16, 0x5EED0002ull));
    needle[8] = 0xFF;
    ss_searcher *s = nullptr, *sp = nullptr;
    CK(ss_searcher_new(needle, 16, &s));
    CK(ss_searcher_new(h_hay.data() + 1000, 16, &sp));
    int found = 0, rc = 0;
    uint64_t pos = 0;
    for (int w = 0; w < 10000; ++w) rc |= ss_search_device(s, d_hay, len, nullptr, &found);
    size_t free0 = 0, free1 = 0, tot = 0;
    HK(hipMemGetInfo(&free0, &tot));
    const long rss0 = rss_kib();
    const auto t0 = clk::now();
    long wrong = 0;
    ss_service *sv = nullptr;
    CK(ss_service_start(0, 0.0, &sv));
    for (long k = 0; k < calls; ++k) {
        if (k % 5 == 4) {                                 // every fifth call goes through the resident search service
            rc |= ss_service_search(sv, (k & 8) ? sp : s, d_hay, len, &found);
            wrong += found != ((k & 8) ? 1 : 0);
            continue;
        }
        switch (k & 3) {
        case 0: rc |= ss_search_device(s, d_hay, len, nullptr, &found); wrong += found != 0; break;
        case 1: rc |= ss_search_device(sp, d_hay, len, nullptr, &found); wrong += found != 1; break;
        case 2: rc |= ss_search_host(sp, h_hay.data(), len, &found); wrong += found != 1; break;
        default: rc |= ss_find_device(sp, d_hay, len, nullptr, &pos); wrong += pos != 1000; break;
        }
    }
    const double secs = seconds_since(t0);
    ss_service_stop(sv);
    HK(hipDeviceSynchronize());
    HK(hipMemGetInfo(&free1, &tot));
    const long rss1 = rss_kib();
    std::printf("{\"mode\": \"soak\", \"calls\": %ld, \"seconds\": %.1f, \"us_per_call\": %.2f, \"rc\": %d, \"wrong_answers\": %ld, "
                "\"rss_kib_before\": %ld, \"rss_kib_after\": %ld, \"device_free_before\": %zu, \"device_free_after\": %zu}\n",
                calls, secs, secs / (double)calls * 1e6, rc, wrong, rss0, rss1,
                free0, free1);
    ss_searcher_free(s);
    ss_searcher_free(sp);
    (void)hipFree(d_hay);
    return (rc == 0 && wrong == 0) ? 0 : 1;
}

// What `DynamicAvx2Searcher::new` costs on this side (x86.rs:454-459: tens of nanoseconds for the reference): ss_searcher_new
// builds the searcher and puts its needle and control block on the current device.  Also with a search service resident,
// where every runtime call that waits for the device waits for the service's lease.
int construct(int count)
{
    std::vector<std::string> needles(count);
    for (int k = 0; k < count; ++k) needles[k] = "needle #" + std::to_string(k) + " of the run";
    std::vector<ss_searcher *> ss(count, nullptr);
    uint8_t *d_hay = nullptr;
    const size_t len = 1 << 16;
    HK(hipMalloc((void **)&d_hay, len));
    HK(hipMemset(d_hay, 'x', len));
    HK(hipMemcpy(d_hay + 1000, needles[count / 2].data(), needles[count / 2].size(), hipMemcpyHostToDevice));
    HK(hipDeviceSynchronize());
    ss_searcher *warm = nullptr;
    CK(ss_searcher_new((const uint8_t *)"warm-up", 7, &warm));
    int found = 0;
    CK(ss_search_device(warm, d_hay, len, nullptr, &found));
    auto t0 = clk::now();
    for (int k = 0; k < count; ++k) CK(ss_searcher_new((const uint8_t *)needles[k].data(), needles[k].size(), &ss[k]));
    const double new_us = seconds_since(t0) / count * 1e6;
    int hits = 0;
    t0 = clk::now();
    for (int k = 0; k < count; ++k) { CK(ss_search_device(ss[k], d_hay, len, nullptr, &found)); hits += found; }
    const double first_search_us = seconds_since(t0) / count * 1e6;
    t0 = clk::now();
    for (int k = 0; k < count; ++k) ss_searcher_free(ss[k]);
    const double free_us = seconds_since(t0) / count * 1e6;
    // once more with a service resident (lease 20 ms): build, search through the service, free
    ss_service *sv = nullptr;
    CK(ss_service_start(0, 0.0, &sv));
    CK(ss_service_search(sv, warm, d_hay, len, &found));
    const int n2 = std::min(count, 200);
    t0 = clk::now();
    int svc_hits = 0;
    for (int k = 0; k < n2; ++k) {
        CK(ss_searcher_new((const uint8_t *)needles[k].data(), needles[k].size(), &ss[k]));
        CK(ss_service_search(sv, ss[k], d_hay, len, &found));
        svc_hits += found;
        ss_searcher_free(ss[k]);
    }
    const double resident_us = seconds_since(t0) / n2 * 1e6;
    ss_service_stop(sv);
    ss_searcher_free(warm);
    (void)hipFree(d_hay);
    std::printf("{\"mode\": \"construct\", \"searchers\": %d, \"new_us\": %.2f, \"first_search_us\": %.2f, \"free_us\": %.2f, "
                "\"hits\": %d, \"with_service_resident_new_search_free_us\": %.2f, \"service_hits\": %d, "
                "\"note\": \"new = ss_searcher_new (needle + control block on the device); first_search = the first ss_search_device of "
                "each; with a service resident: new + ss_service_search + free per needle (a runtime call that waits for the device "
                "would wait for the 20 ms lease)\"}\n",
                count, new_us, first_search_us, free_us, hits, resident_us, svc_hits);
    const int want2 = n2 > count / 2 ? 1 : 0;
    return hits == 1 && svc_hits == want2 ? 0 : 1;
}

// ---- N ranks of a native communicator, one process each, all on device 0 -------------------------------------------------------
// What a sharded search costs OUTSIDE its kernel when more than one rank takes part: every rank scans its shard of one logical
// haystack (ss_shard_range) and calls ss_search_sharded; a search ends when the slowest rank's all-reduce has completed.  The
// ranks share ONE GPU here, so the aggregate GB/s mean nothing (their kernels share the device's bandwidth, overlapping as the
// hardware sees fit) - what the mode reports is the difference between two loops of the same ranks on the same shards at the same
// time: `steps` sharded searches, then - started together by the last of those - `steps` plain ss_search_device calls, the same
// scans without the collective.  The difference is what taking part in a collective search costs a rank per search: the
// all-reduce's barrier (launch skew between the ranks included) and the communicator's stream wait.  Real RCCL refuses several ranks per device; run it with SLICESLICE_RCCL_LIB=tests/native/libfake_rccl.so (the
// shared-memory stand-in: its all-reduce is a host barrier, a LOWER bound for a collective that crosses xGMI).
//   parent:  native_bench ranks <N> [GiB] [steps]      forks N children (itself, re-executed) and prints the summary
//   child:   native_bench rank <r> <N> <idhex> <GiB> <steps> <result-file>
#include <sys/wait.h>
#include <unistd.h>

static int rank_child(int rank, int nranks, const char *idhex, double gib, int steps, const char *result_path)
{
    uint8_t id[SS_UNIQUE_ID_BYTES];
    for (int k = 0; k < SS_UNIQUE_ID_BYTES; ++k) {
        unsigned v = 0;
        if (std::sscanf(idhex + 2 * k, "%2x", &v) != 1) return 2;
        id[k] = (uint8_t)v;
    }
    HK(hipSetDevice(0));
    const size_t total = (size_t)(gib * (double)(1ull << 30));
    uint8_t needle[16];
    CK(ss_fill_random_host(needle, 0, 16, 0x5EED0002ull));
    needle[8] = 0xFF;
    size_t b = 0, e = 0;
    CK(ss_shard_range(total, 16, nranks, rank, &b, &e));
    void *d_shard = nullptr;
    HK(hipMalloc(&d_shard, e - b));
    CK(ss_fill_random_device(d_shard, b, e - b, 0x5EED0001ull, nullptr));
    HK(hipDeviceSynchronize());
    ss_comm *c = nullptr;
    CK(ss_comm_init_rank(id, nranks, rank, &c));
    ss_searcher *s = nullptr;
    CK(ss_searcher_new(needle, 16, &s));
    CK(ss_searcher_set_timing(s, 1));
    hipStream_t st = nullptr;
    HK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int found = 1, rc = 0;
    for (int w = 0; w < 20; ++w) rc |= ss_search_sharded(s, d_shard, e - b, c, st, &found);
    double kernel_ms = 0;
    const auto t0 = clk::now();
    for (int k = 0; k < steps; ++k) {
        rc |= ss_search_sharded(s, d_shard, e - b, c, st, &found);
        float ms = 0;
        rc |= ss_searcher_last_kernel_ms(s, &ms);
        kernel_ms += ms;
    }
    const double wall_ms = seconds_since(t0) / steps * 1e3;
    // the same scans without the collective (every rank left the last all-reduce at the same moment)
    const auto t1 = clk::now();
    for (int k = 0; k < steps; ++k) rc |= ss_search_device(s, d_shard, e - b, st, &found);
    const double local_ms = seconds_since(t1) / steps * 1e3;
    if (rc != 0 || found != 0) {
        std::fprintf(stderr, "rank %d: rc %d found %d: %s\n", rank, rc, found, ss_last_error());
        return 1;
    }
    FILE *f = std::fopen(result_path, "w");
    if (!f) return 3;
    std::fprintf(f, "%.6f %.6f %zu %.6f\n", wall_ms, kernel_ms / steps, e - b, local_ms);
    std::fclose(f);
    ss_comm_free(c);
    ss_searcher_free(s);
    (void)hipStreamDestroy(st);
    (void)hipFree(d_shard);
    return 0;
}

static int ranks_parent(const char *self, int nranks, double gib, int steps)
{
    if (nranks < 1 || nranks > 64) return 2;
    uint8_t id[SS_UNIQUE_ID_BYTES];
    CK(ss_comm_unique_id(id));                        // (no HIP call: the children initialise the runtime, not this process)
    std::string hex;
    char buf[4];
    for (int k = 0; k < SS_UNIQUE_ID_BYTES; ++k) { std::snprintf(buf, sizeof buf, "%02x", id[k]); hex += buf; }
    const std::string base = "/tmp/native_bench_ranks_" + std::to_string((long)getpid()) + "_";
    std::vector<pid_t> kids;
    for (int r = 0; r < nranks; ++r) {
        const pid_t pid = fork();
        if (pid == 0) {
            const std::string rs = std::to_string(r), ns = std::to_string(nranks), gs = std::to_string(gib), ss_ = std::to_string(steps),
                              out = base + rs;
            execl("/proc/self/exe", self, "rank", rs.c_str(), ns.c_str(), hex.c_str(), gs.c_str(), ss_.c_str(), out.c_str(), (char *)nullptr);
            _exit(127);
        }
        kids.push_back(pid);
    }
    int bad = 0;
    for (pid_t pid : kids) {
        int status = 0;
        if (waitpid(pid, &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) ++bad;
    }
    double wall_max = 0, kernel_sum = 0, kernel_max = 0, local_max = 0;
    size_t shard = 0;
    for (int r = 0; r < nranks && !bad; ++r) {
        const std::string path = base + std::to_string(r);
        FILE *f = std::fopen(path.c_str(), "r");
        double w = 0, k = 0, l = 0;
        size_t sb = 0;
        if (!f || std::fscanf(f, "%lf %lf %zu %lf", &w, &k, &sb, &l) != 4) ++bad;
        if (f) std::fclose(f);
        std::remove(path.c_str());
        wall_max = std::max(wall_max, w);
        kernel_sum += k;
        kernel_max = std::max(kernel_max, k);
        local_max = std::max(local_max, l);
        shard = std::max(shard, sb);
    }
    if (bad) {
        std::fprintf(stderr, "ranks: %d rank(s) failed\n", bad);
        return 1;
    }
    const char *lib = std::getenv("SLICESLICE_RCCL_LIB");
    std::printf("{\"mode\": \"ranks\", \"ranks\": %d, \"ranks_share_one_gpu\": true, \"haystack_bytes\": %zu, \"shard_bytes\": %zu, "
                "\"steps\": %d, \"wall_ms_per_sharded_search\": %.4f, \"wall_ms_per_local_search\": %.4f, "
                "\"collective_overhead_ms\": %.4f, \"kernel_ms_sum_over_ranks\": %.4f, \"kernel_ms_slowest_rank\": %.4f, "
                "\"rccl_library\": \"%s\", "
                "\"note\": \"N processes on ONE device.  wall_ms_*: the slowest rank's time per search in a loop of ss_search_sharded calls, and "
                "in a loop of plain ss_search_device calls on the same shards run by all ranks at the same time; their difference is what "
                "the collective costs a rank per search (the all-reduce's barrier with the launch skew between ranks, the communicator's "
                "stream wait) - a lower bound for N devices, where the collective crosses xGMI.  kernel_ms_*: by events, while the other "
                "ranks' kernels share the device\"}\n",
                nranks, (size_t)(gib * (double)(1ull << 30)), shard, steps, wall_max, local_max, wall_max - local_max, kernel_sum, kernel_max,
                lib ? lib : "librccl");
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 7 && std::string(argv[1]) == "rank")
        return rank_child(std::atoi(argv[2]), std::atoi(argv[3]), argv[4], std::atof(argv[5]), std::atoi(argv[6]), argv[7]);
    if (argc > 2 && std::string(argv[1]) == "ranks")
        return ranks_parent(argv[0], std::atoi(argv[2]), argc > 3 ? std::atof(argv[3]) : 8.0, argc > 4 ? std::atoi(argv[4]) : 200);
    if (argc > 2 && std::string(argv[1]) == "set")
        return set_mode(std::atoi(argv[2]), argc > 3 ? std::atof(argv[3]) : 8.0, argc > 4 ? std::atoi(argv[4]) : 200);
    if (argc > 1 && std::string(argv[1]) == "construct") return construct(argc > 2 ? std::atoi(argv[2]) : 2000);
    const std::string mode = argc > 1 ? argv[1] : "headline";
    if (mode == "latency") return latency(argc > 2 ? std::atoi(argv[2]) : 2000);
    if (mode == "config1") {
        if (argc < 4) {
            std::fprintf(stderr, "usage: native_bench config1 <i386.txt> <words.txt> [iters]\n");
            return 2;
        }
        return config1(argv[2], argv[3], argc > 4 ? std::atoi(argv[4]) : 5);
    }
    if (mode == "soak") return soak(argc > 2 ? std::atol(argv[2]) : 2000000);
    if (mode == "sharded") return sharded(argc > 2 ? std::atof(argv[2]) : 8.0, argc > 3 ? std::atoi(argv[3]) : 50);
    if (mode == "headline") return headline(argc > 2 ? std::atof(argv[2]) : 64.0, argc > 3 ? std::atoi(argv[3]) : 20);
    // backwards compatible: native_bench <GiB> <steps>
    return headline(std::atof(argv[1]), argc > 2 ? std::atoi(argv[2]) : 20);
}
