// native_bench.cpp - the headline measurement with nothing but the C ABI and the HIP runtime (no Python,
// no torch): a synthetic haystack generated on the device, a 16-byte absent needle, K timed search_in calls.
//   hipcc -O2 -std=c++17 -I include tools/native_bench.cpp -o /tmp/native_bench \
//         -L sliceslice-rs_amd/csrc -lsliceslice_hip -Wl,-rpath,$PWD/sliceslice-rs_amd/csrc
//   /tmp/native_bench [GiB=64] [steps=20]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sliceslice_hip.h"

#define CK(x)                                                                  \
    do {                                                                       \
        if ((x) != 0) {                                                        \
            std::fprintf(stderr, "%s failed: %s\n", #x, ss_last_error());      \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? std::atof(argv[1]) : 64.0;
    const int steps = argc > 2 ? std::atoi(argv[2]) : 20;
    const size_t len = (size_t)(gib * (double)(1ull << 30));
    void *d_hay = nullptr;
    if (hipMalloc(&d_hay, len) != hipSuccess) {
        std::fprintf(stderr, "hipMalloc(%zu) failed\n", len);
        return 1;
    }
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
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < steps; ++k) {
        CK(ss_search_device(s, d_hay, len, nullptr, &found));
        float ms = 0;
        CK(ss_searcher_last_kernel_ms(s, &ms));
        kms.push_back(ms);
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double ksum = 0;
    for (float m : kms) ksum += m;
    char name[256];
    int cus = 0;
    size_t mem = 0;
    CK(ss_device_info(name, sizeof name, &cus, &mem));
    std::printf("{\"device\": \"%s\", \"haystack_bytes\": %zu, \"found\": %d, \"steps\": %d, \"ms_per_step\": %.4f, "
                "\"value_gbps\": %.1f, \"kernel_ms_avg\": %.4f, \"kernel_gbps\": %.1f, \"frac_of_8tbps\": %.4f}\n",
                name, len, found, steps, wall / steps * 1e3, (double)len * steps / wall / 1e9, ksum / steps,
                (double)len / (ksum / steps) / 1e6, (double)len / (ksum / steps) / 1e6 / 8000.0);
    ss_searcher_free(s);
    (void)hipFree(d_hay);
    return found != 0;
}
