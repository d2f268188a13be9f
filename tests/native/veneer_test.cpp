// Exercises include/sliceslice_hip.hpp (the C++ mirror of the reference interface) on a GPU box.
// Cases are the reference's doc example (src/x86.rs:1-15, README.md:12-26) and its panic contract
// (src/x86.rs:533-543), plus a device-resident haystack.  Run by tests/test_gpu_native.py.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "sliceslice_hip.hpp"

using sliceslice::hip::DeviceSlice;
using sliceslice::hip::DynamicHipSearcher;
using sliceslice::hip::PositionPanic;

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main()
{
    auto searcher = DynamicHipSearcher::new_("ipsum");
    CHECK(searcher.search_in(std::string("Lorem ipsum dolor sit amet, consectetur adipiscing elit")));
    CHECK(!searcher.search_in(std::string("foo bar baz qux quux quuz corge grault garply waldo fred")));
    CHECK(searcher.position() == 4);

    bool panicked = false;
    try {
        (void)DynamicHipSearcher::with_position("foo", 3);       // dynamic_avx2_invalid_position
    } catch (const PositionPanic &) {
        panicked = true;
    }
    CHECK(panicked);
    CHECK(DynamicHipSearcher::new_("").search_in(std::string("")));            // N0
    CHECK(!DynamicHipSearcher::new_("x").search_in(std::string("")));          // MemchrSearcher, empty haystack

    // device-resident haystack, every position, misaligned pointer
    const size_t len = (4u << 20) + 3;
    std::vector<uint8_t> host(len, 0x2E);
    const std::string needle = "Maecenas commodo posuere orci a consectetur";
    std::memcpy(host.data() + len - needle.size(), needle.data(), needle.size());
    uint8_t *d = nullptr;
    CHECK(hipMalloc((void **)&d, len + 16) == hipSuccess);
    CHECK(hipMemcpy(d + 7, host.data(), len, hipMemcpyHostToDevice) == hipSuccess);
    for (size_t position = 0; position < needle.size(); ++position) {
        auto s = DynamicHipSearcher::with_position(needle, position);
        CHECK(s.search_in(DeviceSlice{d + 7, len}));
        CHECK(!s.search_in(DeviceSlice{d + 7, len - 1}));
        CHECK(s.inlined_search_in(DeviceSlice{d + 7, len}));
        CHECK(s.find(DeviceSlice{d + 7, len}) == len - needle.size());
        CHECK(s.find(DeviceSlice{d + 7, len - 1}) == DynamicHipSearcher::npos);
    }
    (void)hipFree(d);

    // host slice: find; file front end (examples/grep.rs shape)
    {
        auto s = DynamicHipSearcher::new_(needle);
        CHECK(s.find(host.data(), len) == len - needle.size());
        CHECK(s.find(host.data(), len - 1) == DynamicHipSearcher::npos);
        const std::string path = "/tmp/veneer_test_haystack.bin";
        std::FILE *fh = std::fopen(path.c_str(), "wb");
        CHECK(fh != nullptr);
        CHECK(std::fwrite(host.data(), 1, len, fh) == len);
        std::fclose(fh);
        CHECK(s.search_in_file(path));
        CHECK(!DynamicHipSearcher::new_("not in the file at all").search_in_file(path));
        std::remove(path.c_str());
    }
    std::puts("veneer_test ok");
    return 0;
}
