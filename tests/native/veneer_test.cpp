// Exercises include/sliceslice_hip.hpp (the C++ mirror of the reference interface) on a GPU box.
// Cases are the reference's doc example (src/x86.rs:1-15, README.md:12-26) and its panic contract
// (src/x86.rs:533-543), plus a device-resident haystack.  Run by tests/test_gpu_native.py.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "sliceslice_hip.hpp"
#ifdef SS_VENEER_WITH_SERVICE       // built a second time against libsliceslice_hip_service.so (tests/test_gpu_native.py)
#include "sliceslice_hip_service.hpp"
#endif

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
    {   // Avx2Searcher's contract (src/x86.rs:533-549): avx2_invalid_position, avx2_empty_needle
        using sliceslice::hip::HipSearcher;
        using sliceslice::hip::MemchrHipSearcher;
        bool p1 = false, p2 = false;
        try { (void)HipSearcher::with_position("foo", 3); } catch (const PositionPanic &) { p1 = true; }
        try { (void)HipSearcher::new_(""); } catch (const PositionPanic &) { p2 = true; }
        CHECK(p1 && p2);
        CHECK(HipSearcher::new_("ipsum").search_in(std::string("Lorem ipsum dolor")));
        // MemchrSearcher (src/lib.rs:303-331)
        CHECK(MemchrHipSearcher::new_('a').search_in(std::string("a")) && !MemchrHipSearcher::new_('a').search_in(std::string("")));
        CHECK(MemchrHipSearcher::new_('z').search_in(std::string("xyz")) && !MemchrHipSearcher::new_('q').search_in(std::string("xyz")));
    }
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
    // the filter bytes: with_position keeps the caller's byte (and needle[0] up to 15 apart), new_ picks rare bytes; results are the same
    {
        const std::string text = " the quick brown fox ";
        auto ref = DynamicHipSearcher::with_position(text, 20);
        auto chosen = DynamicHipSearcher::new_(text);
        CHECK(ref.filter().first == 5 && ref.filter().second == 20 && ref.filter().third == 19 && ref.position() == 20);
        auto near = DynamicHipSearcher::with_position(text, 7);
        CHECK(near.filter().first == 0 && near.filter().second == 7 && near.position() == 7);
        CHECK(chosen.filter().first == 5 && chosen.filter().second == 19 && chosen.filter().third == 9 && chosen.position() == 20);
        const std::string hay = "jumps over the quick brown fox and runs";
        CHECK(ref.search_in(hay) && chosen.search_in(hay));
        chosen.set_filter(1, 2, 3);
        CHECK(chosen.search_in(hay) && !chosen.search_in(std::string("the quick brown cat ")));
    }
    // every visible GPU from this process (one on the test box): NodeSearcher
    {
        int ndev = 0;
        CHECK(hipGetDeviceCount(&ndev) == hipSuccess && ndev >= 1);
        if (ndev > 8) ndev = 8;
        const std::string nd = "sixteen byte key";
        sliceslice::hip::NodeSearcher node(nd, ndev);
        const size_t total = (8u << 20) + 77;
        std::vector<uint8_t *> bufs(ndev);
        std::vector<DeviceSlice> shards(ndev);
        std::vector<uint64_t> begins(ndev);
        for (int g = 0; g < ndev; ++g) {
            auto [b, e] = node.shard_range(total, g);
            CHECK(hipSetDevice(node.device(g)) == hipSuccess);
            CHECK(hipMalloc((void **)&bufs[g], e - b) == hipSuccess);
            CHECK(hipMemset(bufs[g], 0x2E, e - b) == hipSuccess);
            if (g == ndev - 1) CHECK(hipMemcpy(bufs[g] + (e - b) - nd.size(), nd.data(), nd.size(), hipMemcpyHostToDevice) == hipSuccess);
            shards[g] = DeviceSlice{bufs[g], e - b};
            begins[g] = b;
        }
        CHECK(hipSetDevice(0) == hipSuccess);
        for (int mode : {SS_COMBINE_RCCL, SS_COMBINE_HOST}) {
            node.set_combine(mode);
            CHECK(node.search_in(shards.data()));
            CHECK(node.find(shards.data(), begins.data()) == total - nd.size());
        }
        sliceslice::hip::NodeSearcher other(std::string("not there at all"), ndev);
        CHECK(!other.search_in(shards.data()));
        CHECK(other.find(shards.data(), begins.data()) == DynamicHipSearcher::npos);
        for (int g = 0; g < ndev; ++g) (void)hipFree(bufs[g]);
    }
    {   // the shape of the reference's bench loop - searchers first, one search per needle (through the SearchService of
        // sliceslice_hip_service.hpp, the text bound, when built against the service library)
        const std::string text = "the quick brown fox jumps over the lazy dog; pack my box with five dozen liquor jugs";
        uint8_t *dt = nullptr;
        CHECK(hipMalloc((void **)&dt, text.size()) == hipSuccess);
        CHECK(hipMemcpy(dt, text.data(), text.size(), hipMemcpyHostToDevice) == hipSuccess);
        const char *words[] = {"quick", "lazy dog", "liquor jugs", "the", "x", "fox jumps over", "cat", "jugz", "dozens"};
        const bool expect[] = {true, true, true, true, true, true, false, false, false};
        std::vector<DynamicHipSearcher> ss;
        for (const char *w : words) ss.push_back(DynamicHipSearcher::new_(w));
#ifdef SS_VENEER_WITH_SERVICE
        sliceslice::hip::SearchService service;
        service.bind(DeviceSlice{dt, text.size()});
        for (int round = 0; round < 3; ++round)
            for (size_t k = 0; k < ss.size(); ++k) CHECK(service.search_in(ss[k], DeviceSlice{dt, text.size()}) == expect[k]);
        service.unbind();
        CHECK(service.search_in(ss[0], DeviceSlice{dt + 4, text.size() - 4}));
        std::puts("service veneer ok");
#else
        for (size_t k = 0; k < ss.size(); ++k) CHECK(ss[k].search_in(DeviceSlice{dt, text.size()}) == expect[k]);
#endif

        // BatchPlan: the same loop as ONE launch per iteration - the needles as ranges of a blob, every problem the whole text
        std::string blob;
        std::vector<uint64_t> nb, ne, hb, he;
        for (const char *w : words) {
            nb.push_back(blob.size());
            blob += w;
            ne.push_back(blob.size());
            hb.push_back(0);
            he.push_back(text.size());
        }
        const size_t count = nb.size();
        uint8_t *dn = nullptr;
        uint64_t *dr = nullptr, *dpos = nullptr;
        int *dflags = nullptr;
        CHECK(hipMalloc((void **)&dn, blob.size()) == hipSuccess && hipMalloc((void **)&dr, 4 * count * 8) == hipSuccess);
        CHECK(hipMalloc((void **)&dpos, count * 8) == hipSuccess && hipMalloc((void **)&dflags, count * 4) == hipSuccess);
        CHECK(hipMemcpy(dn, blob.data(), blob.size(), hipMemcpyHostToDevice) == hipSuccess);
        CHECK(hipMemcpy(dr, hb.data(), count * 8, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(hipMemcpy(dr + count, he.data(), count * 8, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(hipMemcpy(dr + 2 * count, nb.data(), count * 8, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(hipMemcpy(dr + 3 * count, ne.data(), count * 8, hipMemcpyHostToDevice) == hipSuccess);
        sliceslice::hip::BatchPlan plan(dt, dr, dr + count, dn, dr + 2 * count, dr + 3 * count, nullptr, count, false);
        sliceslice::hip::BatchPlan fplan(dt, dr, dr + count, dn, dr + 2 * count, dr + 3 * count, nullptr, count, true);
        for (int round = 0; round < 3; ++round) {
            std::vector<int> flags(count, -7);
            std::vector<uint64_t> pos(count, 7);
            plan.run(dflags);
            fplan.run(dpos);
            CHECK(hipMemcpy(flags.data(), dflags, count * 4, hipMemcpyDeviceToHost) == hipSuccess);
            CHECK(hipMemcpy(pos.data(), dpos, count * 8, hipMemcpyDeviceToHost) == hipSuccess);
            for (size_t k = 0; k < count; ++k) {
                const size_t at = text.find(words[k]);
                CHECK(flags[k] == (expect[k] ? 1 : 0));
                CHECK(pos[k] == (at == std::string::npos ? DynamicHipSearcher::npos : at));
            }
        }
        (void)hipFree(dflags);
        (void)hipFree(dpos);
        (void)hipFree(dr);
        (void)hipFree(dn);
        (void)hipFree(dt);
    }
    std::puts("veneer_test ok");
    return 0;
}
