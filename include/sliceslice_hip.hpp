// sliceslice_hip.hpp - header-only C++ veneer over the C ABI (sliceslice_hip.h) that mirrors the
// reference's Rust surface name for name, so a C++ caller (and the parity tests in
// tests/native/) reads like the reference's own tests:
//
//   sliceslice::hip::DynamicHipSearcher::new_(needle)                DynamicAvx2Searcher::new            src/x86.rs:454
//   sliceslice::hip::DynamicHipSearcher::with_position(needle, pos)  DynamicAvx2Searcher::with_position  src/x86.rs:468
//   searcher.search_in(haystack)                                     DynamicAvx2Searcher::search_in      src/x86.rs:523
//   searcher.inlined_search_in(haystack)                             ::inlined_search_in                 src/x86.rs:498
//   searcher.find(haystack)            leftmost offset, the Option<usize> of find_subsequence     tests/i386.rs:6-10
//   searcher.search_in_file(path)      open the file + one search_in                              examples/grep.rs:42-56
//
// Contract violations that make the reference panic (src/x86.rs:300,473) throw
// sliceslice::hip::PositionPanic; HIP / RCCL failures throw sliceslice::hip::Error.  The searcher owns
// its needle copy (the reference owns `needle: N` by value, src/x86.rs:266-271); haystacks are
// borrowed for the duration of the call.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "sliceslice_hip.h"

namespace sliceslice {
namespace hip {

struct Error : std::runtime_error {
    int code;
    Error(int c, const char *what) : std::runtime_error(what), code(c) {}
};
struct PositionPanic : Error {
    using Error::Error;
};

inline void check(int rc)
{
    if (rc == SS_OK) return;
    if (rc == SS_ERR_POSITION) throw PositionPanic(rc, ss_last_error());
    throw Error(rc, ss_last_error());
}

// A haystack already resident in device memory (caller-owned hipMalloc memory, any alignment).
struct DeviceSlice {
    const void *ptr;
    size_t len;
};

class DynamicHipSearcher {
public:
    // `new` is a keyword in C++; the reference's `new(needle)` is `new_`.
    static DynamicHipSearcher new_(const uint8_t *needle, size_t n)
    {
        ss_searcher *h = nullptr;
        check(ss_searcher_new(needle, n, &h));
        return DynamicHipSearcher(h);
    }
    static DynamicHipSearcher new_(const std::string &needle)
    {
        return new_(reinterpret_cast<const uint8_t *>(needle.data()), needle.size());
    }
    static DynamicHipSearcher with_position(const uint8_t *needle, size_t n, size_t position)
    {
        ss_searcher *h = nullptr;
        check(ss_searcher_with_position(needle, n, position, &h));
        return DynamicHipSearcher(h);
    }
    static DynamicHipSearcher with_position(const std::string &needle, size_t position)
    {
        return with_position(reinterpret_cast<const uint8_t *>(needle.data()), needle.size(), position);
    }

    DynamicHipSearcher(DynamicHipSearcher &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    DynamicHipSearcher &operator=(DynamicHipSearcher &&o) noexcept
    {
        if (this != &o) {
            ss_searcher_free(h_);
            h_ = std::exchange(o.h_, nullptr);
        }
        return *this;
    }
    DynamicHipSearcher(const DynamicHipSearcher &) = delete;
    DynamicHipSearcher &operator=(const DynamicHipSearcher &) = delete;
    ~DynamicHipSearcher() { ss_searcher_free(h_); }

    // search_in(&[u8]) for a host slice: staged to the device and scanned there (PCIe-bound).
    bool search_in(const uint8_t *haystack, size_t len) const
    {
        int found = 0;
        check(ss_search_host(h_, haystack, len, &found));
        return found != 0;
    }
    bool search_in(const std::string &haystack) const
    {
        return search_in(reinterpret_cast<const uint8_t *>(haystack.data()), haystack.size());
    }
    // search_in for a device-resident haystack on a HIP stream (nullptr = default stream).
    bool search_in(DeviceSlice haystack, void *hip_stream = nullptr) const
    {
        int found = 0;
        check(ss_search_device(h_, haystack.ptr, haystack.len, hip_stream, &found));
        return found != 0;
    }
    template <class H>
    bool inlined_search_in(H &&haystack) const { return search_in(std::forward<H>(haystack)); }

    // Offset of the leftmost occurrence (the `Option<usize>` of tests/i386.rs:6-10):
    // returns npos when absent.
    static constexpr uint64_t npos = UINT64_MAX;
    uint64_t find(DeviceSlice haystack, void *hip_stream = nullptr) const
    {
        uint64_t pos = npos;
        check(ss_find_device(h_, haystack.ptr, haystack.len, hip_stream, &pos));
        return pos;
    }

    // ... and for a host slice (uploaded in chunks like search_in).
    uint64_t find(const uint8_t *haystack, size_t len) const
    {
        uint64_t pos = npos;
        check(ss_find_host(h_, haystack, len, &pos));
        return pos;
    }
    uint64_t find(const std::string &haystack) const
    {
        return find(reinterpret_cast<const uint8_t *>(haystack.data()), haystack.size());
    }
    // examples/grep.rs:42-56: open the file, one search_in (read -> upload -> scan pipeline).
    bool search_in_file(const std::string &path) const
    {
        int found = 0;
        check(ss_search_file(h_, path.c_str(), &found));
        return found != 0;
    }

    size_t position() const
    {
        size_t p = 0;
        check(ss_searcher_info(h_, nullptr, &p));
        return p;
    }
    size_t needle_len() const
    {
        size_t n = 0;
        check(ss_searcher_info(h_, &n, nullptr));
        return n;
    }
    ss_searcher *handle() const { return h_; }

    // The needle bytes the device filter tests (first, second, third; third == second: none).  `with_position`
    // keeps the caller's byte (second == position; first == 0 up to position 15); `new_` lets the library pick all
    // three (sliceslice_hip.h).
    struct Filter {
        size_t first, second, third;
    };
    Filter filter() const
    {
        Filter f{0, 0, 0};
        check(ss_searcher_filter3(h_, &f.first, &f.second, &f.third));
        return f;
    }
    void set_filter(size_t first, size_t second) { check(ss_searcher_set_filter3(h_, first, second, second)); }   // a plain two-byte filter
    void set_filter(size_t first, size_t second, size_t third) { check(ss_searcher_set_filter3(h_, first, second, third)); }

private:
    explicit DynamicHipSearcher(ss_searcher *h) : h_(h) {}
    ss_searcher *h_;
};

// Counterpart of `x86::Avx2Searcher` (src/x86.rs:266-382): needles of at least one byte.  An empty needle panics in the
// reference (`assert!(position < size)` with position = size.wrapping_sub(1), src/x86.rs:285-300; test
// `avx2_empty_needle`, src/x86.rs:545-549) where the dynamic searcher answers true; everything else is the dynamic
// searcher.
class HipSearcher : public DynamicHipSearcher {
public:
    static HipSearcher new_(const uint8_t *needle, size_t n)
    {
        if (n == 0) throw PositionPanic(SS_ERR_POSITION, "Avx2Searcher contract: the needle must not be empty");
        return HipSearcher(DynamicHipSearcher::new_(needle, n));
    }
    static HipSearcher new_(const std::string &needle) { return new_(reinterpret_cast<const uint8_t *>(needle.data()), needle.size()); }
    static HipSearcher with_position(const uint8_t *needle, size_t n, size_t position)
    {
        if (position >= n) throw PositionPanic(SS_ERR_POSITION, "position out of range (src/x86.rs:300)");
        return HipSearcher(DynamicHipSearcher::with_position(needle, n, position));
    }
    static HipSearcher with_position(const std::string &needle, size_t position)
    {
        return with_position(reinterpret_cast<const uint8_t *>(needle.data()), needle.size(), position);
    }

private:
    explicit HipSearcher(DynamicHipSearcher &&d) : DynamicHipSearcher(std::move(d)) {}
};

// Counterpart of `MemchrSearcher` (src/lib.rs:119-142): one byte; false for an empty haystack.
class MemchrHipSearcher {
public:
    static MemchrHipSearcher new_(uint8_t needle) { return MemchrHipSearcher(needle); }
    template <class H>
    bool search_in(H &&haystack) const { return inner_.search_in(std::forward<H>(haystack)); }
    bool search_in(const uint8_t *haystack, size_t len) const { return inner_.search_in(haystack, len); }
    template <class H>
    bool inlined_search_in(H &&haystack) const { return search_in(std::forward<H>(haystack)); }

private:
    explicit MemchrHipSearcher(uint8_t b) : inner_(DynamicHipSearcher::new_(&b, 1)) {}
    DynamicHipSearcher inner_;
};

// All GPUs of a node behind ONE search_in (ss_comm_init_all / ss_search_sharded_all): the haystack is range-partitioned
// into one shard per device (n-1 bytes of overlap: shard_range), each resident in its device's HBM; a search is one scan
// per device plus one grouped all-reduce(MAX) of the found flag.  What a drop-in for `search_in(&self, &[u8]) -> bool`
// (src/x86.rs:523) over several GPUs calls - no launcher, no rendezvous.
class NodeSearcher {
public:
    NodeSearcher(const uint8_t *needle, size_t n, int ndev, const int *devices = nullptr)
        : searcher_(DynamicHipSearcher::new_(needle, n)), n_(n), ndev_(ndev)
    {
        for (int g = 0; g < ndev; ++g) devs_.push_back(devices ? devices[g] : g);
        check(ss_comm_init_all(ndev, devices, &set_));
    }
    NodeSearcher(const std::string &needle, int ndev)
        : NodeSearcher(reinterpret_cast<const uint8_t *>(needle.data()), needle.size(), ndev) {}
    NodeSearcher(const NodeSearcher &) = delete;
    NodeSearcher &operator=(const NodeSearcher &) = delete;
    ~NodeSearcher() { ss_comm_set_free(set_); }

    int devices() const { return ndev_; }
    int device(int index) const { return devs_.at((size_t)index); }
    // byte range [begin, end) of shard `g` of a haystack of `len` bytes
    std::pair<size_t, size_t> shard_range(size_t len, int g) const
    {
        size_t b = 0, e = 0;
        check(ss_shard_range(len, n_, ndev_, g, &b, &e));
        return {b, e};
    }
    // SS_COMBINE_RCCL (default): one grouped ncclAllReduce; SS_COMBINE_HOST: the host ORs the pinned flag mirrors
    void set_combine(int mode) { check(ss_comm_set_combine(set_, mode)); }
    // SS_ISSUE_THREADS (default from two devices up): one issue thread per device; SS_ISSUE_SERIAL: the calling thread issues everything
    void set_issue(int mode) { check(ss_comm_set_issue(set_, mode)); }
    // ncclCommCount of every communicator of the set (they must agree)
    int rccl_ranks() const
    {
        int n = 0;
        check(ss_comm_set_count(set_, &n));
        return n;
    }

    // shards[g]: device pointer + length of shard g, resident on device(g)
    bool search_in(const DeviceSlice *shards) const
    {
        const void *ptrs[64];
        size_t lens[64];
        for (int g = 0; g < ndev_ && g < 64; ++g) { ptrs[g] = shards[g].ptr; lens[g] = shards[g].len; }
        int found = 0;
        check(ss_search_sharded_all(searcher_.handle(), ptrs, lens, set_, &found));
        return found != 0;
    }
    uint64_t find(const DeviceSlice *shards, const uint64_t *shard_begins) const
    {
        const void *ptrs[64];
        size_t lens[64];
        for (int g = 0; g < ndev_ && g < 64; ++g) { ptrs[g] = shards[g].ptr; lens[g] = shards[g].len; }
        uint64_t pos = DynamicHipSearcher::npos;
        check(ss_find_sharded_all(searcher_.handle(), ptrs, lens, shard_begins, set_, &pos));
        return pos;
    }

private:
    DynamicHipSearcher searcher_;
    size_t n_;
    int ndev_;
    std::vector<int> devs_;
    ss_comm_set *set_ = nullptr;
};

// Plan once, search many (ss_batch_plan_*): the reference's bench shape - build the searchers once (bench/benches/i386.rs:246-250),
// time the searches (:252-256) - for a whole batch of (needle, haystack) problems given as ranges of one haystack blob and one
// needle blob in device memory.  run() is the scan launch alone (plus one small kernel in plans with long haystacks) and produces
// the outputs itself - nothing allocated, capturable into a hipGraph; the
// problems' CONTENTS may change between runs, their ranges may not.  One run at a time per plan.
class BatchPlan {
public:
    // bool plan: run() writes `count` int flags (1 found, 0 absent, SS_BATCH_BAD_POSITION); find plan: `count` uint64 leftmost
    // offsets (SS_NPOS: absent).  `d_position` (bool plans): the with_position argument per problem, or nullptr for n - 1.
    BatchPlan(const void *d_haystacks, const uint64_t *d_hay_begin, const uint64_t *d_hay_end, const void *d_needles,
              const uint64_t *d_needle_begin, const uint64_t *d_needle_end, const uint64_t *d_position, size_t count, bool find,
              void *hip_stream = nullptr)
    {
        check(ss_batch_plan_create(d_haystacks, d_hay_begin, d_hay_end, d_needles, d_needle_begin, d_needle_end, d_position, count,
                                   find ? 1 : 0, hip_stream, &plan_));
    }
    BatchPlan(const BatchPlan &) = delete;
    BatchPlan &operator=(const BatchPlan &) = delete;
    ~BatchPlan() { ss_batch_plan_free(plan_); }

    void run(void *d_out, void *hip_stream = nullptr) const { check(ss_batch_plan_run(plan_, hip_stream, d_out)); }

private:
    ss_batch_plan *plan_ = nullptr;
};

}  // namespace hip
}  // namespace sliceslice
