// sliceslice_hip_service.hpp - header-only C++ veneer of the resident search service (include/sliceslice_hip_service.h): an opt-in
// component outside the hot path.  A program that includes this links libsliceslice_hip_service.so INSTEAD of libsliceslice_hip.so
// (the service library holds every function of the drop-in library as well).
#pragma once
#include "sliceslice_hip.hpp"
#include "sliceslice_hip_service.h"

namespace sliceslice {
namespace hip {

// A resident search service on the current device (ss_service_*): a kernel that stays on the GPU and answers one search_in at a
// time without a launch (5 us per search instead of 8.5-9.5).  The shape of the reference's own bench loop
// (bench/benches/i386.rs:246-256): build the searchers FIRST (building allocates, and allocation waits for the service's lease
// to run out), bind() the text if it does not change between searches, then one search_in per needle.
class SearchService {
public:
    explicit SearchService(int workgroups = 0, double lease_ms = 0.0) { check(ss_service_start(workgroups, lease_ms, &sv_)); }
    SearchService(const SearchService &) = delete;
    SearchService &operator=(const SearchService &) = delete;
    ~SearchService() { ss_service_stop(sv_); }

    // the semantics of DynamicHipSearcher::search_in(DeviceSlice) for a haystack that is COMPLETE in device memory
    bool search_in(const DynamicHipSearcher &s, DeviceSlice haystack) const
    {
        int found = 0;
        check(ss_service_search(sv_, s.handle(), haystack.ptr, haystack.len, &found));
        return found != 0;
    }
    // the caller vouches that `haystack` stays unchanged until unbind() / the next bind()
    void bind(DeviceSlice haystack) { check(ss_service_bind(sv_, haystack.ptr, haystack.len)); }
    void unbind() { check(ss_service_bind(sv_, nullptr, 0)); }

private:
    ss_service *sv_ = nullptr;
};

}  // namespace hip
}  // namespace sliceslice
