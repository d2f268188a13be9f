// scan_inst_find_nt0.hip - explicit instantiation of one slice of the scan kernel family (see scan_launch.hpp); the
// family is spread over six translation units so that they compile in parallel.
#define SS_DEFINE_LAUNCH 1
#include "scan_launch.hpp"

namespace ss {
template bool launch_scan_un<4, 0, true>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
}  // namespace ss
