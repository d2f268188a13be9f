// scan_inst_u4.hip - explicit instantiations of the scan kernel family (see scan_launch.hpp).
#define SS_DEFINE_LAUNCH 1
#include "scan_launch.hpp"

namespace ss {
template void launch_scan_un<4, 0, false>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
template void launch_scan_un<4, 1, false>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
}  // namespace ss
