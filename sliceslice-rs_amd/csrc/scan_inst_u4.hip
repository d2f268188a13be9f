// scan_inst_u4.hip - explicit instantiations of the scan kernel family (see scan_launch.hpp).
#define SS_DEFINE_LAUNCH 1
#include "scan_launch.hpp"

namespace ss {
template void launch_scan_un<4, 0, false>(const Problem &, int, int, bool, dim3, hipStream_t, void *, uint64_t, bool, uint32_t);
template void launch_scan_un<4, 1, false>(const Problem &, int, int, bool, dim3, hipStream_t, void *, uint64_t, bool, uint32_t);
}  // namespace ss
