// scan_inst_u8_nt0.hip - explicit instantiation of one slice of the scan kernel family (see scan_launch.hpp); the
// family is spread over translation units so that they compile in parallel.  U = 8: part of the TUNING build only
// (-DSS_TUNING_VARIANTS, libsliceslice_hip_tuning.so).
#define SS_DEFINE_LAUNCH 1
#include "scan_launch.hpp"

namespace ss {
template bool launch_scan_un<8, 0, false>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
}  // namespace ss
