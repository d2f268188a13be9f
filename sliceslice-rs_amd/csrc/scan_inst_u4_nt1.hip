// scan_inst_u4_nt1.hip - explicit instantiation of one slice of the scan kernel family (see scan_launch.hpp); the
// family is spread over six translation units so that they compile in parallel.
#define SS_DEFINE_LAUNCH 1
#include "scan_launch.hpp"
#include <cstring>

namespace ss {
template bool launch_scan_un<4, 1, false>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
}  // namespace ss

#ifdef SS_CAND_PROF   // instrumented A/B builds only (tools/cand_prof.py)
extern "C" __attribute__((visibility("default"))) int ss_debug_cand_prof(unsigned long long out[8], int reset)
{
    static unsigned long long h[ss::kCandProfSlots][8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(ss::g_cand_prof), sizeof h) != hipSuccess) return -1;
    for (int k = 0; k < 8; ++k) out[k] = 0;
    for (int i = 0; i < ss::kCandProfSlots; ++i)
        for (int k = 0; k < 8; ++k) out[k] += h[i][k];
    if (reset) {
        memset(h, 0, sizeof h);
        if (hipMemcpyToSymbol(HIP_SYMBOL(ss::g_cand_prof), h, sizeof h) != hipSuccess) return -1;
    }
    return 0;
}
#endif
