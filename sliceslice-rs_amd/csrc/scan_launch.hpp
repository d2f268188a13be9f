// scan_launch.hpp - host-side launcher of the scan kernel family, split out so that the template
// instantiations can be compiled as separate translation units (scan_inst_*.hip) in parallel.
#pragma once
#include "scan_kernels.hpp"

namespace ss {

// q = (position % 16) / 4, mode = 0/1/2 (see scan_tiles), `sink` = int flag or uint64 best (FIND).
template <int U, int NT, bool FIND>
void launch_scan_un(const Problem &pr, int q, int mode, bool one_byte, dim3 grid, hipStream_t st, void *sink,
                    uint64_t tpb);

#ifdef SS_DEFINE_LAUNCH
template <int U, int NT, bool FIND>
void launch_scan_un(const Problem &pr, int q, int mode, bool one_byte, dim3 grid, hipStream_t st, void *flag,
                    uint64_t tpb)
{
    dim3 blk(kBlock);
    if (one_byte) {
        scan_kernel<0, 0, true, U, NT, FIND><<<grid, blk, 0, st>>>(pr, flag, tpb);
        return;
    }
#define SS_CASE(QQ, MM)                                                                            \
    case (QQ) * 3 + (MM):                                                                          \
        scan_kernel<QQ, MM, false, U, NT, FIND><<<grid, blk, 0, st>>>(pr, flag, tpb);              \
        break;
    switch (q * 3 + mode) {
        SS_CASE(0, 0) SS_CASE(0, 1) SS_CASE(0, 2) SS_CASE(1, 0) SS_CASE(1, 1) SS_CASE(1, 2)
        SS_CASE(2, 0) SS_CASE(2, 1) SS_CASE(2, 2) SS_CASE(3, 0) SS_CASE(3, 1) SS_CASE(3, 2)
    }
#undef SS_CASE
}
#else
extern template void launch_scan_un<4, 0, false>(const Problem &, int, int, bool, dim3, hipStream_t, void *, uint64_t);
extern template void launch_scan_un<4, 1, false>(const Problem &, int, int, bool, dim3, hipStream_t, void *, uint64_t);
extern template void launch_scan_un<8, 0, false>(const Problem &, int, int, bool, dim3, hipStream_t, void *, uint64_t);
extern template void launch_scan_un<8, 1, false>(const Problem &, int, int, bool, dim3, hipStream_t, void *, uint64_t);
extern template void launch_scan_un<4, 0, true>(const Problem &, int, int, bool, dim3, hipStream_t, void *, uint64_t);
extern template void launch_scan_un<4, 1, true>(const Problem &, int, int, bool, dim3, hipStream_t, void *, uint64_t);
#endif

}  // namespace ss
