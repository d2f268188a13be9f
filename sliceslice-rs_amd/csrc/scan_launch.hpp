// scan_launch.hpp - host-side launcher of the scan kernel family, split out so that the template
// instantiations can be compiled as separate translation units (scan_inst_*.hip) in parallel.
#pragma once
#include "scan_kernels.hpp"

namespace ss {

// q = (position % 16) / 4, mode = 0/1/2 (see scan_tiles), `sink` = int flag or uint64 best (FIND).
// l8 = use the 8-bytes-per-lane first phase (mode 0 / one-byte needles, bool result only).
// Shape = the launch geometry: workgroups, threads per workgroup (128 / 256 / 512), tiles per workgroup
// (0 = grid-stride), and unused dynamic LDS per workgroup (caps the workgroups resident per CU; tuning only).
struct Shape {
    unsigned blocks;
    unsigned block;
    uint64_t tpb;
    uint32_t lds_pad;
};
template <int U, int NT, bool FIND>
void launch_scan_un(const Problem &pr, int q, int mode, bool one_byte, const Shape &sh, hipStream_t st, void *sink,
                    bool l8);

#ifdef SS_DEFINE_LAUNCH
template <int U, int NT, bool FIND>
void launch_scan_un(const Problem &pr, int q, int mode, bool one_byte, const Shape &sh, hipStream_t st, void *flag,
                    bool l8)
{
    const dim3 grid(sh.blocks), blk(sh.block);
    const uint64_t tpb = sh.tpb;
    const uint32_t dyn_lds = sh.lds_pad + (sh.block / kWave) * kNeedleLds;   // one needle slice per wave
    if constexpr (!FIND) {
        if (l8 && (one_byte || mode == 0)) {
            if (one_byte) {
                scan_kernel<0, 0, true, U, NT, false, true><<<grid, blk, dyn_lds, st>>>(pr, flag, tpb);
                return;
            }
            switch (q) {
            case 0: scan_kernel<0, 0, false, U, NT, false, true><<<grid, blk, dyn_lds, st>>>(pr, flag, tpb); break;
            case 1: scan_kernel<1, 0, false, U, NT, false, true><<<grid, blk, dyn_lds, st>>>(pr, flag, tpb); break;
            case 2: scan_kernel<2, 0, false, U, NT, false, true><<<grid, blk, dyn_lds, st>>>(pr, flag, tpb); break;
            default: scan_kernel<3, 0, false, U, NT, false, true><<<grid, blk, dyn_lds, st>>>(pr, flag, tpb); break;
            }
            return;
        }
    }
    if (one_byte) {
        scan_kernel<0, 0, true, U, NT, FIND><<<grid, blk, dyn_lds, st>>>(pr, flag, tpb);
        return;
    }
#define SS_CASE(QQ, MM)                                                                            \
    case (QQ) * 3 + (MM):                                                                          \
        scan_kernel<QQ, MM, false, U, NT, FIND><<<grid, blk, dyn_lds, st>>>(pr, flag, tpb);              \
        break;
    switch (q * 3 + mode) {
        SS_CASE(0, 0) SS_CASE(0, 1) SS_CASE(0, 2) SS_CASE(1, 0) SS_CASE(1, 1) SS_CASE(1, 2)
        SS_CASE(2, 0) SS_CASE(2, 1) SS_CASE(2, 2) SS_CASE(3, 0) SS_CASE(3, 1) SS_CASE(3, 2)
    }
#undef SS_CASE
}
#else
extern template void launch_scan_un<4, 0, false>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
extern template void launch_scan_un<4, 1, false>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
extern template void launch_scan_un<8, 0, false>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
extern template void launch_scan_un<8, 1, false>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
extern template void launch_scan_un<4, 0, true>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
extern template void launch_scan_un<4, 1, true>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
#endif

}  // namespace ss
