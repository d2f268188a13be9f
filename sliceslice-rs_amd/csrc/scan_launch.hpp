// scan_launch.hpp - host-side launcher of the scan kernel family, split out so that the template
// instantiations can be compiled as separate translation units (scan_inst_*.hip) in parallel.
#pragma once
#include "scan_kernels.hpp"

namespace ss {

// q = (position % 16) / 4, mode = 0/2 (see scan_tiles), `sink` = int flag or uint64 best (FIND).
// l8 = use the 8-bytes-per-lane first phase (mode 0 / one-byte needles, bool result only).
// Shape = the launch geometry: workgroups, threads per workgroup (128 / 256 / 512), tiles per workgroup
// (0 = grid-stride), and unused dynamic LDS per workgroup (caps the workgroups resident per CU; tuning only).
struct Shape {
    unsigned blocks;
    unsigned block;
    uint64_t tpb;
    uint32_t lds_pad;
};
// Which kernels a build holds.  The PRODUCT library has what the constructors and ss_searcher_set_filter3 can reach - U = 4,
// non-temporal loads, the single-stream (MODE 0) and cross-lane (MODE 2; MODE 3 = without the third byte, search only) kernels, the
// 8-byte first phase for one-byte needles only: 22 scan kernels (4 Q x 2 modes + one-byte, search and find; 4 Q of MODE 3).  -DSS_TUNING_VARIANTS adds every other combination
// ss_searcher_set_variant can name (U = 8, plain loads, the 8-byte phase for two-byte filters, the 16-byte layout for one-byte
// needles) - tuning residue: libsliceslice_hip_tuning.so (sliceslice_rs_amd._build.build_tuning), used by tools/ and by the
// variant tests.
constexpr bool kernel_built(int mode, bool one_byte, int U, int NT, bool FIND, bool L8)
{
#ifdef SS_TUNING_VARIANTS
    return (void)one_byte, (void)U, (void)NT, (void)L8, !(mode == 3 && FIND);
#else
    if (mode == 3 && FIND) return false;          // find() keeps the third byte (one kernel family less; the leftmost match is rarely far)
    if (U != 4 || NT != 1) return false;
    if (one_byte) return FIND ? !L8 : L8;
    return !L8;
#endif
}

// Returns false when the selected kernel is not part of this build (nothing has been launched then).
template <int U, int NT, bool FIND>
bool launch_scan_un(const Problem &pr, int q, int mode, bool one_byte, const Shape &sh, hipStream_t st, void *sink,
                    bool l8);

#ifdef SS_DEFINE_LAUNCH
template <int Q, int MODE, bool ONE_BYTE, int U, int NT, bool FIND, bool L8>
bool launch_one(const Problem &pr, const Shape &sh, hipStream_t st, void *flag)
{
    if constexpr (kernel_built(MODE, ONE_BYTE, U, NT, FIND, L8)) {
        const uint32_t dyn_lds = sh.lds_pad + (sh.block / kWave) * kNeedleLds;   // one needle slice per wave
        scan_kernel<Q, MODE, ONE_BYTE, U, NT, FIND, L8><<<dim3(sh.blocks), dim3(sh.block), dyn_lds, st>>>(pr, flag, sh.tpb);
        return true;
    } else {
        return (void)pr, (void)sh, (void)st, (void)flag, false;
    }
}

template <int U, int NT, bool FIND>
bool launch_scan_un(const Problem &pr, int q, int mode, bool one_byte, const Shape &sh, hipStream_t st, void *flag,
                    bool l8)
{
    if constexpr (!FIND) {
        if (l8 && (one_byte || mode == 0)) {
            if (one_byte) return launch_one<0, 0, true, U, NT, false, true>(pr, sh, st, flag);
            switch (q) {
            case 0: return launch_one<0, 0, false, U, NT, false, true>(pr, sh, st, flag);
            case 1: return launch_one<1, 0, false, U, NT, false, true>(pr, sh, st, flag);
            case 2: return launch_one<2, 0, false, U, NT, false, true>(pr, sh, st, flag);
            default: return launch_one<3, 0, false, U, NT, false, true>(pr, sh, st, flag);
            }
        }
    }
    if (one_byte) return launch_one<0, 0, true, U, NT, FIND, false>(pr, sh, st, flag);
#define SS_CASE(QQ, MM)                                                                            \
    case (QQ) * 4 + (MM):                                                                          \
        return launch_one<QQ, MM, false, U, NT, FIND, false>(pr, sh, st, flag);
    switch (q * 4 + mode) {
        SS_CASE(0, 0) SS_CASE(0, 2) SS_CASE(1, 0) SS_CASE(1, 2)
        SS_CASE(2, 0) SS_CASE(2, 2) SS_CASE(3, 0) SS_CASE(3, 2)
        SS_CASE(0, 3) SS_CASE(1, 3) SS_CASE(2, 3) SS_CASE(3, 3)
    }
#undef SS_CASE
    return false;
}
#else
extern template bool launch_scan_un<4, 1, false>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
extern template bool launch_scan_un<4, 1, true>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
#ifdef SS_TUNING_VARIANTS
extern template bool launch_scan_un<4, 0, false>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
extern template bool launch_scan_un<4, 0, true>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
extern template bool launch_scan_un<8, 0, false>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
extern template bool launch_scan_un<8, 1, false>(const Problem &, int, int, bool, const Shape &, hipStream_t, void *, bool);
#endif
#endif

}  // namespace ss
