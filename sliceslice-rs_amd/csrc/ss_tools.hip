// ss_tools.hip - libsliceslice_hip_tools.so: what the benchmark and the tests need AROUND the searcher and a product that
// replaces DynamicAvx2Searcher does not - the synthetic haystack generator of SURVEY.md 8d (device and host, bit-identical),
// the plain streaming read that bench.py prints next to the scan's GB/s, and the self-test of the cross-lane primitives.
// Declared in include/sliceslice_hip_tuning.h; independent of libsliceslice_hip.so (nothing here touches a searcher).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/sliceslice_hip_tuning.h"
#define SS_AUX_TOOLS 1
#include "aux_kernels.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorNoDevice ? SS_ERR_NO_DEVICE : SS_ERR_HIP, "%s: %s (%s:%d)",  \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

int compute_units(int *cus)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(cus, hipDeviceAttributeMultiprocessorCount, dev));
    return SS_OK;
}

}  // namespace

extern "C" {

const char *ss_tools_last_error(void) { return g_err; }

int ss_fill_random_device(void *d_dst, uint64_t global_offset, size_t len, uint64_t seed, void *hip_stream)
{
    if (len == 0) return SS_OK;
    if (!d_dst) return fail(SS_ERR_ARGUMENT, "dst is NULL");
    int cus = 0;
    if (int rc = compute_units(&cus)) return rc;
    uint64_t words = (len + 15) / 8;
    uint64_t blocks = (words + ss::kBlock - 1) / ss::kBlock;
    if (blocks > (uint64_t)cus * 16) blocks = (uint64_t)cus * 16;
    ss::fill_random_kernel<<<dim3((unsigned)blocks), dim3(ss::kBlock), 0, static_cast<hipStream_t>(hip_stream)>>>(
        static_cast<uint8_t *>(d_dst), global_offset, len, seed);
    HIP_TRY(hipGetLastError());
    return SS_OK;
}

int ss_fill_random_host(uint8_t *dst, uint64_t global_offset, size_t len, uint64_t seed)
{
    if (len && !dst) return fail(SS_ERR_ARGUMENT, "dst is NULL");
    size_t k = 0;
    while (k < len) {
        const uint64_t i = global_offset + k;
        uint64_t v = ss::synth_word(seed, i >> 3) >> (8 * (i & 7));
        size_t take = 8 - (size_t)(i & 7);
        if (take > len - k) take = len - k;
        for (size_t j = 0; j < take; ++j, v >>= 8) dst[k + j] = (uint8_t)v;
        k += take;
    }
    return SS_OK;
}

int ss_read_ceiling(const void *d_src, size_t len, void *hip_stream, int reps, float *ms_per_rep)
{
    if (!d_src || !ms_per_rep || reps < 1) return fail(SS_ERR_ARGUMENT, "bad argument");
    if (((uintptr_t)d_src & 15) != 0) return fail(SS_ERR_ARGUMENT, "source must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    uint32_t *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    constexpr int U = 4;
    // A handful of launch shapes (bytes per lane, tiles per workgroup, workgroups per CU through unused LDS);
    // the fastest is reported.  The 16-byte / two-tile shape is the scan's own.
    struct Shape { int lane_bytes; uint64_t tpb; uint32_t lds; };
    const Shape shapes[] = {{16, 2, 0}, {16, 2, 32 << 10}, {16, 1, 32 << 10}, {8, 2, 0}, {8, 2, 32 << 10}, {8, 1, 32 << 10}};
    auto launch = [&](const Shape &sh) {
        const uint64_t ntiles = (len / 1024) / (ss::kWavesPerBlock * U);
        uint64_t blocks = (ntiles + sh.tpb - 1) / sh.tpb;
        if (blocks < 1) blocks = 1;
        const dim3 grid((unsigned)blocks);
        if (sh.lane_bytes == 16)
            ss::read_ceiling_kernel<U, ss::u32x4><<<grid, dim3(ss::kBlock), sh.lds, st>>>(static_cast<const ss::u32x4 *>(d_src), len / 16, sink, sh.tpb);
        else
            ss::read_ceiling_kernel<U, ss::u32x2><<<grid, dim3(ss::kBlock), sh.lds, st>>>(static_cast<const ss::u32x2 *>(d_src), len / 8, sink, sh.tpb);
    };
    auto run = [&]() -> hipError_t {
        hipError_t e;
        if ((e = hipMalloc((void **)&sink, 64)) != hipSuccess) return e;
        if ((e = hipEventCreate(&e0)) != hipSuccess) return e;
        if ((e = hipEventCreate(&e1)) != hipSuccess) return e;
        float best = 0;
        for (const Shape &sh : shapes) {
            launch(sh);                                             // warm-up
            if ((e = hipEventRecord(e0, st)) != hipSuccess) return e;
            for (int r = 0; r < reps; ++r) launch(sh);
            if ((e = hipEventRecord(e1, st)) != hipSuccess) return e;
            if ((e = hipEventSynchronize(e1)) != hipSuccess) return e;
            if ((e = hipGetLastError()) != hipSuccess) return e;
            float t = 0;
            if ((e = hipEventElapsedTime(&t, e0, e1)) != hipSuccess) return e;
            if (best == 0 || t < best) best = t;
        }
        ms = best;
        return hipSuccess;
    };
    const hipError_t e = run();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return fail(SS_ERR_HIP, "read ceiling: %s", hipGetErrorString(e));
    *ms_per_rep = ms / (float)reps;
    return SS_OK;
}

// DPP / alignbyte self-test used by the GPU tests: out must hold 320 uint32 (host memory).
int ss_selftest_dpp(uint32_t *out)
{
    if (!out) return fail(SS_ERR_ARGUMENT, "out is NULL");
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 320 * sizeof(uint32_t)));
    ss::dpp_probe_kernel<<<1, 64>>>(d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d, 320 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return SS_OK;
}

}  // extern "C"
