// readbench.hip - standalone microbenchmark of streaming-read access patterns on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/readbench.hip -o /tmp/readbench && /tmp/readbench [GiB]
// Tuning aid (not product code): cache-policy bits, one vs two (shifted) streams, block sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename F>
static double time_ms_fwd(F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return best;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// AUX: cache policy for raw buffer loads on gfx94x/950: bit0 sc0, bit1 nt, bit4 sc1
template <int AUX>
__device__ __forceinline__ u32x4 bload(__amdgpu_buffer_rsrc_t rsrc, uint32_t off)
{
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, AUX);
}

// one wave = U consecutive KiB; block = 4 waves; tpb contiguous tiles per block. SHIFT>0: second stream at +SHIFT chunks
template <int U, int AUXA, int AUXB, int SHIFT>
__global__ void __launch_bounds__(256) k_read(const uint8_t* src, uint64_t nbytes, uint32_t* sink, uint64_t tpb)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t tile_bytes = 4ull * U * 1024;
    const uint64_t ntiles = nbytes / tile_bytes;
    uint64_t t0 = (uint64_t)blockIdx.x * tpb;
    const uint64_t t1 = t0 + tpb < ntiles ? t0 + tpb : ntiles;
    u32x4 acc = {0,0,0,0};
    for (; t0 < t1; ++t0) {
        const uint8_t* base = src + t0 * tile_bytes + (uint64_t)wave * U * 1024;   // wave-uniform
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(U * 1024 + 4096), 0x00020000);
        u32x4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = bload<AUXA>(rsrc, (uint32_t)(u * 1024 + lane * 16));
            if (SHIFT) b[u] = bload<AUXB>(rsrc, (uint32_t)(u * 1024 + lane * 16 + SHIFT * 16));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc ^= a[u]; if (SHIFT) acc ^= b[u]; }
    }
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x9E3779B9u) sink[0] = r;
}

__device__ __forceinline__ uint32_t zf(uint32_t x) { return (x - 0x01010101u) & ~x; }
__device__ __forceinline__ uint32_t shl1(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

// the same with scan-like VALU work per dword (two zero-byte flag computations, one DPP, one alignbyte, and/or):
// does the width effect survive next to the filter's instruction stream?
template <int W, int AUX>
__global__ void __launch_bounds__(256) k_filter_w(const uint8_t* src, uint64_t nbytes, uint32_t* sink, uint64_t tpb,
                                                  uint32_t n0, uint32_t nl, uint32_t r)
{
    constexpr int U = 4;
    constexpr int L = (U * 1024) / (64 * 4 * W);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t tile_bytes = 4ull * U * 1024;
    const uint64_t ntiles = nbytes / tile_bytes;
    uint64_t t0 = (uint64_t)blockIdx.x * tpb;
    const uint64_t t1 = t0 + tpb < ntiles ? t0 + tpb : ntiles;
    uint32_t hits = 0;
    for (; t0 < t1; ++t0) {
        const uint8_t* base = src + t0 * tile_bytes + (uint64_t)wave * U * 1024;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(U * 1024), 0x00020000);
        uint32_t v[L][W];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const uint32_t off = (uint32_t)(l * 64 * 4 * W + lane * 4 * W);
            if (W == 2) { auto t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, AUX); v[l][0] = t[0]; v[l][1 % W] = t[1]; }
            else { auto t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, AUX); v[l][0] = t[0]; v[l][1 % W] = t[1]; v[l][2 % W] = t[2]; v[l][3 % W] = t[3]; }
        }
        uint32_t any = 0;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            uint32_t w[W + 1];
#pragma unroll
            for (int k = 0; k < W; ++k) w[k] = zf(v[l][k] ^ nl);
            w[W] = shl1(w[0]);
#pragma unroll
            for (int k = 0; k < W; ++k) any |= zf(v[l][k] ^ n0) & __builtin_amdgcn_alignbyte(w[k + 1], w[k], r);
        }
        if (__ballot((any & 0x80808080u) != 0)) hits += 1;
    }
    if (hits == 0x9E3779B9u) sink[0] = hits;
}

// VALU/DPP sensitivity: 16 B/lane loads + NV plain VALU ops and ND wave_shl DPP moves per 16 bytes
template <int NV, int ND>
__global__ void __launch_bounds__(256) k_valu(const uint8_t* src, uint64_t nbytes, uint32_t* sink, uint64_t tpb, uint32_t k)
{
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t tile_bytes = 4ull * U * 1024;
    const uint64_t ntiles = nbytes / tile_bytes;
    uint64_t t0 = (uint64_t)blockIdx.x * tpb;
    const uint64_t t1 = t0 + tpb < ntiles ? t0 + tpb : ntiles;
    uint32_t acc = 0;
    for (; t0 < t1; ++t0) {
        const uint8_t* base = src + t0 * tile_bytes + (uint64_t)wave * U * 1024;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(U * 1024), 0x00020000);
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)(u * 1024 + lane * 16), 0, 2);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t x = v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
#pragma unroll
            for (int i = 0; i < NV; ++i) x = (x ^ k) + 0x01010101u * (i + 1);
#pragma unroll
            for (int i = 0; i < ND; ++i) x = shl1(x) ^ k;
            acc |= x;
        }
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}

template <int NV, int ND>
static void run_valu(const uint8_t* d, uint64_t nbytes, uint32_t* sink)
{
    const uint64_t blocks = nbytes / (16ull * 1024) / 64;
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        double ms = time_ms_fwd([&]() { k_valu<NV, ND><<<dim3((unsigned)blocks), dim3(256)>>>(d, nbytes, sink, 64, 0x5a5a5a5au); });
        if (ms < best) best = ms;
    }
    printf("16 B/lane nt loads + %2d VALU + %2d DPP per 16 B: %8.1f GB/s\n", 2 * NV + 3, ND, nbytes / best / 1e6);
    fflush(stdout);
}

// load width: W = 1, 2, 4 dwords per lane per load (uchar4 / 8 B / 16 B); the same bytes per wave-tile
template <int W, int AUX>
__global__ void __launch_bounds__(256) k_read_w(const uint8_t* src, uint64_t nbytes, uint32_t* sink, uint64_t tpb)
{
    constexpr int U = 4;                       // KiB per wave per tile
    constexpr int L = (U * 1024) / (64 * 4 * W);   // loads per lane per tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t tile_bytes = 4ull * U * 1024;
    const uint64_t ntiles = nbytes / tile_bytes;
    uint64_t t0 = (uint64_t)blockIdx.x * tpb;
    const uint64_t t1 = t0 + tpb < ntiles ? t0 + tpb : ntiles;
    uint32_t acc = 0;
    for (; t0 < t1; ++t0) {
        const uint8_t* base = src + t0 * tile_bytes + (uint64_t)wave * U * 1024;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(U * 1024), 0x00020000);
        uint32_t v[L][W];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const uint32_t off = (uint32_t)(l * 64 * 4 * W + lane * 4 * W);
            if (W == 1) v[l][0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, AUX);
            else if (W == 2) { auto t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, AUX); v[l][0] = t[0]; v[l][1 % W] = t[1]; }
            else { auto t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, AUX); v[l][0] = t[0]; v[l][1 % W] = t[1]; v[l][2 % W] = t[2]; v[l][3 % W] = t[3]; }
        }
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int w = 0; w < W; ++w) acc ^= v[l][w];
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}

// 16 contiguous bytes per lane fetched as TWO dwordx2 loads (lane stride 16 B, halves at +0 and +8)
template <int AUX>
__global__ void __launch_bounds__(256) k_read_split(const uint8_t* src, uint64_t nbytes, uint32_t* sink, uint64_t tpb)
{
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t tile_bytes = 4ull * U * 1024;
    const uint64_t ntiles = nbytes / tile_bytes;
    uint64_t t0 = (uint64_t)blockIdx.x * tpb;
    const uint64_t t1 = t0 + tpb < ntiles ? t0 + tpb : ntiles;
    uint32_t acc = 0;
    for (; t0 < t1; ++t0) {
        const uint8_t* base = src + t0 * tile_bytes + (uint64_t)wave * U * 1024;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(U * 1024), 0x00020000);
        uint32_t v[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            auto a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (uint32_t)(u * 1024 + lane * 16), 0, AUX);
            auto b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (uint32_t)(u * 1024 + lane * 16 + 8), 0, AUX);
            v[u][0] = a[0]; v[u][1] = a[1]; v[u][2] = b[0]; v[u][3] = b[1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}

// workgroup size study: WAVES waves per workgroup, each wave U KiB per tile, tpb tiles per workgroup
template <int WAVES, int U>
__global__ void __launch_bounds__(WAVES * 64) k_read_bs(const uint8_t* src, uint64_t nbytes, uint32_t* sink, uint64_t tpb)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t tile_bytes = (uint64_t)WAVES * U * 1024;
    const uint64_t ntiles = nbytes / tile_bytes;
    uint64_t t0 = (uint64_t)blockIdx.x * tpb;
    const uint64_t t1 = t0 + tpb < ntiles ? t0 + tpb : ntiles;
    u32x4 acc = {0, 0, 0, 0};
    for (; t0 < t1; ++t0) {
        const uint8_t* base = src + t0 * tile_bytes + (uint64_t)wave * U * 1024;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(U * 1024), 0x00020000);
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)(u * 1024 + lane * 16), 0, 2);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x9E3779B9u) sink[0] = r;
}

template <int WAVES, int U>
static void run_bs(const uint8_t* d, uint64_t nbytes, uint32_t* sink, uint64_t run_bytes);

// block -> address map variants: 0 = block b reads run b (linear); 1 = XCD-partitioned: blocks with the same
// b % 8 (observed: same XCD) read one contiguous eighth of the buffer; 2 = bit-reversed-ish scatter of runs
template <int MAP>
__global__ void __launch_bounds__(256) k_read_map(const uint8_t* src, uint64_t nbytes, uint32_t* sink, uint64_t tpb)
{
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t tile_bytes = 4ull * U * 1024;
    const uint64_t ntiles = nbytes / tile_bytes;
    const uint64_t nruns = ntiles / tpb;
    uint64_t run = blockIdx.x;
    if (MAP == 1) run = (blockIdx.x % 8) * (nruns / 8) + blockIdx.x / 8;
    if (MAP == 2) run = (blockIdx.x * 2654435761ull) % nruns;      // nruns is a power of two: odd multiplier = bijection
    if (run >= nruns) return;
    uint64_t t0 = run * tpb;
    const uint64_t t1 = t0 + tpb;
    u32x4 acc = {0, 0, 0, 0};
    for (; t0 < t1; ++t0) {
        const uint8_t* base = src + t0 * tile_bytes + (uint64_t)wave * U * 1024;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(U * 1024), 0x00020000);
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)(u * 1024 + lane * 16), 0, 2);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x9E3779B9u) sink[0] = r;
}

// 16 B/lane, but each load instruction is issued for HALF the wave (512 B per instruction)
template <int AUX>
__global__ void __launch_bounds__(256) k_read_half(const uint8_t* src, uint64_t nbytes, uint32_t* sink, uint64_t tpb)
{
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t tile_bytes = 4ull * U * 1024;
    const uint64_t ntiles = nbytes / tile_bytes;
    uint64_t t0 = (uint64_t)blockIdx.x * tpb;
    const uint64_t t1 = t0 + tpb < ntiles ? t0 + tpb : ntiles;
    u32x4 acc = {0, 0, 0, 0};
    for (; t0 < t1; ++t0) {
        const uint8_t* base = src + t0 * tile_bytes + (uint64_t)wave * U * 1024;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(U * 1024), 0x00020000);
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (lane < 32) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)(u * 1024 + lane * 16), 0, AUX);
            __builtin_amdgcn_sched_barrier(0);
            if (lane >= 32) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)(u * 1024 + lane * 16), 0, AUX);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x9E3779B9u) sink[0] = r;
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    std::vector<float> v;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.push_back(ms);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

template <int U, int AUXA, int AUXB, int SHIFT>
static void run(const char* name, const uint8_t* d, uint64_t nbytes, uint32_t* sink, uint64_t tpb)
{
    const uint64_t ntiles = nbytes / (4ull * U * 1024);
    const uint64_t blocks = (ntiles + tpb - 1) / tpb;
    double ms = time_ms([&]() { k_read<U, AUXA, AUXB, SHIFT><<<dim3((unsigned)blocks), dim3(256)>>>(d, nbytes, sink, tpb); }, 7);
    printf("%-44s U=%d tpb=%-4llu  %8.3f ms  %8.1f GB/s\n", name, U, (unsigned long long)tpb, ms, nbytes / ms / 1e6);
    fflush(stdout);
}

template <int W, int AUX>
static void run_w(const char* name, const uint8_t* d, uint64_t nbytes, uint32_t* sink, uint64_t tpb)
{
    const uint64_t ntiles = nbytes / (16ull * 1024);
    const uint64_t blocks = (ntiles + tpb - 1) / tpb;
    double ms = time_ms([&]() { k_read_w<W, AUX><<<dim3((unsigned)blocks), dim3(256)>>>(d, nbytes, sink, tpb); }, 7);
    printf("%-44s W=%d tpb=%-4llu  %8.3f ms  %8.1f GB/s\n", name, W, (unsigned long long)tpb, ms, nbytes / ms / 1e6);
    fflush(stdout);
}

template <int WAVES, int U>
static void run_bs(const uint8_t* d, uint64_t nbytes, uint32_t* sink, uint64_t run_bytes)
{
    const uint64_t tile_bytes = (uint64_t)WAVES * U * 1024;
    const uint64_t tpb = run_bytes / tile_bytes;
    const uint64_t blocks = (nbytes / tile_bytes + tpb - 1) / tpb;
    double ms = time_ms([&]() { k_read_bs<WAVES, U><<<dim3((unsigned)blocks), dim3(WAVES * 64)>>>(d, nbytes, sink, tpb); }, 7);
    printf("workgroup %4d threads, %d KiB/wave/tile, %4llu KiB runs: %8.1f GB/s\n", WAVES * 64, U,
           (unsigned long long)(run_bytes >> 10), nbytes / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const uint64_t nbytes = (uint64_t)(gib * (1ull << 30));
    uint8_t* d; uint32_t* sink;
    CK(hipMalloc((void**)&d, nbytes + (1 << 20))); CK(hipMalloc((void**)&sink, 64));
    CK(hipMemset(d, 0x5A, nbytes + (1 << 20)));
    run_valu<0, 0>(d, nbytes, sink);
    run_valu<4, 0>(d, nbytes, sink);
    run_valu<10, 0>(d, nbytes, sink);
    run_valu<20, 0>(d, nbytes, sink);
    run_valu<40, 0>(d, nbytes, sink);
    run_valu<4, 2>(d, nbytes, sink);
    run_valu<4, 4>(d, nbytes, sink);
    run_valu<4, 8>(d, nbytes, sink);
    run_valu<4, 16>(d, nbytes, sink);
    run_valu<0, 0>(d, nbytes, sink);
    for (uint64_t run : {256ull << 10, 1024ull << 10}) {
        run_bs<1, 4>(d, nbytes, sink, run);
        run_bs<2, 4>(d, nbytes, sink, run);
        run_bs<4, 4>(d, nbytes, sink, run);
        run_bs<8, 4>(d, nbytes, sink, run);
        run_bs<16, 4>(d, nbytes, sink, run);
        run_bs<4, 2>(d, nbytes, sink, run);
        run_bs<8, 2>(d, nbytes, sink, run);
        run_bs<4, 8>(d, nbytes, sink, run);
        run_bs<2, 8>(d, nbytes, sink, run);
    }
    for (int rep = 0; rep < 2; ++rep) {
        const uint64_t blocks = nbytes / (16ull * 1024) / 64;
        double m0 = time_ms([&]() { k_read_map<0><<<dim3((unsigned)blocks), dim3(256)>>>(d, nbytes, sink, 64); }, 7);
        double m1 = time_ms([&]() { k_read_map<1><<<dim3((unsigned)blocks), dim3(256)>>>(d, nbytes, sink, 64); }, 7);
        double m2 = time_ms([&]() { k_read_map<2><<<dim3((unsigned)blocks), dim3(256)>>>(d, nbytes, sink, 64); }, 7);
        printf("block->address map: linear %.1f GB/s | XCD-partitioned %.1f GB/s | scattered runs %.1f GB/s\n",
               nbytes / m0 / 1e6, nbytes / m1 / 1e6, nbytes / m2 / 1e6);
    }
    run_w<1, 2>("4 B/lane  (uchar4, dword) loads, nt", d, nbytes, sink, 64);
    run_w<2, 2>("8 B/lane  (dwordx2) loads, nt", d, nbytes, sink, 64);
    run_w<4, 2>("16 B/lane (dwordx4) loads, nt", d, nbytes, sink, 64);
    {
        const uint64_t blocks = (nbytes / (16ull * 1024) + 63) / 64;
        double ms = time_ms([&]() { k_read_split<2><<<dim3((unsigned)blocks), dim3(256)>>>(d, nbytes, sink, 64); }, 7);
        printf("%-44s      tpb=64    %8.3f ms  %8.1f GB/s\n", "16 B/lane as two strided dwordx2 loads, nt", ms, nbytes / ms / 1e6);
    }
    {
        const uint64_t blocks = (nbytes / (16ull * 1024) + 63) / 64;
        double ms = time_ms([&]() { k_read_half<2><<<dim3((unsigned)blocks), dim3(256)>>>(d, nbytes, sink, 64); }, 7);
        printf("%-44s      tpb=64    %8.3f ms  %8.1f GB/s\n", "16 B/lane, half-wave (512 B) load instructions, nt", ms, nbytes / ms / 1e6);
    }
    for (int rep = 0; rep < 2; ++rep) {
        const uint64_t blocks = (nbytes / (16ull * 1024) + 63) / 64;
        double ms2 = time_ms([&]() { k_filter_w<2, 2><<<dim3((unsigned)blocks), dim3(256)>>>(d, nbytes, sink, 64, 0x11111111u, 0x22222222u, 3); }, 7);
        printf("%-44s      tpb=64    %8.3f ms  %8.1f GB/s\n", "filter-like VALU + 8 B/lane loads, nt", ms2, nbytes / ms2 / 1e6);
        double ms4 = time_ms([&]() { k_filter_w<4, 2><<<dim3((unsigned)blocks), dim3(256)>>>(d, nbytes, sink, 64, 0x11111111u, 0x22222222u, 3); }, 7);
        printf("%-44s      tpb=64    %8.3f ms  %8.1f GB/s\n", "filter-like VALU + 16 B/lane loads, nt", ms4, nbytes / ms4 / 1e6);
    }
    run_w<1, 0>("4 B/lane  (uchar4, dword) loads, plain", d, nbytes, sink, 64);
    run_w<4, 0>("16 B/lane (dwordx4) loads, plain", d, nbytes, sink, 64);
    for (uint64_t tpb : {16ull, 64ull}) {
        run<4, 0, 0, 0>("one stream, plain", d, nbytes, sink, tpb);
        run<4, 1, 0, 0>("one stream, sc0", d, nbytes, sink, tpb);
        run<4, 2, 0, 0>("one stream, nt", d, nbytes, sink, tpb);
        run<4, 16, 0, 0>("one stream, sc1", d, nbytes, sink, tpb);
        run<4, 17, 0, 0>("one stream, sc0 sc1", d, nbytes, sink, tpb);
        run<4, 18, 0, 0>("one stream, sc1 nt", d, nbytes, sink, tpb);
        run<4, 19, 0, 0>("one stream, sc0 sc1 nt", d, nbytes, sink, tpb);
        run<4, 3, 0, 0>("one stream, sc0 nt", d, nbytes, sink, tpb);
        run<8, 2, 0, 0>("one stream, nt", d, nbytes, sink, tpb);
        run<2, 2, 0, 0>("one stream, nt", d, nbytes, sink, tpb);
        run<4, 0, 0, 1>("two streams (+1 chunk), plain/plain", d, nbytes, sink, tpb);
        run<4, 2, 0, 1>("two streams (+1 chunk), nt/plain", d, nbytes, sink, tpb);
        run<4, 0, 2, 1>("two streams (+1 chunk), plain/nt", d, nbytes, sink, tpb);
        run<4, 2, 2, 1>("two streams (+1 chunk), nt/nt", d, nbytes, sink, tpb);
        run<4, 2, 1, 1>("two streams (+1 chunk), nt/sc0", d, nbytes, sink, tpb);
        run<4, 3, 3, 1>("two streams (+1 chunk), sc0nt/sc0nt", d, nbytes, sink, tpb);
        run<4, 2, 2, 7>("two streams (+7 chunks), nt/nt", d, nbytes, sink, tpb);
        run<4, 0, 0, 7>("two streams (+7 chunks), plain/plain", d, nbytes, sink, tpb);
    }
    return 0;
}
