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

int main(int argc, char** argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const uint64_t nbytes = (uint64_t)(gib * (1ull << 30));
    uint8_t* d; uint32_t* sink;
    CK(hipMalloc((void**)&d, nbytes + (1 << 20))); CK(hipMalloc((void**)&sink, 64));
    CK(hipMemset(d, 0x5A, nbytes + (1 << 20)));
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
