// v_mqsad_pk_u16_u8 on gfx950, checked against a software model: four masked sums of absolute differences (an 8-byte window
// against a 4-byte reference at byte offsets 0..3), reference bytes that are ZERO left out.  Measured beside it (round 3): it
// issues at ~1/3.5 of the plain VALU rate.  A tile pre-filter built on it (four needle bytes compared at sixteen offsets in
// four instructions, the byte-wise three-byte filter only for tiles it flags) was tried and rejected:
// profiles/r03/ab_presad_mqsad_rejected.jsonl.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mqsad_probe tools/mqsad_probe.hip && /tmp/mqsad_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__global__ void k(const uint64_t *s0, const uint32_t *s1, uint64_t *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_mqsad_pk_u16_u8(s0[i], s1[i], 0ull);
}
static uint64_t model(uint64_t a, uint32_t r, int mask_on)   // mask_on: 1 = ref byte zero is skipped, 2 = source byte zero skipped, 0 = none
{
    uint64_t d = 0;
    for (int i = 0; i < 4; ++i) {
        unsigned sum = 0;
        for (int j = 0; j < 4; ++j) {
            const int sb = (a >> (8 * (i + j))) & 0xFF, rb = (r >> (8 * j)) & 0xFF;
            if (mask_on == 1 && rb == 0) continue;
            if (mask_on == 2 && sb == 0) continue;
            sum += (unsigned)abs(sb - rb);
        }
        d |= (uint64_t)(sum & 0xFFFF) << (16 * i);
    }
    return d;
}
int main()
{
    const int n = 4096;
    uint64_t *hs0 = new uint64_t[n], *ho = new uint64_t[n];
    uint32_t *hs1 = new uint32_t[n];
    srand(7);
    for (int i = 0; i < n; ++i) {
        uint64_t a = 0; uint32_t r = 0;
        for (int k = 0; k < 8; ++k) a |= (uint64_t)((rand() % 4 == 0) ? 0 : (rand() & 0xFF)) << (8 * k);
        for (int k = 0; k < 4; ++k) r |= (uint32_t)((rand() % 3 == 0) ? 0 : (rand() & 0xFF)) << (8 * k);
        if (i % 5 == 0) { r = (uint32_t)(a >> 8); }           // exact match at offset 1
        hs0[i] = a; hs1[i] = r;
    }
    uint64_t *d0, *dout; uint32_t *d1;
    hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 4); hipMalloc(&dout, n * 8);
    hipMemcpy(d0, hs0, n * 8, hipMemcpyHostToDevice); hipMemcpy(d1, hs1, n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d0, d1, dout, n);
    hipMemcpy(ho, dout, n * 8, hipMemcpyDeviceToHost);
    int bad[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int m = 0; m < 3; ++m) bad[m] += ho[i] != model(hs0[i], hs1[i], m);
    printf("{\"mismatches_no_mask\": %d, \"mismatches_ref_byte_zero_skipped\": %d, \"mismatches_source_byte_zero_skipped\": %d, \"example\": \"%016llx %08x -> %016llx\"}\n",
           bad[0], bad[1], bad[2], (unsigned long long)hs0[1], hs1[1], (unsigned long long)ho[1]);
    return 0;
}
