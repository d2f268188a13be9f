// dispatch_id_probe.hip - does a gfx950 kernel under this HIP runtime see a per-dispatch identity that is the same for every
// workgroup of a launch and different for consecutive launches - plain launches, launches on several streams, hipGraph replays?
// (What a batch plan's single-launch run needs: ss_batched.hip.)   hipcc --offload-arch=gfx950 -O2 dispatch_id_probe.hip -o probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

extern "C" __device__ unsigned long long ss_dispatch_id(void) __asm("llvm.amdgcn.dispatch.id");

struct Seen {
    unsigned long long id_min, id_max, qid_min, qid_max, qptr;
};

__global__ void probe(Seen *out)
{
    const unsigned long long id = ss_dispatch_id();
    const uint64_t *q = (const uint64_t *)__builtin_amdgcn_queue_ptr();
    const unsigned long long qid = q[4];            // hsa_queue_t::id (byte 32: type u32, features u32, base_address, doorbell_signal, size u32, reserved u32, id)
    if (threadIdx.x == 0) {
        atomicMin(&out->id_min, id);
        atomicMax(&out->id_max, id);
        atomicMin(&out->qid_min, qid);
        atomicMax(&out->qid_max, qid);
        if (blockIdx.x == 0) out->qptr = (unsigned long long)q;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    Seen *d = nullptr, h;
    CK(hipMalloc((void **)&d, sizeof(Seen) * 64));
    auto reset = [&](int k) {
        Seen s = {~0ull, 0, ~0ull, 0, 0};
        return hipMemcpy(d + k, &s, sizeof s, hipMemcpyHostToDevice);
    };
    hipStream_t st[3];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int k = 0;
    printf("{\"launches\": [\n");
    auto show = [&](const char *what, int slot) {
        (void)hipMemcpy(&h, d + slot, sizeof h, hipMemcpyDeviceToHost);
        printf("  {\"what\": \"%s\", \"dispatch_id_min\": %llu, \"dispatch_id_max\": %llu, \"queue_id_min\": %llu, \"queue_id_max\": %llu, \"queue_ptr\": \"0x%llx\"},\n",
               what, h.id_min, h.id_max, h.qid_min, h.qid_max, h.qptr);
    };
    for (int rep = 0; rep < 4; ++rep) {
        CK(reset(k));
        probe<<<4096, 64, 0, 0>>>(d + k);
        CK(hipDeviceSynchronize());
        show("null stream, 4096 workgroups", k++);
    }
    for (int rep = 0; rep < 2; ++rep)
        for (int s = 0; s < 3; ++s) {
            CK(reset(k));
            probe<<<2048, 64, 0, st[s]>>>(d + k);
            CK(hipStreamSynchronize(st[s]));
            char name[64];
            snprintf(name, sizeof name, "stream %d", s);
            show(name, k++);
        }
    // a captured graph of TWO launches, replayed three times
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(reset(k));
    CK(reset(k + 1));
    CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal));
    probe<<<1024, 64, 0, st[0]>>>(d + k);
    probe<<<1024, 64, 0, st[0]>>>(d + k + 1);
    CK(hipStreamEndCapture(st[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CK(reset(k));
        CK(reset(k + 1));
        CK(hipGraphLaunch(ge, st[0]));
        CK(hipStreamSynchronize(st[0]));
        show("graph replay, node 0", k);
        show("graph replay, node 1", k + 1);
    }
    printf("  {}\n]}\n");
    return 0;
}
