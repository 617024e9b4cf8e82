// Can two virtual ranges share physical pages on this driver?  (what versions of the structure that share their unmoved parts would need: DESIGN.md, "what comes next")
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/vmm_alias tools/micro/vmm_alias.hip && /tmp/vmm_alias
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(unsigned *p, unsigned n, unsigned v) { unsigned i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + i; }
__global__ void k_sum(const unsigned *p, unsigned n, unsigned long long *out) { unsigned i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) atomicAdd(out, (unsigned long long)p[i]); }
int main() {
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    std::printf("allocation granularity %zu bytes\n", gran);
    const size_t chunk = gran < (2u << 20) ? (2u << 20) : gran, n_chunks = 4;
    hipMemGenericAllocationHandle_t shared, own_a, own_b;
    CK(hipMemCreate(&shared, chunk, &prop, 0)); CK(hipMemCreate(&own_a, chunk, &prop, 0)); CK(hipMemCreate(&own_b, chunk, &prop, 0));
    void *va = nullptr, *vb = nullptr;
    CK(hipMemAddressReserve(&va, chunk * n_chunks, 0, nullptr, 0)); CK(hipMemAddressReserve(&vb, chunk * n_chunks, 0, nullptr, 0));
    // version A: [shared][own_a]; version B: [shared][own_b]  (two chunks each mapped; the same physical chunk behind both first halves)
    CK(hipMemMap(va, chunk, 0, shared, 0)); CK(hipMemMap((char *)va + chunk, chunk, 0, own_a, 0));
    CK(hipMemMap(vb, chunk, 0, shared, 0)); CK(hipMemMap((char *)vb + chunk, chunk, 0, own_b, 0));
    hipMemAccessDesc acc{}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, 2 * chunk, &acc, 1)); CK(hipMemSetAccess(vb, 2 * chunk, &acc, 1));
    const unsigned n = (unsigned)(chunk / 4);
    unsigned long long *d_out; CK(hipMalloc(&d_out, 16)); CK(hipMemset(d_out, 0, 16));
    k_fill<<<(n + 255) / 256, 256>>>((unsigned *)va, n, 7u);                       // through A's mapping of the shared chunk
    k_fill<<<(n + 255) / 256, 256>>>((unsigned *)((char *)va + chunk), n, 100u);   // A's own
    k_fill<<<(n + 255) / 256, 256>>>((unsigned *)((char *)vb + chunk), n, 200u);   // B's own
    k_sum<<<(n + 255) / 256, 256>>>((const unsigned *)vb, n, d_out);               // the shared chunk read through B's mapping
    k_sum<<<(n + 255) / 256, 256>>>((const unsigned *)((char *)vb + chunk), n, d_out + 1);
    unsigned long long h[2]; CK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost));
    const unsigned long long tri = (unsigned long long)n * (n - 1) / 2;
    std::printf("shared chunk written through A, read through B: %s; B's own chunk: %s\n", h[0] == 7ull * n + tri ? "the same bytes" : "DIFFERENT", h[1] == 200ull * n + tri ? "its own" : "WRONG");
    CK(hipMemUnmap(va, 2 * chunk)); CK(hipMemUnmap(vb, 2 * chunk)); CK(hipMemAddressFree(va, chunk * n_chunks)); CK(hipMemAddressFree(vb, chunk * n_chunks));
    CK(hipMemRelease(shared)); CK(hipMemRelease(own_a)); CK(hipMemRelease(own_b));
    std::printf("ok\n");
    return 0;
}
