// Issue rate of a few vector instructions on gfx950, eight waves a SIMD (every CU full): cycles of the SIMD per wave-instruction, from s_memtime around a long unrolled run.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/micro/valu_rate.hip && /tmp/valu_rate
// (what profiles/README.md round 4c leans on: v_fma_mix_f32 with a half-precision operand from an SGPR against v_fma_f32)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
constexpr int kIters = 8192;   // long enough that the dispatch ramp (tens of microseconds) does not count
template <int KIND> __global__ __launch_bounds__(256) void k_rate(unsigned long long *out, float seed, unsigned su) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f, c = 0.5f;
    unsigned s = su;   // wave-uniform: lives in an SGPR
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, pb = {b, b}, pc = {c, c};
    unsigned long long sp = ((unsigned long long)su << 32) | (su ^ 0x00010000u);   // wave-uniform: an SGPR pair
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; it++) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (KIND == 1) { REP16(asm volatile("v_fma_mix_f32 %0, %6, %4, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %6, %4, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %6, %4, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %6, %4, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "s"(s));) }
        if (KIND == 2) { REP16(asm volatile("v_fma_f32 %0, %6, %4, %5\n v_fma_f32 %1, %6, %4, %5\n v_fma_f32 %2, %6, %4, %5\n v_fma_f32 %3, %6, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "s"(s));) }
        if (KIND == 3) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(p0), "+v"(p1) : "v"(pb), "v"(pc));) }
        if (KIND == 6) { REP16(asm volatile("v_pk_fma_f32 %0, %4, %0, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f32 %1, %4, %1, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]\n v_pk_fma_f32 %0, %4, %0, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %4, %1, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "+v"(p0), "+v"(p1) : "v"(pb), "v"(pc), "s"(sp));) }
        if (KIND == 4) { REP16(asm volatile("v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16 %1, %4\n v_cvt_f32_f16 %2, %4\n v_cvt_f32_f16 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s));) }
        if (KIND == 5) { REP16(asm volatile("v_max3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_max_f32 %2, %2, %4\n v_min_f32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if (a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y == 12345.678f) out[0] = 0;
}
template <int KIND> static void run(const char *name) {
    const int blocks = 256 * 8, threads = 256;   // 8 waves a SIMD on every CU
    unsigned long long *d; hipMalloc(&d, blocks * 4 * 8);
    k_rate<KIND><<<blocks, threads>>>(d, 1.0f, 0x3C003C00u); hipDeviceSynchronize();
    k_rate<KIND><<<blocks, threads>>>(d, 1.0f, 0x3C003C00u); hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4); hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    // a wave issued 256 * 16 * 4 instructions; eight waves share the SIMD: SIMD cycles per wave-instruction = elapsed / (instructions * 8)
    std::printf("%-46s %.2f shader cycles of the SIMD per wave-instruction (a wave's %d instructions took %.0f cycles among 8 waves)\n", name, mean / ((double)kIters * 64 * 8), kIters * 64, mean);
    hipFree(d);
}
int main() {
    run<0>("v_fma_f32 (three VGPR operands)"); run<2>("v_fma_f32 (one SGPR operand)"); run<1>("v_fma_mix_f32 (half operand from an SGPR)");
    run<3>("v_pk_fma_f32 (two products a lane)"); run<6>("v_pk_fma_f32 (SGPR pair operand, halves selected)"); run<4>("v_cvt_f32_f16 (SGPR operand)"); run<5>("v_max3 / v_min3 / v_max / v_min");
    return 0;
}
