// tools/micro/valu_rate.hip -- wave64 VALU issue rate of one SIMD, measured: independent and dependent chains of v_fma_f32,
// v_pk_fma_f32, v_cndmask, v_add_u32 and v_rcp_f32 at 1..8 waves per SIMD.   hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int KIND>
__global__ void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {        // 8 independent chains of v_fma_f32: 64 instructions per REP
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        }
        else if (KIND == 1) {   // one dependent chain
            REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));)
        }
        else if (KIND == 2) {   // transcendental
            REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        }
        else if (KIND == 3) {   // integer add
            REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));)
        }
        else if (KIND == 4) {   // 64-bit multiply-add of the RNG
            unsigned long long t;
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0\n v_mad_u64_u32 %0, vcc, %3, %2, 0\n v_mad_u64_u32 %0, vcc, %4, %2, 0\n v_mad_u64_u32 %0, vcc, %5, %2, 0" : "=&v"(t) : "v"(a0), "v"(m), "v"(a1), "v"(a2), "v"(a3) : "vcc"); a0 += (float)(unsigned)t;)
        }
        else if (KIND == 5) {   // packed fma, 4 independent pairs
            typedef float v2 __attribute__((ext_vector_type(2)));
            v2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, mm = { m, m }, cc = { c, c };
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));)
            a0 = p0.x + p0.y; a2 = p1.x + p1.y; a4 = p2.x + p2.y; a6 = p3.x + p3.y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
void run(const char* name, int instrPerIter) {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int wavesPerSimd : { 1, 2, 4, 8 }) {
        const int threads = 64 * 4 * wavesPerSimd > 1024 ? 1024 : 64 * 4 * wavesPerSimd;       // waves per CU = 4 * wavesPerSimd
        const int blocksPerCu = (64 * 4 * wavesPerSimd) / threads;
        const int grid = 256 * blocksPerCu;
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, d, 10, 1.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, d, iters, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instrPerSimd = (double)iters * instrPerIter * wavesPerSimd;
        std::printf("%-28s %d waves/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, wavesPerSimd, ms, ms * 1e6 / instrPerSimd, ms * 1e6 / instrPerSimd * 2.4);
    }
    hipFree(d);
}

int main() {
    run<0>("v_fma_f32 x8 independent", 64);
    run<1>("v_fma_f32 dependent chain", 64);
    run<2>("v_rcp_f32 x4 independent", 64);
    run<3>("v_add_u32 x4 independent", 64);
    run<4>("v_mad_u64_u32", 64);
    run<5>("v_pk_fma_f32 x4 independent", 64);
    return 0;
}
