// Developer tool: per-instruction VALU issue cost on gfx950 for the ops of the int4 GEMV inner loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    float f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0, f6 = 0, f7 = 0;
    uint32_t m = 0x00F000F0u ^ (seed & 1), g = 0x64006400u;
    asm volatile("" : "+s"(m));
    asm volatile("" : "+v"(g));
    for (int i = 0; i < iters; ++i) {
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
        if (OP == 0) {  // v_dot2c_f32_f16, 8 independent accumulators
#define D(n) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(f##n) : "v"(a##n), "v"(a0));
            REP8(D) REP8(D) REP8(D) REP8(D)
#undef D
        } else if (OP == 1) {  // v_and_or_b32
#define D(n) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a##n) : "s"(m), "v"(g));
            REP8(D) REP8(D) REP8(D) REP8(D)
#undef D
        } else if (OP == 2) {  // v_pk_mul_f16
#define D(n) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a##n) : "v"(g));
            REP8(D) REP8(D) REP8(D) REP8(D)
#undef D
        } else if (OP == 3) {  // v_lshrrev_b32
#define D(n) asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(a##n));
            REP8(D) REP8(D) REP8(D) REP8(D)
#undef D
        } else if (OP == 4) {  // v_fma_f32
#define D(n) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f##n) : "v"(f0), "v"(f1));
            REP8(D) REP8(D) REP8(D) REP8(D)
#undef D
        } else if (OP == 5) {  // v_dot2c, ONE accumulator (dependent chain)
#define D(n) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(f0) : "v"(a##n), "v"(a1));
            REP8(D) REP8(D) REP8(D) REP8(D)
#undef D
        } else if (OP == 6) {  // v_pk_fma_f16
#define D(n) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(a##n) : "v"(g));
            REP8(D) REP8(D) REP8(D) REP8(D)
#undef D
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (float)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}

template <int OP>
int run(const char* name, int waves_per_simd) {
    float* out; CK(hipMalloc(&out, 256 * 2048 * 4));
    const int iters = 2000, blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<OP><<<blocks, 256>>>(out, 10, 1); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k<OP><<<blocks, 256>>>(out, iters, 1); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double inst_per_wave = (double)iters * 32;
    const double ns_per_inst = ms * 1e6 / (inst_per_wave * waves_per_simd);
    printf("%-34s waves/SIMD=%d  %.3f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ns_per_inst, ns_per_inst * 2.4);
    return 0;
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_dot2c_f32_f16 (8 accumulators)", w);
        run<5>("v_dot2c_f32_f16 (1 accumulator)", w);
        run<1>("v_and_or_b32", w);
        run<2>("v_pk_mul_f16", w);
        run<6>("v_pk_fma_f16", w);
        run<3>("v_lshrrev_b32", w);
        run<4>("v_fma_f32", w);
    }
    return 0;
}
