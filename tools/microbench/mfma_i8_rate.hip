// Issue rate of the int8 / f16 MFMA shapes on gfx950: one or two waves per SIMD, 4 independent accumulators, shader cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w mfma_i8_rate.hip -o mfma_i8_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void k(unsigned long long* out, int iters) {
    i32x4 a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)threadIdx.x, 8};
    i32x16 c32[4] = {};
    i32x4 c16[4] = {};
    f32x16 cf[4] = {};
    const f16x8 ah = __builtin_bit_cast(f16x8, a), bh = __builtin_bit_cast(f16x8, b);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KIND == 0) c32[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c32[u], 0, 0, 0);
            if (KIND == 1) c16[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c16[u], 0, 0, 0);
            if (KIND == 2) cf[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, cf[u], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int x = 0;
    for (int u = 0; u < 4; ++u) x ^= c32[u][0] ^ c16[u][0] ^ (int)cf[u][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}

int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    unsigned long long h[2];
    const char* names[3] = {"v_mfma_i32_32x32x32_i8", "v_mfma_i32_16x16x64_i8", "v_mfma_f32_32x32x16_f16"};
    for (int waves = 4; waves <= 8; waves += 4)
        for (int kind = 0; kind < 3; ++kind) {
            const int iters = 4096;
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) k<0><<<256, waves * 64>>>(d, iters);
                if (kind == 1) k<1><<<256, waves * 64>>>(d, iters);
                if (kind == 2) k<2><<<256, waves * 64>>>(d, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            // sustained: a 100 x longer run under HIP events -> wall-clock rate and the clock the chip holds on this instruction alone
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            const int long_iters = iters * 100;
            hipEventRecord(e0);
            if (kind == 0) k<0><<<256, waves * 64>>>(d, long_iters);
            if (kind == 1) k<1><<<256, waves * 64>>>(d, long_iters);
            if (kind == 2) k<2><<<256, waves * 64>>>(d, long_iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h2[2]; hipMemcpy(h2, d, 16, hipMemcpyDeviceToHost);
            const double macs_total = (kind == 1 ? 16.0 * 16 * 64 : kind == 0 ? 32.0 * 32 * 32 : 32.0 * 32 * 16) * 4.0 * long_iters * waves * 256;
            printf("    sustained %.2f ms: %.2f P(FL)OP/s wall, clock %.2f GHz\n", ms, macs_total * 2 / (ms * 1e-3) / 1e15, (double)h2[0] / (ms * 1e-3) / 1e9);
            const double per_wave = (double)h[0] / (iters * 4.0), per_simd = per_wave / (waves / 4);
            const double macs = kind == 1 ? 16.0 * 16 * 64 : kind == 0 ? 32.0 * 32 * 32 : 32.0 * 32 * 16;
            printf("%-26s %d wave(s) per SIMD: %6.1f cycles per MFMA per wave = %5.1f per SIMD issue -> %6.0f MAC/cycle/SIMD -> %5.2f P(FL)OP/s at 2.4 GHz\n",
                   names[kind], waves / 4, per_wave, per_simd, macs / per_simd, macs / per_simd * 2 * 4 * 256 * 2.4e9 / 1e15);
        }
    return 0;
}
