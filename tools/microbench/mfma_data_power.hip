// How much of the dense MFMA peak survives RANDOM operands?  MFMA-only loops (two waves per SIMD, 4 independent accumulators), the A / B
// fragments either constant small values or 8 rotating register sets of random bits per lane; sustained over tens of ms (DVFS settles).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w mfma_data_power.hip -o mfma_data_power.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int DATA>   // KIND 0: i8 32x32x32, 1: f16 32x32x16, 2: i8 16x16x64, 3: f16 16x16x32 (round 4).  DATA 0: constant operands, 1: random bits, 2: random f16 in [-1, 1) / int8 full range
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters) {
    i32x4 a[8], b[8];
    for (int s = 0; s < 8; ++s)
        for (int e = 0; e < 4; ++e) {
            unsigned ra = hash(threadIdx.x * 131u + s * 17u + e), rb = hash(threadIdx.x * 257u + s * 29u + e + 99u);
            if (DATA == 0) { ra = (KIND & 1) ? 0x3c003c00u : 0x01010101u; rb = ra; }
            if (DATA == 2 && (KIND & 1)) { ra = (ra & 0x83ff83ffu) | 0x38003800u; rb = (rb & 0x83ff83ffu) | 0x38003800u; }   // +-[0.5, 1)
            a[s][e] = (int)ra; b[s][e] = (int)rb;
        }
    i32x16 ci[4] = {};
    f32x16 cf[4] = {};
    i32x4 di[8] = {};
    f32x4 df[8] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (KIND == 2) {
                    di[2 * u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(s + u) & 7], b[(s + 2 * u) & 7], di[2 * u], 0, 0, 0);
                    di[2 * u + 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(s + u + 3) & 7], b[(s + 2 * u + 1) & 7], di[2 * u + 1], 0, 0, 0);
                } else if (KIND == 3) {
                    df[2 * u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(s + u) & 7]), __builtin_bit_cast(f16x8, b[(s + 2 * u) & 7]), df[2 * u], 0, 0, 0);
                    df[2 * u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(s + u + 3) & 7]), __builtin_bit_cast(f16x8, b[(s + 2 * u + 1) & 7]), df[2 * u + 1], 0, 0, 0);
                } else if (KIND == 0) ci[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[(s + u) & 7], b[(s + 2 * u) & 7], ci[u], 0, 0, 0);
                else cf[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(s + u) & 7]), __builtin_bit_cast(f16x8, b[(s + 2 * u) & 7]), cf[u], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int x = 0;
    for (int u = 0; u < 4; ++u) x ^= ci[u][0] ^ (int)cf[u][0];
    for (int u = 0; u < 8; ++u) x ^= di[u][0] ^ (int)df[u][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}
template <int KIND, int DATA>
void run(const char* name) {
    unsigned long long* d; hipMalloc(&d, 64);
    const int iters = 40000;
    k<KIND, DATA><<<256, 512>>>(d, iters / 20);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<KIND, DATA><<<256, 512>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double macs = ((KIND & 1) == 0 ? 32.0 * 32 * 32 : 32.0 * 32 * 16) * 32.0 * iters * 8 * 256;   // the 16x16 kinds issue two half-size MFMAs per slot
    printf("%-52s %7.2f ms: %5.2f P(FL)OP/s, clock %.2f GHz, %.1f cycles per MFMA per SIMD\n", name, ms, macs * 2 / (ms * 1e-3) / 1e15,
           (double)h[0] / (ms * 1e-3) / 1e9, (double)h[0] / (32.0 * iters) / 2);
    hipFree(d);
}
int main() {
    run<0, 0>("i8 32x32x32, constant operands");
    run<0, 1>("i8 32x32x32, random int8 operands");
    run<1, 0>("f16 32x32x16, constant operands (1.0)");
    run<1, 2>("f16 32x32x16, random operands in +-[0.5, 1)");
    run<1, 1>("f16 32x32x16, random bit patterns");
    run<2, 0>("i8 16x16x64 (x2), constant operands");
    run<2, 1>("i8 16x16x64 (x2), random int8 operands");
    run<3, 0>("f16 16x16x32 (x2), constant operands (1.0)");
    run<3, 2>("f16 16x16x32 (x2), random operands in +-[0.5, 1)");
    run<3, 1>("f16 16x16x32 (x2), random bit patterns");
    run<0, 1>("i8 32x32x32, random int8 operands (again)");
    run<1, 2>("f16 32x32x16, random +-[0.5, 1) (again)");
    return 0;
}
