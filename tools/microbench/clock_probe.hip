// Is s_memtime's tick the shader clock while the matrix pipes are saturated?  Blocks of 9 waves: waves 0..7 (two per SIMD) run
// independent v_mfma_i32_32x32x32_i8 back to back (or idle, MODE 0), wave 8 runs a chain of DEPENDENT v_fma_f32 - a fixed number
// of real cycles each - and stamps s_memtime (ticks) and s_memrealtime (100 MHz) around it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w clock_probe.hip -o clock_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(576) void k(unsigned long long* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    if (wave < 8) {
        if (MODE == 0) return;
        i32x4 a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)threadIdx.x, 8};
        i32x16 c[4] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[u], 0, 0, 0);
        if ((c[0][0] ^ c[1][0] ^ c[2][0] ^ c[3][0]) == 0x12345) out[8] = 1;
        return;
    }
    float x = seed;
    const int n = iters / 2;                       // well inside the time the MFMA waves run (they take ~5 x longer)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 512 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)n * 16; out[3] = (unsigned long long)x; }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 128);
    unsigned long long h[4];
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) k<0><<<256, 576>>>(d, 20000, 1.0f); else k<1><<<256, 576>>>(d, 20000, 1.0f);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("%s: dependent v_fma_f32 chain: %.2f s_memtime ticks per op, %.2f ns per op (s_memrealtime) -> tick rate %.2f GHz\n",
               mode ? "matrix pipes saturated (2 waves per SIMD of i8 MFMA)" : "matrix pipes idle", (double)h[0] / h[2], (double)h[1] * 10.0 / h[2],
               (double)h[0] / (h[1] * 10.0));
    }
    return 0;
}
