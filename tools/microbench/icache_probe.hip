// Micro-benchmark (developer tool): what a COLD instruction stream costs at the start of a kernel.
// One workgroup of W waves runs a straight-line block of N VALU instructions (8 bytes each, four independent chains) three times inside
// one launch: pass 0 fetches the block from L2 / HBM (the instruction cache is invalidated at every kernel boundary), passes 1 and 2 run
// from the instruction cache.  The 100 MHz wall clock is read around each pass (lane 0 of wave 0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 icache_probe.hip -o icache_probe.bin && ./icache_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define BLOCK_ASM(REPT)                                                                                                  \
    asm volatile(".rept " #REPT "\n v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n" \
                 " v_add3_u32 %3, %3, %4, %4\n .endr"                                                                    \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d)                                                                    \
                 : "v"(one))

template <int REPT>
__global__ __launch_bounds__(512) void code_kernel(unsigned long long* stamps, uint32_t* out, int passes) {
    uint32_t a = threadIdx.x, b = 1, c = 2, d = 3, one = 1;
    const bool rec = threadIdx.x == 0;
    unsigned long long t[5];
    t[0] = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int p = 0; p < passes; ++p) {
        if constexpr (REPT == 64) BLOCK_ASM(64);
        if constexpr (REPT == 256) BLOCK_ASM(256);
        if constexpr (REPT == 1024) BLOCK_ASM(1024);
        asm volatile("" : "+v"(a));
        if (p < 4) t[p + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (rec)
        for (int i = 0; i < 5; ++i) stamps[i] = t[i];
    if (a + b + c + d == 0x12345) out[0] = a;
}

__global__ void other_kernel(uint32_t* out) { if (threadIdx.x == 9999) out[0] = 1; }

template <int REPT>
static void run(int waves, unsigned long long* dst, uint32_t* out) {
    std::vector<double> p0, p1, p2;
    for (int it = 0; it < 20; ++it) {
        other_kernel<<<256, 64>>>(out);
        code_kernel<REPT><<<1, waves * 64>>>(dst, out, 3);
        unsigned long long h[5];
        hipMemcpy(h, dst, sizeof(h), hipMemcpyDeviceToHost);
        p0.push_back((h[1] - h[0]) * 0.01), p1.push_back((h[2] - h[1]) * 0.01), p2.push_back((h[3] - h[2]) * 0.01);
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double kb = REPT * 4 * 8 / 1024.0;
    const double c = med(p0), w = med(p2);
    printf("%5.1f KB of straight-line code (%4d instructions), %d wave(s): cold pass %.2f us, warm passes %.2f / %.2f us -> fetch adds %.2f us = %.0f ns per 64-byte line\n",
           kb, REPT * 4, waves, c, med(p1), w, c - w, (c - w) * 1e3 / (kb * 16));
}

int main() {
    unsigned long long* dst;
    uint32_t* out;
    hipMalloc(&dst, 64);
    hipMalloc(&out, 64);
    for (int waves : {1, 8}) {
        run<64>(waves, dst, out);
        run<256>(waves, dst, out);
        run<1024>(waves, dst, out);
    }
    return 0;
}
