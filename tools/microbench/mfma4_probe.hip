// Lane mapping probe for v_mfma_f32_4x4x4_16b_f16 (16 blocks of 4x4x4): which (A lane, k) meets (B lane, k) in which (D lane, vgpr).
//   hipcc --offload-arch=gfx950 -O3 mfma4_probe.hip -o mfma4_probe.bin && ./mfma4_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(int lb, int kb, float* out) {
    const int l = threadIdx.x;
    h4 a, b;
    for (int e = 0; e < 4; ++e) {
        a[e] = (_Float16)(float)(1 + l * 4 + e);
        b[e] = (_Float16)((l == lb && e == kb) ? 1.f : 0.f);
    }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = c[e];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    float h[256];
    const int probes[][2] = {{0, 0}, {0, 1}, {1, 0}, {2, 3}, {5, 2}, {63, 1}};
    for (auto& p : probes) {
        k<<<1, 64>>>(p[0], p[1], d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("B lane %d k %d ->", p[0], p[1]);
        for (int i = 0; i < 256; ++i)
            if (h[i] != 0.f) { int id = (int)h[i] - 1; printf("  D[lane %d][v %d] = A[lane %d][k %d]", i / 4, i % 4, id / 4, id % 4); }
        printf("\n");
    }
    return 0;
}
