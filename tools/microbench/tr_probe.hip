#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int pitch_elems) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    // group q reads block rows 4q..4q+3, 16 cols; lane i supplies row 4q + i/4, cols 4*(i%4)..+3
    const uint16_t* p = lds + (4 * q + i / 4) * pitch_elems + 4 * (i % 4);
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (uint16_t)r[e];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    for (int pitch : {16, 144}) {
        probe<<<1, 64>>>(d, pitch);
        uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pitch %d elements (value = row*pitch + col)\n", pitch);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l * 4 + e] / pitch, h[l * 4 + e] % pitch);
            printf("\n");
        }
    }
    return 0;
}
