// Launch-floor probe (developer tool): what an (almost) empty kernel costs as a function of block size, VGPR
// allocation and LDS allocation, measured as a chain of 20 dependent launches replayed from one hipGraph.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 launch_probe.hip -o launch_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int THREADS, int NV>
__global__ __launch_bounds__(THREADS) void hog_kernel(int* out, int flag) {
    extern __shared__ char smem[];
    if (NV >= 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (NV >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (NV >= 188) asm volatile("v_mov_b32 v187, 0" ::: "v187");
    if (NV >= 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    if (flag == 12345) { smem[threadIdx.x] = 1; out[threadIdx.x] = smem[threadIdx.x ^ 1]; }
}

template <typename F>
static float time_chain(hipStream_t st, F launch, int n = 20, int reps = 5) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) launch();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best * 1e3f / n;
}

template <int THREADS, int NV>
static void run(hipStream_t st, int* out, int blocks, int lds) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&hog_kernel<THREADS, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float us = time_chain(st, [&] { hog_kernel<THREADS, NV><<<blocks, THREADS, lds, st>>>(out, 0); });
    printf("threads %4d  vgpr>=%3d  lds %6d  blocks %5d : %7.2f us per launch\n", THREADS, NV, lds, blocks, us);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    int* out; CK(hipMalloc(&out, 1 << 20));
    for (int blocks : {256, 2048}) {
        run<256, 32>(st, out, blocks, 0);
        run<512, 32>(st, out, blocks, 0);
        run<512, 128>(st, out, blocks, 0);
        run<512, 188>(st, out, blocks, 0);
        run<512, 256>(st, out, blocks, 0);
        run<512, 188>(st, out, blocks, 32768);
        run<512, 256>(st, out, blocks, 65536);
        run<256, 256>(st, out, blocks, 0);
        run<256, 128>(st, out, blocks, 32768);
        run<1024, 128>(st, out, blocks, 0);
    }
    return 0;
}
