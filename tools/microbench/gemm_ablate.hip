// Developer tool: time the int4g32 MFMA GEMM kernel with parts of its loop compiled out (QL_GEMM_ABLATE bits, see
// w4_gemm.hip) to attribute per-step time.  Results are garbage by construction; only durations matter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DQL_GEMM_ABLATE=<bits> gemm_ablate.hip -o gemm_ablate_<bits>.bin
//   ./gemm_ablate_<bits>.bin M [MT] [KSPLIT]
#include "../../chatglm_q_amd/csrc/w4_gemm.hip"
#include <stdio.h>
#include <vector>

namespace ql { int finish_launch() { return (int)hipGetLastError(); } }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 256, N = 4096, K = 4096;
    const int NSETS = 24, REPS = 4;
    const size_t pbytes = (size_t)N * (K / 32) * 16 + (size_t)N * (K / 32) * 2;
    std::vector<void*> packed(NSETS);
    std::vector<uint32_t> h(pbytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) & 0x3BFF3BFFu;   // finite fp16 scales, any nibbles
    for (auto& p : packed) { CK(hipMalloc(&p, pbytes)); CK(hipMemcpy(p, h.data(), pbytes, hipMemcpyHostToDevice)); }
    void *A, *C, *ws;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMemset(A, 0x3c, (size_t)M * K * 2));
    CK(hipMalloc(&C, (size_t)M * N * 2));
    const size_t wsb = (size_t)16 * M * N * 4;
    CK(hipMalloc(&ws, wsb));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto run = [&] { for (int r = 0; r < REPS; ++r) for (auto p : packed) ql::w4_packed_gemm(QL_DTYPE_F16, A, p, nullptr, C, M, N, K, K, N, ws, wsb, st); };
    run(); CK(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal)); run(); CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int i = 0; i < 3; ++i) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double us = best * 1e3 / (NSETS * REPS);
    printf("ablate=%2d M=%5d  %8.2f us  %7.1f TFLOP/s-equivalent\n", QL_GEMM_ABLATE, M, us, 2.0 * M * N * K / us / 1e6);
    return 0;
}
