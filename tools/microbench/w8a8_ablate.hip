// Developer tool: time the i8 x i8 MFMA GEMM (512 x 4096 x 4096 by default) with parts of its loop compiled out
// (QL_W8A8_ABLATE bits, see w8_kernels.hip).  Results are garbage by construction; only durations matter.
#include "../../chatglm_q_amd/csrc/w8_kernels.hip"
#include <stdio.h>
#include <vector>
namespace ql { int finish_launch() { return (int)hipGetLastError(); } }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 512, N = 4096, K = 4096, NL = 16;
    std::vector<void*> w(NL);
    std::vector<uint32_t> h((size_t)N * K / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u);
    for (auto& p : w) { CK(hipMalloc(&p, (size_t)N * K)); CK(hipMemcpy(p, h.data(), (size_t)N * K, hipMemcpyHostToDevice)); }
    void *aq, *as, *s, *c;
    CK(hipMalloc(&aq, (size_t)M * K)); CK(hipMemcpy(aq, h.data(), (size_t)M * K, hipMemcpyHostToDevice));
    CK(hipMalloc(&as, M * 4)); CK(hipMemset(as, 0x3c, M * 4)); CK(hipMalloc(&s, N * 2)); CK(hipMemset(s, 0x2c, N * 2));
    CK(hipMalloc(&c, (size_t)M * N * 2));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto run = [&] { for (int r = 0; r < 4; ++r) for (auto p : w) ql::w8a8_gemm(QL_DTYPE_F16, (const int8_t*)aq, (const float*)as, (const int8_t*)p, s, nullptr, c, M, N, K, N, nullptr, 0, st); };
    run(); CK(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal)); run(); CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int i = 0; i < 3; ++i) { CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    const double us = best * 1e3 / (NL * 4);
    printf("ablate=%2d M=%d  %7.2f us  %7.1f TOP/s-equivalent\n", QL_W8A8_ABLATE, M, us, 2.0 * M * N * K / us / 1e6);
    return 0;
}
