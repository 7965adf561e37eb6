// Developer tool: time the one-launch rotary + cache write + attention step with parts compiled out
// (QL_ATT_ABLATE bits, see decode_ops.hip): 1 rotary operand loads, 2 key loads, 4 softmax block reductions,
// 8 P.V phase, 16 Q.K phase.  Results are garbage by construction; only durations matter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DQL_ATT_ABLATE=<bits> attn_ablate.hip -o attn_ablate_<bits>.bin
#include "../../chatglm_q_amd/csrc/decode_ops.hip"
#include <stdio.h>
#include <vector>
namespace ql { int finish_launch() { return (int)hipGetLastError(); } }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    const int cap = argc > 1 ? atoi(argv[1]) : 192, n = cap / 2, B = 1, H = 32, G = 2, D = 128, NL = 28;
    std::vector<void*> kc(NL), vc(NL);
    for (int i = 0; i < NL; ++i) { CK(hipMalloc(&kc[i], (size_t)B * cap * G * D * 2)); CK(hipMemset(kc[i], 0x2c, (size_t)B * cap * G * D * 2));
                                   CK(hipMalloc(&vc[i], (size_t)B * cap * G * D * 2)); CK(hipMemset(vc[i], 0x2c, (size_t)B * cap * G * D * 2)); }
    void *qkv, *table, *out; int64_t *pos, *widx; float* mask;
    CK(hipMalloc(&qkv, (H + 2 * G) * D * 2)); CK(hipMemset(qkv, 0x2c, (H + 2 * G) * D * 2));
    CK(hipMalloc(&table, (size_t)(cap + 8) * D * 2)); CK(hipMemset(table, 0x38, (size_t)(cap + 8) * D * 2));
    CK(hipMalloc(&out, H * D * 2)); CK(hipMalloc(&pos, 8)); CK(hipMalloc(&widx, 8)); CK(hipMalloc(&mask, cap * 4));
    int64_t hp = n + 1, hw = n; CK(hipMemcpy(pos, &hp, 8, hipMemcpyHostToDevice)); CK(hipMemcpy(widx, &hw, 8, hipMemcpyHostToDevice));
    std::vector<float> hm(cap, -1e10f); for (int i = 0; i <= n; ++i) hm[i] = 0.f; CK(hipMemcpy(mask, hm.data(), cap * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto run = [&] { for (int r = 0; r < 4; ++r) for (int i = 0; i < NL; ++i) ql::decode_attention_rope(QL_DTYPE_F16, qkv, table, pos, widx, kc[i], vc[i], mask, out, B, H, G, D, cap, (H + 2 * G) * D, st); };
    run(); CK(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal)); run(); CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int i = 0; i < 3; ++i) { CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("ablate=%2d capacity=%d  %6.2f us per launch\n", QL_ATT_ABLATE, cap, best * 1e3 / (NL * 4));
    return 0;
}
