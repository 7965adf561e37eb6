// extern "C" surface of the DEVELOPER library only (libqlinear_hip_dev.so, include/qlinear_hip_dev.h): the measured dead ends kept
// as recorded experiments - the chained-grid MLP pair, the persistent MLP engine, W4A8.  Same conventions as abi.hip.
#include "launch.h"
#include "../../include/qlinear_hip.h"
#include "../../include/qlinear_hip_dev.h"

namespace ql {
static inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }
static inline bool fits_i32(int64_t v) { return v > 0 && v < ((int64_t)1 << 31); }
}  // namespace ql

using namespace ql;

extern "C" {

size_t qlinear_w4g32_mlp_pair_workspace_bytes(void) { return w4_mlp_pair_workspace_bytes(); }

int qlinear_w4g32_mlp_pair(const void* X, const void* ln_weight, float eps, const void* packed_in, const void* bias_in, int64_t N_in,
                           const void* packed_out, const void* bias_out, int64_t N_out, int64_t K, const void* residual, void* mid,
                           void* Out, void* workspace, int dtype, void* stream) {
    if (!X || !ln_weight || !packed_in || !packed_out || !residual || !mid || !Out || !workspace) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(N_in) || !fits_i32(N_out) || !fits_i32(K) || N_in <= 0 || N_out <= 0 || K <= 0 || K % 32 != 0 || N_in % 4 != 0 ||
        (N_in / 2) % 32 != 0)
        return QL_ERR_BAD_SHAPE;
    if (!aligned(X, 16) || !aligned(ln_weight, 16) || !aligned(packed_in, 16) || !aligned(packed_out, 16) || !aligned(mid, 16) ||
        !aligned(workspace, 64))
        return QL_ERR_MISALIGNED;
    return w4_mlp_pair(dtype, X, ln_weight, eps, packed_in, bias_in, N_in, K, packed_out, bias_out, N_out, N_in / 2, residual, mid, Out,
                       workspace, (hipStream_t)stream);
}

size_t qlinear_w4g32_mlp_engine_workspace_bytes(int64_t N_in) { return N_in > 0 ? w4_mlp_engine_workspace_bytes(N_in) : 0; }

int qlinear_w4g32_mlp_engine_supported(int64_t N_in, int64_t K, int64_t N_out) {
    return w4_mlp_engine_supported(N_in, K, N_out, N_in / 2) ? 1 : 0;
}

static int mlp_engine_checked(const void* X, const void* ln_weight, float eps, const void* packed_in, const void* bias_in, int64_t N_in,
                              const void* packed_out, const void* bias_out, int64_t N_out, int64_t K, void* Out, void* workspace, int dtype,
                              int flags, void* trace, void* stream) {
    if (!X || !ln_weight || !packed_in || !packed_out || !Out || !workspace) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (flags & ~QL_FLAG_STRICT_ROUNDING) return QL_ERR_UNSUPPORTED;
    if (!fits_i32(N_in) || !fits_i32(N_out) || !fits_i32(K) || N_in <= 0 || N_out <= 0 || K <= 0 || K % 32 != 0 || N_in % 4 != 0 ||
        (N_in / 2) % 32 != 0 || N_out != K)
        return QL_ERR_BAD_SHAPE;
    if (!aligned(X, 16) || !aligned(ln_weight, 16) || !aligned(packed_in, 16) || !aligned(packed_out, 16) || !aligned(Out, 8) ||
        !aligned(workspace, 64) || (bias_out && !aligned(bias_out, 8)))
        return QL_ERR_MISALIGNED;
    if (X == Out) return QL_ERR_UNSUPPORTED;                  // every workgroup reads X (norm + residual) while others write Out
    return w4_mlp_engine(dtype, (flags & QL_FLAG_STRICT_ROUNDING) != 0, X, ln_weight, eps, packed_in, bias_in, N_in, K, packed_out, bias_out,
                         N_out, N_in / 2, Out, workspace, trace, (hipStream_t)stream);
}

int qlinear_w4g32_mlp_engine(const void* X, const void* ln_weight, float eps, const void* packed_in, const void* bias_in, int64_t N_in,
                             const void* packed_out, const void* bias_out, int64_t N_out, int64_t K, void* Out, void* workspace, int dtype,
                             int flags, void* stream) {
    return mlp_engine_checked(X, ln_weight, eps, packed_in, bias_in, N_in, packed_out, bias_out, N_out, K, Out, workspace, dtype, flags, nullptr,
                              stream);
}

#ifdef QL_ENGINE_TRACE
// developer build (make trace): `trace` receives 16 s_memrealtime stamps per workgroup (tools/mlp_engine.py)
int qlinear_w4g32_mlp_engine_trace(const void* X, const void* ln_weight, float eps, const void* packed_in, const void* bias_in, int64_t N_in,
                                   const void* packed_out, const void* bias_out, int64_t N_out, int64_t K, void* Out, void* workspace,
                                   int dtype, int flags, void* trace, void* stream) {
    return mlp_engine_checked(X, ln_weight, eps, packed_in, bias_in, N_in, packed_out, bias_out, N_out, K, Out, workspace, dtype, flags, trace,
                              stream);
}
#endif

size_t qlinear_w4a8_packed_bytes(int64_t N, int64_t K, int64_t group, int dtype) {
    if (N <= 0 || K <= 0 || group != 32 || K % 32 != 0 || (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16)) return 0;
    return w4a8_packed_bytes(N, K, dtype);
}

int qlinear_w4a8_pack(const uint8_t* Wq, const void* S, void* packed_a8, int64_t N, int64_t K, int64_t group, int dtype,
                      void* stream) {
    if (!Wq || !S || !packed_a8) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(N) || !fits_i32(K) || (K & 1)) return QL_ERR_BAD_SHAPE;
    if (group != 32 || K % 32 != 0) return QL_ERR_BAD_GROUP;
    if (!aligned(packed_a8, 16)) return QL_ERR_MISALIGNED;
    return w4a8_pack(dtype, Wq, S, packed_a8, N, K, (hipStream_t)stream);
}

int qlinear_w4a8_fwd(const int8_t* Aq, const float* a_scale, const void* packed_a8, const void* bias, void* C, int64_t M,
                     int64_t N, int64_t K, int64_t ldc, int dtype, void* stream) {
    if (!Aq || !a_scale || !packed_a8 || !C) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || ldc < N) return QL_ERR_BAD_SHAPE;
    if (K % 32 != 0) return QL_ERR_BAD_GROUP;
    if (!aligned(Aq, 16) || !aligned(packed_a8, 16)) return QL_ERR_MISALIGNED;
    return w4a8_gemm(dtype, Aq, a_scale, packed_a8, bias, C, M, N, K, ldc, (hipStream_t)stream);
}

int qlinear_w4a8_linear(const void* A, const void* packed_a8, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                        int64_t lda, int64_t ldc, int dtype, int flags, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !packed_a8 || !C) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || lda < K || ldc < N) return QL_ERR_BAD_SHAPE;
    if (K % 32 != 0) return QL_ERR_BAD_GROUP;
    if (flags & ~QL_FLAG_ACT_PER_TENSOR) return QL_ERR_UNSUPPORTED;
    if (!workspace || !aligned(workspace, 16) || workspace_bytes < qlinear_workspace_bytes(QL_OP_W4A8_LINEAR, M, N, K, 32))
        return QL_ERR_WORKSPACE;
    if (!aligned(packed_a8, 16)) return QL_ERR_MISALIGNED;
    int8_t* Aq = (int8_t*)workspace;
    float* a_scale = (float*)((char*)workspace + (((size_t)M * (size_t)K + 15) & ~(size_t)15));
    const int rc = act_quant_rowwise(dtype, A, Aq, a_scale, M, K, lda, (flags & QL_FLAG_ACT_PER_TENSOR) != 0, (hipStream_t)stream);
    if (rc) return rc;
    return w4a8_gemm(dtype, Aq, a_scale, packed_a8, bias, C, M, N, K, ldc, (hipStream_t)stream);
}

/* round 4 experiment: dequantise once per call, then a dense 16-bit GEMM (w4_dense256.hip) */
size_t qlinear_dev_dense256_image_bytes(int64_t N, int64_t K) { return N > 0 && K > 0 && K % 32 == 0 ? dense256_image_bytes(N, K) : 0; }

int qlinear_dev_dense256_expand(const void* tiled, const void* S, void* image, int64_t N, int64_t K, int dtype, int weight_bits, void* stream) {
    if (!tiled || !image || (weight_bits == 8 && !S)) return QL_ERR_NULL_POINTER;
    if (weight_bits != 4 && weight_bits != 8) return QL_ERR_UNSUPPORTED;
    if (!fits_i32(N) || !fits_i32(K) || K % 32 != 0 || (weight_bits == 8 && K % 64 != 0)) return QL_ERR_BAD_SHAPE;
    if (!aligned(tiled, 16) || !aligned(image, 16)) return QL_ERR_MISALIGNED;
    return dense256_expand(dtype, weight_bits == 8, tiled, S, image, N, K, (hipStream_t)stream);
}

int qlinear_dev_dense256_fwd(const void* A, const void* image, const void* bias, const void* residual, void* C, int64_t M, int64_t N, int64_t K,
                             int64_t lda, int64_t ldc, int64_t ldr, int dtype, int gate, void* stream) {
    if (!A || !image || !C) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || lda < K || ldc < (gate ? N / 2 : N) || (gate && N % 8 != 0)) return QL_ERR_BAD_SHAPE;
    if (!dense256_can_run(M, N, K, lda, A)) return QL_ERR_UNSUPPORTED;
    if (gate && (residual || ldc % 4 != 0 || !aligned(C, 8))) return QL_ERR_MISALIGNED;
    return dense256(dtype, gate != 0, A, image, bias, residual, C, M, N, K, lda, ldc, ldr, (hipStream_t)stream);
}

size_t qlinear_dev_w8a8_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || !w8a8_splitk_serves(M, N, K)) return 0;
    return w8a8_splitk_workspace_bytes(M, N);
}

int qlinear_dev_w8a8_fwd_tiled_splitk(const int8_t* Aq, const float* a_scale, const void* tiled, const void* S, const void* bias, void* C,
                                  int64_t M, int64_t N, int64_t K, int64_t ldc, int dtype, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    if (!Aq || !a_scale || !tiled || !S || !C) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F32 && dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || ldc < N) return QL_ERR_BAD_SHAPE;
    if (!aligned(Aq, 16) || !aligned(tiled, 16)) return QL_ERR_MISALIGNED;
    if (!w8a8_splitk_serves(M, N, K)) return QL_ERR_UNSUPPORTED;
    if (!workspace || !aligned(workspace, 256) || workspace_bytes < w8a8_splitk_workspace_bytes(M, N)) return QL_ERR_WORKSPACE;
    return w8a8_gemm_tiled_splitk(dtype, Aq, a_scale, (const int8_t*)tiled, S, bias, C, M, N, K, ldc, workspace, (hipStream_t)stream);
}

}  // extern "C"
