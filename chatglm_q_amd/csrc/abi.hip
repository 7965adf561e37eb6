// extern "C" surface of libqlinear_hip.so: argument validation, kernel selection, status mapping.
// Contract: include/qlinear_hip.h.  Nothing here allocates, synchronises or touches the default
// stream; every launch goes to the caller's stream.
#include <atomic>
#include <cstring>

#include "launch.h"
#include "../../include/qlinear_hip.h"

namespace ql {

static std::atomic<uint64_t> g_launches{0};
static thread_local uint64_t t_dispatch_log = 0;       // newest launch in the low byte (qlinear_last_dispatch)

int finish_launch(int kernel_family) {
    g_launches.fetch_add(1, std::memory_order_relaxed);
    t_dispatch_log = (t_dispatch_log << 8) | (uint64_t)(kernel_family & 0xFF);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? QL_OK : (int)e;
}

// QLINEAR_DISPATCH (tune.h): the one environment variable the product library reads
static unsigned parse_dispatch_env() {
    const char* e = getenv("QLINEAR_DISPATCH");
    unsigned v = 0x80000000u;                          // "parsed" marker
    if (!e) return v;
    if (strstr(e, "no256")) v |= QL_D_NO256;
    if (strstr(e, "nopeel")) v |= QL_D_NOPEEL;
    if (strstr(e, "nohalf")) v |= QL_D_NOHALF;
    if (strstr(e, "nof32mfma")) v |= QL_D_NOF32MFMA;
    if (strstr(e, "nofewrow")) v |= QL_D_NOFEWROW;
    if (strstr(e, "norows16")) v |= QL_D_NOROWS16;
    if (strstr(e, "norows4")) v |= QL_D_NOROWS4;
    if (strstr(e, "nogroupattn")) v |= QL_D_NOGROUPATTN;
    return v;
}
static std::atomic<unsigned> g_dispatch{0};
unsigned dispatch_flags() {
    unsigned v = g_dispatch.load(std::memory_order_relaxed);
    if (!v) {
        v = parse_dispatch_env();
        g_dispatch.store(v, std::memory_order_relaxed);
    }
    return v;
}

// CU count of the current device: the idempotent per-device property cache the ABI allows.  256 (the MI355X this
// library is written for) when no device answers, so that the host-side planning functions stay usable without a GPU.
int cu_count() {
    static std::atomic<int> cache[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    cache[dev].store(n, std::memory_order_relaxed);
    return n;
}

static inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }
static inline bool dtype_ok(int d) { return d == QL_DTYPE_F32 || d == QL_DTYPE_F16 || d == QL_DTYPE_BF16; }
static inline int64_t esize(int d) { return d == QL_DTYPE_F32 ? 4 : 2; }
static inline bool fits_i32(int64_t v) { return v > 0 && v < ((int64_t)1 << 31); }

// A rows must be addressable with 16-byte vector loads by the fast kernels
static inline bool act_vec_ok(const void* A, int64_t lda, int dtype) {
    return aligned(A, 16) && (lda * esize(dtype)) % 16 == 0;
}

}  // namespace ql

using namespace ql;

extern "C" {

int qlinear_abi_version(void) { return QLINEAR_ABI_VERSION; }

const char* qlinear_status_string(int status) {
    switch (status) {
    case QL_OK: return "ok";
    case QL_ERR_NULL_POINTER: return "null pointer argument";
    case QL_ERR_BAD_SHAPE: return "bad shape (M, N, K must be positive, K even and divisible by group, ld* >= row length)";
    case QL_ERR_BAD_DTYPE: return "unsupported activation dtype (expected f32=0, f16=1, bf16=2)";
    case QL_ERR_BAD_GROUP: return "unsupported quantisation group size for this entry point";
    case QL_ERR_MISALIGNED: return "pointer or leading dimension violates the alignment contract";
    case QL_ERR_WORKSPACE: return "workspace missing or too small (see qlinear_workspace_bytes)";
    case QL_ERR_UNSUPPORTED: return "shape not supported by this entry point";
    default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown qlinear status";
}

uint64_t qlinear_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

uint64_t qlinear_last_dispatch(void) { return t_dispatch_log; }
void qlinear_dispatch_reset(void) { t_dispatch_log = 0; }
void qlinear_dispatch_reload(void) { g_dispatch.store(parse_dispatch_env(), std::memory_order_relaxed); }

size_t qlinear_workspace_bytes(int op, int64_t M, int64_t N, int64_t K, int64_t group) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    switch (op) {
    case QL_OP_W4G32_FWD:
        if (group == 32 && N % 8 == 0) return w4_canon_workspace_bytes(M, N, K);
        return 0;
    case QL_OP_W4G32_FWD_PACKED:     // optional: without it the few-row MFMA GEMM runs unsplit (slower, same results)
        return (group == 32 && K % 32 == 0) ? w4_packed_workspace_bytes(M, N, K) : 0;
    case QL_OP_W8_FWD:
        return (M > 4 && K % 16 == 0) ? w8_gemm_workspace_bytes(M, N, K) : 0;
    case QL_OP_W8_FWD_TILED:
        return (M > 2 && K % 16 == 0) ? w8_tiled_workspace_bytes(M, N, K) : 0;
    case QL_OP_W8A8_FWD:             // optional too: int32 split-K slabs for shapes with few row tiles
        return K % 16 == 0 ? w8a8_workspace_bytes(M, N, K) : 0;
    case QL_OP_W4A8_LINEAR:
    case QL_OP_W8A8_LINEAR_TILED:    // required: Aq (M x K int8, padded to 16 bytes) + a_scale (M floats)
        return (((size_t)M * (size_t)K + 15) & ~(size_t)15) + (size_t)M * sizeof(float);
    default: return 0;
    }
}

int qlinear_w4g32_fwd(const void* A, const uint8_t* Wq, const void* S, const void* bias, void* C, int64_t M,
                      int64_t N, int64_t K, int64_t group, int64_t lda, int64_t ldc, int dtype, void* workspace,
                      size_t workspace_bytes, void* stream) {
    if (!A || !Wq || !S || !C) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || (K & 1) || lda < K || ldc < N) return QL_ERR_BAD_SHAPE;
    if (group <= 0 || K % group != 0) return QL_ERR_BAD_GROUP;
    hipStream_t st = (hipStream_t)stream;
    // fp32 activations, 128+ rows: the fp32 matrix instruction on the canonical buffers (wq_gemm_f32.hip, round 5)
    if (dtype == QL_DTYPE_F32 && group == 32 && act_vec_ok(A, lda, dtype) && wq_gemm_f32_serves(M, N, K) && !(dispatch_flags() & QL_D_NOF32MFMA))
        return w4_gemm_f32(A, Wq, S, bias, C, M, N, K, lda, ldc, st);
    const bool fast = group == 32 && N % 8 == 0 && act_vec_ok(A, lda, dtype) && aligned(Wq, 8) && aligned(S, 16);
    if (!fast) return w4_generic(dtype, A, Wq, S, bias, C, M, N, K, group, lda, ldc, st);
    const size_t need = w4_canon_workspace_bytes(M, N, K);
    if (need && (!workspace || workspace_bytes < need || !aligned(workspace, 16))) return QL_ERR_WORKSPACE;
    return w4_canon(dtype, A, Wq, S, bias, C, workspace, M, N, K, lda, ldc, st);
}

int qlinear_w4g32_bwd_input(const void* Gout, const uint8_t* Wq, const void* S, void* dA, int64_t M, int64_t N, int64_t K,
                            int64_t group, int64_t ldg, int64_t ldda, int dtype, void* stream) {
    if (!Gout || !Wq || !S || !dA) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || (K & 1) || ldg < N || ldda < K) return QL_ERR_BAD_SHAPE;
    if (group != 32 || K % 32 != 0) return QL_ERR_BAD_GROUP;
    if (N % 16 != 0 || N < 16 || (ldda & 1)) return QL_ERR_UNSUPPORTED;    /* 16-byte loads along the contraction */
    if (!aligned(Wq, 16) || !aligned(S, 16) || !act_vec_ok(Gout, ldg, dtype) || !aligned(dA, 4)) return QL_ERR_MISALIGNED;
    return w4_tgemm(dtype, Gout, Wq, S, dA, M, K, N, ldg, N, N, ldda, (hipStream_t)stream);
}

size_t qlinear_w4g32_packed_bytes(int64_t N, int64_t K, int64_t group, int dtype) {
    if (N <= 0 || K <= 0 || group != 32 || K % 32 != 0 || !dtype_ok(dtype)) return 0;
    return w4_layout(N, K, esize(dtype)).bytes;
}

int qlinear_w4g32_repack(const uint8_t* Wq, const void* S, void* packed, int64_t N, int64_t K, int64_t group,
                         int dtype, void* stream) {
    if (!Wq || !S || !packed) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(N) || !fits_i32(K) || (K & 1)) return QL_ERR_BAD_SHAPE;
    if (group != 32 || K % 32 != 0) return QL_ERR_BAD_GROUP;
    if (!aligned(packed, 16)) return QL_ERR_MISALIGNED;
    return w4_repack(dtype, Wq, S, packed, N, K, (hipStream_t)stream);
}

size_t qlinear_w4g32_gemv_bytes(int64_t N, int64_t K, int64_t group, int dtype) {
    if (N <= 0 || K <= 0 || group != 32 || K % 32 != 0 || !dtype_ok(dtype)) return 0;
    return w4_layout(N, K, esize(dtype)).off_wm;
}

size_t qlinear_w4g32_tiled_bytes(int64_t N, int64_t K, int64_t group, int dtype) {
    if (N <= 0 || K <= 0 || group != 32 || K % 32 != 0 || (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16)) return 0;
    const W4Layout L = w4_layout(N, K, esize(dtype));
    return L.bytes - L.off_wm;
}

int qlinear_w4g32_repack_gemv(const uint8_t* Wq, const void* S, void* gemv, int64_t N, int64_t K, int64_t group, int dtype,
                              void* stream) {
    if (!Wq || !S || !gemv) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(N) || !fits_i32(K) || (K & 1)) return QL_ERR_BAD_SHAPE;
    if (group != 32 || K % 32 != 0) return QL_ERR_BAD_GROUP;
    if (!aligned(gemv, 16)) return QL_ERR_MISALIGNED;
    return w4_repack_gemv(dtype, Wq, S, gemv, N, K, (hipStream_t)stream);
}

int qlinear_w4g32_unpack_gemv(const void* gemv, uint8_t* Wq, void* S, int64_t N, int64_t K, int64_t group, int dtype, void* stream) {
    if (!Wq || !S || !gemv) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(N) || !fits_i32(K) || N <= 0 || K <= 0 || (K & 1)) return QL_ERR_BAD_SHAPE;
    if (group != 32 || K % 32 != 0) return QL_ERR_BAD_GROUP;
    if (!aligned(gemv, 16)) return QL_ERR_MISALIGNED;
    return w4_unpack_gemv(dtype, gemv, Wq, S, N, K, (hipStream_t)stream);
}

int qlinear_w4g32_tile(const void* gemv, void* tiled, int64_t N, int64_t K, int64_t group, int dtype, void* stream) {
    if (!gemv || !tiled) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(N) || !fits_i32(K) || N <= 0 || K <= 0) return QL_ERR_BAD_SHAPE;
    if (group != 32 || K % 32 != 0) return QL_ERR_BAD_GROUP;
    if (!aligned(gemv, 16) || !aligned(tiled, 16)) return QL_ERR_MISALIGNED;
    return w4_tile(dtype, gemv, tiled, N, K, (hipStream_t)stream);
}

int qlinear_w4g32_fwd_tiled256(const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                               int64_t group, int64_t lda, int64_t ldc, int dtype, void* stream) {
    if (!A || !tiled || !C) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || lda < K || ldc < N) return QL_ERR_BAD_SHAPE;
    if (group != 32 || K % 32 != 0) return QL_ERR_BAD_GROUP;
    if (!aligned(tiled, 16) || !act_vec_ok(A, lda, dtype)) return QL_ERR_MISALIGNED;
    if (!w4_gemm256_can_run(M, N, K, lda, A, esize(dtype))) return QL_ERR_UNSUPPORTED;
    return w4_gemm256(dtype, A, tiled, bias, C, M, N, K, lda, ldc, (hipStream_t)stream);
}

int qlinear_gemm256_serves(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return w4_gemm256_supported(M, N, K, K, nullptr, 2) ? 1 : 0;
}

int qlinear_tiled_dispatch(int weight_bits, int64_t M, int64_t N, int64_t K, int64_t* rows_first) {
    if (M <= 0 || N <= 0 || K <= 0 || (weight_bits != 4 && weight_bits != 8)) return 0;
    int64_t first = M;
    int family;
    if (weight_bits == 4) {
        if (w4_fewrow_supported(M, N, K)) family = QL_K_W4_FEWROW;
        else {
            const int64_t m256 = w4_gemm256_rows(M, N, K, K, nullptr, 2);
            family = m256 > 0 ? QL_K_W4_GEMM256 : QL_K_W4_GEMM128;
            if (m256 > 0) first = m256;
        }
    } else {
        family = M <= 32 ? QL_K_W8_FEWROW : (w4_gemm256_supported(M, N, K, K, nullptr, 2) ? QL_K_W8_GEMM256 : QL_K_W8_GEMM128);
    }
    if (rows_first) *rows_first = first;
    return family;
}

int qlinear_w4g32_fwd_tiled(const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                            int64_t group, int64_t lda, int64_t ldc, int dtype, void* workspace, size_t workspace_bytes,
                            void* stream) {
    if (!A || !tiled || !C) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || (K & 1) || lda < K || ldc < N) return QL_ERR_BAD_SHAPE;
    if (group != 32 || K % 32 != 0) return QL_ERR_BAD_GROUP;
    if (!aligned(tiled, 16) || !act_vec_ok(A, lda, dtype)) return QL_ERR_MISALIGNED;
    return w4_tiled(dtype, A, tiled, bias, C, M, N, K, lda, ldc, workspace, workspace_bytes, (hipStream_t)stream);
}

int qlinear_w4g32_rows_on_tiled(int64_t M, int64_t N, int64_t K, int dtype, int flags) {
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return 0;         // fp32 has no MFMA path
    if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0) return 0;
    return w4_rows_use_gemm(M, N, K) && !w4_rows4_serves(dtype, M, N, K, K, (flags & QL_FLAG_STRICT_ROUNDING) != 0) &&
                   !w4_rows16_serves(dtype, M, N, K, K)
               ? 1
               : 0;
}

int qlinear_w4g32_packed_dispatch(int64_t M, int64_t N, int64_t K, int dtype, int flags) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0 || !dtype_ok(dtype)) return 0;
    if (qlinear_w4g32_rows_on_tiled(M, N, K, dtype, flags)) return 0;
    if (w4_rows4_serves(dtype, M, N, K, K, (flags & QL_FLAG_STRICT_ROUNDING) != 0)) return QL_K_W4_ROWS4;
    if (w4_rows_use_gemm(M, N, K) && w4_rows16_serves(dtype, M, N, K, K)) return QL_K_W4_ROWS16;
    return QL_K_W4_GEMV;
}

unsigned qlinear_dispatch_flags(void) { return dispatch_flags() & 0x7FFFFFFFu; }

int qlinear_gated_serves(int64_t M, int64_t N, int64_t K, int dtype, int weight_bits) {
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return 0;
    if (M <= 0 || N <= 0 || K <= 0 || N % 32 != 0) return 0;
    // weight_bits 88: int8 ACTIVATIONS x int8 weights (qlinear_w8a8_fwd_tiled_gated) - the ring kernel's own rule (contiguous 16-byte aligned rows assumed)
    if (weight_bits == 88) return w8a8_gemm256_supported(dtype, M, N, K, nullptr) ? 1 : 0;
    // the 256 x 256-tile GEMM's SiLU * gate epilogue (contiguous 16-byte aligned rows assumed: the entry points re-check the operands)
    if (w4_gemm256_supported(M, N, K, K, nullptr, 2)) return 1;
    if (weight_bits != 4 || K % 32 != 0) return 0;
    // int4g32 only: the few-row kernel without K slabs (wide first MLP projections, 3..32 rows)
    return w4_rows_use_gemm(M, N, K) && w4_fewrow_supported(M, N, K) && w4_fewrow_workspace_bytes(M, N, K) == 0 ? 1 : 0;
}

int qlinear_w4g32_fwd_packed(const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N,
                             int64_t K, int64_t group, int64_t lda, int64_t ldc, int dtype, int flags,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !packed || !C) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || (K & 1) || lda < K || ldc < N) return QL_ERR_BAD_SHAPE;
    if (group != 32 || K % 32 != 0) return QL_ERR_BAD_GROUP;
    if (!aligned(packed, 16) || !act_vec_ok(A, lda, dtype)) return QL_ERR_MISALIGNED;
    return w4_packed(dtype, A, packed, bias, C, M, N, K, lda, ldc, (flags & QL_FLAG_STRICT_ROUNDING) != 0, workspace,
                     workspace_bytes, (hipStream_t)stream);
}

int qlinear_w4g32_fwd_packed_fused(int prologue, const void* A, const void* packed, const void* bias, void* C,
                                   int64_t N, int64_t K, const void* delta, const void* ln_weight, void* hout,
                                   float eps, int dtype, void* stream) {
    if (!A || !packed || !C) return QL_ERR_NULL_POINTER;
    const bool gate = (prologue & QL_EPI_SILU_GATE) != 0, strict = (prologue & QL_FUSED_STRICT) != 0;
    prologue &= ~(QL_EPI_SILU_GATE | QL_FUSED_STRICT);
    if (prologue == QL_PRO_ADDNORM && !ln_weight) return QL_ERR_NULL_POINTER;
    if (prologue != QL_PRO_SILU && prologue != QL_PRO_ADDNORM) return QL_ERR_UNSUPPORTED;
    if (gate && N % 4 != 0) return QL_ERR_BAD_SHAPE;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(N) || !fits_i32(K) || K % 32 != 0) return QL_ERR_BAD_SHAPE;
    if (!aligned(packed, 16) || !aligned(A, 16) || (delta && !aligned(delta, 16)) || (ln_weight && !aligned(ln_weight, 16)) ||
        (hout && !aligned(hout, 16)))
        return QL_ERR_MISALIGNED;
    return w4_packed_fused(dtype, prologue, gate, strict, A, packed, bias, C, N, K, delta, ln_weight, hout, eps, (hipStream_t)stream);
}

int qlinear_w4g32_fwd_tiled_gated(const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                                  int64_t lda, int64_t ldc, int dtype, void* stream) {
    if (!A || !tiled || !C) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || M <= 0 || N <= 0 || K <= 0 || K % 32 != 0 || N % 4 != 0 || lda < K ||
        ldc < N / 2)
        return QL_ERR_BAD_SHAPE;
    if (!aligned(tiled, 16) || !aligned(A, 16) || lda % 8 != 0) return QL_ERR_MISALIGNED;
    // prefill row counts: the 256 x 256-tile GEMM with the same epilogue (8-byte output chunks)
    if (N % 32 == 0 && w4_gemm256_supported(M, N, K, lda, A, 2) && ldc % 4 == 0 && aligned(C, 8))
        return w4_gemm256_gated(dtype, A, tiled, bias, C, M, N, K, lda, ldc, (hipStream_t)stream);
    // the epilogue lives in the few-row kernel without K slabs (wide first MLP projections, 3..32 rows)
    if (!w4_rows_use_gemm(M, N, K) || !w4_fewrow_supported(M, N, K) || w4_fewrow_workspace_bytes(M, N, K) != 0 || N % 32 != 0)
        return QL_ERR_UNSUPPORTED;
    return w4_fewrow(dtype, A, tiled, bias, C, M, N, K, lda, ldc, nullptr, 0, (hipStream_t)stream, true);
}

int qlinear_w4g32_fwd_tiled_residual(const void* A, const void* tiled, const void* bias, const void* residual, void* C, int64_t M,
                                     int64_t N, int64_t K, int64_t lda, int64_t ldc, int64_t ldr, int dtype, void* stream) {
    if (!A || !tiled || !C || !residual) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || K % 32 != 0 || lda < K || ldc < N || ldr < N) return QL_ERR_BAD_SHAPE;
    if (!aligned(tiled, 16) || !aligned(A, 16) || !aligned(C, 16) || !aligned(residual, 16) || lda % 8 != 0 || ldc % 8 != 0 || ldr % 8 != 0)
        return QL_ERR_MISALIGNED;
    if (N % 8 != 0 || !w4_gemm256_supported(M, N, K, lda, A, 2)) return QL_ERR_UNSUPPORTED;   // callers add the residual themselves
    return w4_gemm256_residual(dtype, A, tiled, bias, residual, C, M, N, K, lda, ldc, ldr, (hipStream_t)stream);
}

int qlinear_w4g32_fwd_packed_gated(const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                                   int64_t lda, int64_t ldc, int dtype, void* stream) {
    if (!packed) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (N <= 0 || K <= 0 || K % 32 != 0) return QL_ERR_BAD_SHAPE;
    if (w4_rows4_serves(dtype, M, N, K, lda, false)) {          // 2..4 rows: part 1 of the gate-interleaved copy, 4x4x4 MFMA
        if (!A || !C) return QL_ERR_NULL_POINTER;
        if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || N % 4 != 0 || lda < K || ldc < N / 2) return QL_ERR_BAD_SHAPE;
        if (!aligned(packed, 16) || !aligned(A, 16) || lda % 8 != 0) return QL_ERR_MISALIGNED;
        return w4_rows4_gated(dtype, A, packed, bias, C, M, N, K, lda, ldc, (hipStream_t)stream);
    }
    if (w4_rows16_serves(dtype, M, N, K, lda)) {                // 3..16 rows: part 1 of the gate-interleaved copy, 16x16x32 MFMA (w4_rows16.hip)
        if (!A || !C) return QL_ERR_NULL_POINTER;
        if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || N % 4 != 0 || lda < K || ldc < N / 2) return QL_ERR_BAD_SHAPE;
        if (!aligned(packed, 16) || !aligned(A, 16) || lda % 8 != 0) return QL_ERR_MISALIGNED;
        return w4_rows16(dtype, A, packed, bias, C, M, N, K, lda, ldc, (hipStream_t)stream, true);
    }
    return qlinear_w4g32_fwd_tiled_gated(A, (const char*)packed + w4_layout(N, K, 2).off_wm, bias, C, M, N, K, lda, ldc, dtype,
                                         stream);
}

int qlinear_w4g32_fwd_rows_fused(int prologue, const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N,
                                 int64_t K, const void* delta, const void* ln_weight, void* hout, float eps, int dtype, void* stream) {
    if (!A || !packed || !C || !ln_weight) return QL_ERR_NULL_POINTER;
    const bool gate = (prologue & QL_EPI_SILU_GATE) != 0;
    if ((prologue & ~QL_EPI_SILU_GATE) != QL_PRO_ADDNORM) return QL_ERR_UNSUPPORTED;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || M <= 0 || N <= 0 || K <= 0 || K % 32 != 0 || (gate && N % 4 != 0))
        return QL_ERR_BAD_SHAPE;
    if (!aligned(packed, 16) || !aligned(A, 16) || (delta && !aligned(delta, 16)) || !aligned(ln_weight, 16) || (hout && !aligned(hout, 16)))
        return QL_ERR_MISALIGNED;
    if (M < 2 || K > 8192 || !w4_rows4_serves(dtype, M, N, K, K, false)) return QL_ERR_UNSUPPORTED;
    return w4_rows4_fused(dtype, gate, A, packed, bias, C, M, N, K, delta, ln_weight, hout, eps, (hipStream_t)stream);
}

int qlinear_w4g32_fwd_packed_residual(const void* A, const void* packed, const void* bias, const void* residual, void* C,
                                      int64_t N, int64_t K, int dtype, int flags, void* stream) {
    if (!A || !packed || !C || !residual) return QL_ERR_NULL_POINTER;
    if (flags & ~QL_FLAG_STRICT_ROUNDING) return QL_ERR_UNSUPPORTED;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(N) || !fits_i32(K) || N <= 0 || K <= 0 || K % 32 != 0) return QL_ERR_BAD_SHAPE;
    if (!aligned(packed, 16) || !aligned(A, 16)) return QL_ERR_MISALIGNED;
    return w4_packed_residual(dtype, (flags & QL_FLAG_STRICT_ROUNDING) != 0, A, packed, bias, residual, C, N, K, (hipStream_t)stream);
}

int qlinear_w8_fwd_residual(const void* A, const int8_t* W, const void* S, const void* bias, const void* residual, void* C,
                            int64_t N, int64_t K, int64_t ldw, int dtype, void* stream) {
    if (!A || !W || !S || !C || !residual) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(N) || !fits_i32(K) || N <= 0 || K <= 0 || ldw < K) return QL_ERR_BAD_SHAPE;
    if (!aligned(A, 16) || !aligned(W, 16) || ldw % 16 != 0) return QL_ERR_MISALIGNED;
    return w8_gemv_residual(dtype, A, W, S, bias, residual, C, N, K, ldw, (hipStream_t)stream);
}

int qlinear_w8_fwd(const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t M, int64_t N,
                   int64_t K, int64_t ldw_k, int64_t ldw_n, int64_t lda, int64_t ldc, int dtype, int flags,
                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !W || !S || !C) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || lda < K || ldc < N || ldw_k <= 0 || ldw_n <= 0)
        return QL_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const bool k_contig = ldw_k == 1 && ldw_n >= K;
    if (k_contig && aligned(W, 16) && ldw_n % 16 == 0 && act_vec_ok(A, lda, dtype) && M > 4 && K % 16 == 0 &&
        (dtype == QL_DTYPE_F16 || dtype == QL_DTYPE_BF16))
        return w8_gemm(dtype, A, W, S, bias, C, M, N, K, ldw_n, lda, ldc, workspace, workspace_bytes, st);   // many rows: MFMA, reference rounding
    if (k_contig && aligned(W, 16) && ldw_n % 16 == 0 && act_vec_ok(A, lda, dtype) && dtype == QL_DTYPE_F32 && wq_gemm_f32_serves(M, N, K) &&
        !(dispatch_flags() & QL_D_NOF32MFMA))
        return w8_gemm_f32(A, W, S, bias, C, M, N, K, ldw_n, lda, ldc, st);   // fp32, 128+ rows: the fp32 matrix instruction (wq_gemm_f32.hip, round 5)
    if (k_contig && aligned(W, 16) && ldw_n % 16 == 0 && act_vec_ok(A, lda, dtype))
        return w8_gemv(dtype, A, W, S, bias, C, M, N, K, ldw_n, lda, ldc, (flags & QL_FLAG_STRICT_ROUNDING) != 0, st);
    return w8_generic(dtype, A, W, S, bias, C, M, N, K, ldw_k, ldw_n, lda, ldc, st);
}

size_t qlinear_w8_tiled_bytes(int64_t N, int64_t K) {
    if (N <= 0 || K <= 0 || K % 16 != 0) return 0;
    return w8_tiled_bytes(N, K);
}

int qlinear_w8_tile(const int8_t* W, void* tiled, int64_t N, int64_t K, int64_t ldw_n, void* stream) {
    if (!W || !tiled) return QL_ERR_NULL_POINTER;
    if (!fits_i32(N) || !fits_i32(K) || K % 16 != 0 || ldw_n < K) return QL_ERR_BAD_SHAPE;
    if (!aligned(W, 16) || ldw_n % 16 != 0 || !aligned(tiled, 16)) return QL_ERR_MISALIGNED;
    return w8_tile(W, (int8_t*)tiled, N, K, ldw_n, (hipStream_t)stream);
}

int qlinear_w8_fwd_tiled256(const void* A, const void* tiled, const void* S, const void* bias, void* C, int64_t M, int64_t N,
                            int64_t K, int64_t lda, int64_t ldc, int dtype, void* stream) {
    if (!A || !tiled || !S || !C) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || K % 16 != 0 || lda < K || ldc < N) return QL_ERR_BAD_SHAPE;
    if (!aligned(tiled, 16) || !act_vec_ok(A, lda, dtype)) return QL_ERR_MISALIGNED;
    if (!w4_gemm256_can_run(M, N, K, lda, A, esize(dtype))) return QL_ERR_UNSUPPORTED;
    return w8_gemm256(dtype, A, (const int8_t*)tiled, S, bias, C, M, N, K, lda, ldc, (hipStream_t)stream);
}

int qlinear_w8_fwd_tiled(const void* A, const void* tiled, const void* S, const void* bias, void* C, int64_t M, int64_t N,
                         int64_t K, int64_t lda, int64_t ldc, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !tiled || !S || !C) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || K % 16 != 0 || lda < K || ldc < N) return QL_ERR_BAD_SHAPE;
    if (!aligned(tiled, 16) || !act_vec_ok(A, lda, dtype)) return QL_ERR_MISALIGNED;
    return w8_fwd_tiled(dtype, A, (const int8_t*)tiled, S, bias, C, M, N, K, lda, ldc, workspace, workspace_bytes,
                        (hipStream_t)stream);
}

int qlinear_w8_fwd_tiled_gated(const void* A, const void* tiled, const void* S, const void* bias, void* C, int64_t M, int64_t N,
                               int64_t K, int64_t lda, int64_t ldc, int dtype, void* stream) {
    if (!A || !tiled || !S || !C) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || K % 16 != 0 || N % 4 != 0 || lda < K || ldc < N / 2) return QL_ERR_BAD_SHAPE;
    if (!aligned(tiled, 16) || !act_vec_ok(A, lda, dtype) || !aligned(C, 8) || ldc % 4 != 0) return QL_ERR_MISALIGNED;
    if (N % 32 != 0 || !w4_gemm256_supported(M, N, K, lda, A, 2)) return QL_ERR_UNSUPPORTED;   // projection + qlinear_silu_mul instead
    return w8_gemm256_gated(dtype, A, (const int8_t*)tiled, S, bias, C, M, N, K, lda, ldc, (hipStream_t)stream);
}

int qlinear_w8_fwd_tiled_residual(const void* A, const void* tiled, const void* S, const void* bias, const void* residual, void* C,
                                  int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc, int64_t ldr, int dtype, void* stream) {
    if (!A || !tiled || !S || !C || !residual) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || K % 16 != 0 || lda < K || ldc < N || ldr < N) return QL_ERR_BAD_SHAPE;
    if (!aligned(tiled, 16) || !act_vec_ok(A, lda, dtype) || !aligned(C, 16) || !aligned(residual, 16) || ldc % 8 != 0 || ldr % 8 != 0)
        return QL_ERR_MISALIGNED;
    if (N % 8 != 0 || !w4_gemm256_supported(M, N, K, lda, A, 2)) return QL_ERR_UNSUPPORTED;    // callers add the residual themselves
    return w8_gemm256_residual(dtype, A, (const int8_t*)tiled, S, bias, residual, C, M, N, K, lda, ldc, ldr, (hipStream_t)stream);
}

int qlinear_w8_fwd_fused(int prologue, const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t N,
                         int64_t K, int64_t ldw_n, const void* delta, const void* ln_weight, void* hout, float eps, int dtype,
                         void* stream) {
    if (!A || !W || !S || !C || !ln_weight) return QL_ERR_NULL_POINTER;
    const bool gate = (prologue & QL_EPI_SILU_GATE) != 0;
    if ((prologue & ~QL_EPI_SILU_GATE) != QL_PRO_ADDNORM) return QL_ERR_UNSUPPORTED;
    if (dtype != QL_DTYPE_F16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(N) || !fits_i32(K) || K % 16 != 0 || ldw_n < K || (gate && N % 4 != 0)) return QL_ERR_BAD_SHAPE;
    if (!aligned(W, 16) || ldw_n % 16 != 0 || !aligned(A, 16) || (delta && !aligned(delta, 16)) || !aligned(ln_weight, 16) ||
        (hout && !aligned(hout, 16)))
        return QL_ERR_MISALIGNED;
    return w8_gemv_fused(dtype, gate, A, W, S, bias, C, N, K, ldw_n, delta, ln_weight, hout, eps, (hipStream_t)stream);
}

int qlinear_w8_bwd_input(const void* Gout, const int8_t* Wkn, const void* S, void* dA, int64_t M, int64_t N, int64_t K,
                         int64_t ldg, int64_t ldda, int dtype, void* stream) {
    if (!Gout || !Wkn || !S || !dA) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || ldg < N || ldda < K) return QL_ERR_BAD_SHAPE;
    if (N % 16 != 0 || N < 16) return QL_ERR_UNSUPPORTED;                  /* 16-byte loads along the contraction */
    if (!aligned(Wkn, 16) || !aligned(S, 16) || !act_vec_ok(Gout, ldg, dtype)) return QL_ERR_MISALIGNED;
    /* the forward GEMM with the roles renamed: rows = K outputs, contraction = N, scale on the contraction */
    return w8_gemm_scale_k(dtype, Gout, Wkn, S, dA, M, K, N, N, ldg, ldda, (hipStream_t)stream);
}

int qlinear_act_quant_i8_rowwise(const void* A, int8_t* Aq, float* a_scale, int64_t M, int64_t K, int64_t lda,
                                 int dtype, void* stream) {
    if (!A || !Aq || !a_scale) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(K) || lda < K) return QL_ERR_BAD_SHAPE;
    return act_quant_rowwise(dtype, A, Aq, a_scale, M, K, lda, false, (hipStream_t)stream);
}

int qlinear_act_quant_i8(const void* A, int8_t* Aq, float* a_scale, int64_t M, int64_t K, int64_t lda, int dtype, int flags,
                         void* stream) {
    if (!A || !Aq || !a_scale) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(K) || lda < K) return QL_ERR_BAD_SHAPE;
    if (flags & ~QL_FLAG_ACT_PER_TENSOR) return QL_ERR_UNSUPPORTED;
    return act_quant_rowwise(dtype, A, Aq, a_scale, M, K, lda, (flags & QL_FLAG_ACT_PER_TENSOR) != 0, (hipStream_t)stream);
}

int qlinear_w8a8_fwd_tiled256(const int8_t* Aq, const float* a_scale, const void* tiled, const void* S, const void* bias, void* C,
                              int64_t M, int64_t N, int64_t K, int64_t ldc, int dtype, void* stream) {
    if (!Aq || !a_scale || !tiled || !S || !C) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || ldc < N) return QL_ERR_BAD_SHAPE;
    if (!aligned(Aq, 16) || !aligned(tiled, 16)) return QL_ERR_MISALIGNED;
    if (!w8a8_gemm256_can_run(dtype, M, N, K, Aq)) return QL_ERR_UNSUPPORTED;
    return w8a8_gemm256(dtype, Aq, a_scale, (const int8_t*)tiled, S, bias, C, M, N, K, ldc, (hipStream_t)stream);
}

int qlinear_w8a8_fwd_tiled(const int8_t* Aq, const float* a_scale, const void* tiled, const void* S, const void* bias, void* C,
                           int64_t M, int64_t N, int64_t K, int64_t ldc, int dtype, void* stream) {
    if (!Aq || !a_scale || !tiled || !S || !C) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || ldc < N) return QL_ERR_BAD_SHAPE;
    if (K % 16 != 0) return QL_ERR_UNSUPPORTED;     /* rows are read in 16-byte units */
    if (!aligned(Aq, 16) || !aligned(tiled, 16)) return QL_ERR_MISALIGNED;
    return w8a8_gemm_tiled(dtype, Aq, a_scale, (const int8_t*)tiled, S, bias, C, M, N, K, ldc, (hipStream_t)stream);
}

int qlinear_w8a8_fwd_tiled_gated(const int8_t* Aq, const float* a_scale, const void* tiled, const void* S, const void* bias, void* C,
                                 int64_t M, int64_t N, int64_t K, int64_t ldc, int dtype, void* stream) {
    if (!Aq || !a_scale || !tiled || !S || !C) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || N % 4 != 0 || ldc < N / 2) return QL_ERR_BAD_SHAPE;
    if (!aligned(Aq, 16) || !aligned(tiled, 16) || !aligned(C, 8) || ldc % 4 != 0) return QL_ERR_MISALIGNED;
    if (N % 32 != 0 || !w8a8_gemm256_supported(dtype, M, N, K, Aq)) return QL_ERR_UNSUPPORTED;   /* projection + qlinear_silu_mul(_quant_i8) instead */
    return w8a8_gemm256_gated(dtype, Aq, a_scale, (const int8_t*)tiled, S, bias, C, M, N, K, ldc, (hipStream_t)stream);
}

int qlinear_w8a8_linear_tiled(const void* A, const void* tiled, const void* S, const void* bias, void* C, int64_t M, int64_t N,
                              int64_t K, int64_t lda, int64_t ldc, int dtype, int flags, void* workspace, size_t workspace_bytes,
                              void* stream) {
    if (!A || !tiled || !S || !C) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || lda < K || ldc < N) return QL_ERR_BAD_SHAPE;
    if (K % 16 != 0 || (flags & ~QL_FLAG_ACT_PER_TENSOR)) return QL_ERR_UNSUPPORTED;
    if (!workspace || !aligned(workspace, 16) || workspace_bytes < qlinear_workspace_bytes(QL_OP_W8A8_LINEAR_TILED, M, N, K, 0))
        return QL_ERR_WORKSPACE;
    if (!aligned(tiled, 16)) return QL_ERR_MISALIGNED;
    int8_t* Aq = (int8_t*)workspace;
    float* a_scale = (float*)((char*)workspace + (((size_t)M * (size_t)K + 15) & ~(size_t)15));
    const int rc = act_quant_rowwise(dtype, A, Aq, a_scale, M, K, lda, (flags & QL_FLAG_ACT_PER_TENSOR) != 0, (hipStream_t)stream);
    if (rc) return rc;
    return w8a8_gemm_tiled(dtype, Aq, a_scale, (const int8_t*)tiled, S, bias, C, M, N, K, ldc, (hipStream_t)stream);
}

int qlinear_w8a8_fwd(const int8_t* Aq, const float* a_scale, const int8_t* W, const void* S, const void* bias,
                     void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, int dtype, void* workspace,
                     size_t workspace_bytes, void* stream) {
    if (!Aq || !a_scale || !W || !S || !C) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(M) || !fits_i32(N) || !fits_i32(K) || ldc < N) return QL_ERR_BAD_SHAPE;
    if (K % 16 != 0) return QL_ERR_UNSUPPORTED;     /* rows are read in 16-byte units */
    if (!aligned(Aq, 16) || !aligned(W, 16)) return QL_ERR_MISALIGNED;
    return w8a8_gemm(dtype, Aq, a_scale, W, S, bias, C, M, N, K, ldc, workspace, workspace_bytes, (hipStream_t)stream);
}

int qlinear_qembedding_w4(const int64_t* ids, const uint8_t* Wq, const void* S, void* out, int64_t count, int64_t V,
                          int64_t D, int64_t group, int dtype, void* stream) {
    if (!ids || !Wq || !S || !out) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(count) || !fits_i32(V) || !fits_i32(D) || (V & 1)) return QL_ERR_BAD_SHAPE;
    if (group <= 0 || V % group != 0) return QL_ERR_BAD_GROUP;
    return qembedding_w4(dtype, ids, Wq, S, out, count, V, D, group, (hipStream_t)stream);
}

int qlinear_qembedding_w8(const int64_t* ids, const int8_t* W, const void* S, void* out, int64_t count, int64_t V,
                          int64_t D, int dtype, void* stream) {
    if (!ids || !W || !S || !out) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(count) || !fits_i32(V) || !fits_i32(D)) return QL_ERR_BAD_SHAPE;
    return qembedding_w8(dtype, ids, W, S, out, count, V, D, (hipStream_t)stream);
}

int qlinear_rmsnorm(const void* X, const void* W, void* Out, int64_t rows, int64_t dim, int64_t ldx, int64_t ldo,
                    float eps, int dtype, void* stream) {
    if (!X || !W || !Out) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(rows) || !fits_i32(dim) || ldx < dim || ldo < dim || dim % 8 != 0) return QL_ERR_BAD_SHAPE;
    if (!aligned(X, 16) || !aligned(W, 16) || !aligned(Out, 16) || (ldx * esize(dtype)) % 16 || (ldo * esize(dtype)) % 16)
        return QL_ERR_MISALIGNED;
    return rmsnorm(dtype, X, nullptr, W, nullptr, Out, rows, dim, ldx, ldo, eps, (hipStream_t)stream);
}

int qlinear_add_rmsnorm(const void* X, const void* Delta, const void* W, void* Hout, void* Out, int64_t rows,
                        int64_t dim, int64_t ld, float eps, int dtype, void* stream) {
    if (!X || !Delta || !W || !Hout || !Out) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(rows) || !fits_i32(dim) || ld < dim || dim % 8 != 0) return QL_ERR_BAD_SHAPE;
    if (!aligned(X, 16) || !aligned(Delta, 16) || !aligned(W, 16) || !aligned(Hout, 16) || !aligned(Out, 16) ||
        (ld * esize(dtype)) % 16)
        return QL_ERR_MISALIGNED;
    return rmsnorm(dtype, X, Delta, W, Hout, Out, rows, dim, ld, ld, eps, (hipStream_t)stream);
}

int qlinear_rmsnorm_quant_i8(const void* X, const void* Delta, const void* W, void* Hout, void* Out, int8_t* Aq, float* a_scale,
                             int64_t rows, int64_t dim, int64_t ld, float eps, int dtype, void* stream) {
    if (!X || !W || !Aq || !a_scale || (Delta && !Hout)) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(rows) || !fits_i32(dim) || ld < dim || dim % 8 != 0) return QL_ERR_BAD_SHAPE;
    if (!aligned(X, 16) || !aligned(W, 16) || !aligned(Aq, 16) || (Delta && (!aligned(Delta, 16) || !aligned(Hout, 16))) ||
        (Out && !aligned(Out, 16)) || (ld * esize(dtype)) % 16)
        return QL_ERR_MISALIGNED;
    return rmsnorm_quant(dtype, X, Delta, W, Hout, Out, Aq, a_scale, rows, dim, ld, ld, eps, (hipStream_t)stream);
}

int qlinear_silu_mul_quant_i8(const void* In, void* Out, int8_t* Aq, float* a_scale, int64_t rows, int64_t hidden, int64_t ldin,
                              int64_t ldo, int dtype, void* stream) {
    if (!In || !Aq || !a_scale) return QL_ERR_NULL_POINTER;
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(rows) || !fits_i32(hidden) || ldin < 2 * hidden || (Out && ldo < hidden) || hidden % 8 != 0) return QL_ERR_BAD_SHAPE;
    if (!aligned(In, 16) || !aligned(Aq, 8) || (Out && !aligned(Out, 16)) || (ldin * esize(dtype)) % 16 ||
        (Out && (ldo * esize(dtype)) % 16) || (hidden * esize(dtype)) % 16)
        return QL_ERR_MISALIGNED;
    return silu_mul_quant(dtype, In, Out, Aq, a_scale, rows, hidden, ldin, ldo, (hipStream_t)stream);
}

int qlinear_rope_kv_write(const void* QKV, const void* table, const int64_t* pos, const int64_t* widx, void* Qout,
                          void* Kcache, void* Vcache, int64_t B, int64_t S, int64_t H, int64_t G, int64_t D,
                          int64_t capacity, int64_t ldqkv, int dtype, void* stream) {
    if (!QKV || !table || !pos || !widx || !Qout || !Kcache || !Vcache) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(B) || !fits_i32(S) || !fits_i32(H) || !fits_i32(G) || !fits_i32(D) || !fits_i32(capacity) ||
        (D % 8) || H % G != 0 || ldqkv < (H + 2 * G) * D)
        return QL_ERR_BAD_SHAPE;
    if (!aligned(QKV, 16) || !aligned(table, 16) || !aligned(Qout, 16) || !aligned(Kcache, 16) || !aligned(Vcache, 16) ||
        (ldqkv * esize(dtype)) % 16)
        return QL_ERR_MISALIGNED;
    return rope_kv_write(dtype, QKV, table, pos, widx, Qout, Kcache, Vcache, B, S, H, G, D, capacity, ldqkv,
                         (hipStream_t)stream);
}

int qlinear_decode_attention(const void* Q, const void* Kcache, const void* Vcache, const float* mask, void* Out,
                             int64_t B, int64_t H, int64_t G, int64_t D, int64_t capacity, int dtype, void* stream) {
    if (!Q || !Kcache || !Vcache || !mask || !Out) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(B) || !fits_i32(H) || !fits_i32(G) || !fits_i32(D) || !fits_i32(capacity) || H % G != 0)
        return QL_ERR_BAD_SHAPE;
    if ((D != 128 && D != 64 && D != 32) || (capacity + 35 * D + 8) * 4 > 64 * 1024) return QL_ERR_UNSUPPORTED;
    if (!aligned(Kcache, 16) || !aligned(Vcache, 16)) return QL_ERR_MISALIGNED;
    return decode_attention(dtype, Q, Kcache, Vcache, mask, Out, B, H, G, D, capacity, (hipStream_t)stream);
}

size_t qlinear_decode_attention_split_bytes(int64_t B, int64_t H, int64_t D, int64_t capacity) {
    if (B <= 0 || H <= 0 || D <= 0 || capacity <= 0) return 0;
    return decode_attention_split_bytes(B, H, D, capacity);
}

static int attention_rope_checked(const void* QKV, const void* table, const int64_t* pos, const int64_t* widx, void* Kcache,
                                  void* Vcache, const float* mask, void* Out, int64_t B, int64_t H, int64_t G, int64_t D,
                                  int64_t capacity, int64_t ldqkv, int dtype, void* split_workspace,
                                  size_t split_workspace_bytes, const Prefetch& pf, void* stream) {
    if (!QKV || !table || !pos || !widx || !Kcache || !Vcache || !mask || !Out) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(B) || !fits_i32(H) || !fits_i32(G) || !fits_i32(D) || !fits_i32(capacity) || H % G != 0 ||
        ldqkv < (H + 2 * G) * D)
        return QL_ERR_BAD_SHAPE;
    const bool split = split_workspace != nullptr;
    if (split && (split_workspace_bytes < decode_attention_split_bytes(B, H, D, capacity) || !aligned(split_workspace, 4)))
        return QL_ERR_WORKSPACE;
    if ((D != 128 && D != 64 && D != 32) || (!split && (capacity + 35 * D + 8) * 4 > 64 * 1024)) return QL_ERR_UNSUPPORTED;
    if (!aligned(Kcache, 16) || !aligned(Vcache, 16)) return QL_ERR_MISALIGNED;
    return decode_attention_rope(dtype, QKV, table, pos, widx, Kcache, Vcache, mask, Out, B, H, G, D, capacity, ldqkv,
                                 (float*)split_workspace, pf, (hipStream_t)stream);
}

int qlinear_decode_attention_rope(const void* QKV, const void* table, const int64_t* pos, const int64_t* widx, void* Kcache,
                                  void* Vcache, const float* mask, void* Out, int64_t B, int64_t H, int64_t G, int64_t D,
                                  int64_t capacity, int64_t ldqkv, int dtype, void* split_workspace,
                                  size_t split_workspace_bytes, void* stream) {
    return attention_rope_checked(QKV, table, pos, widx, Kcache, Vcache, mask, Out, B, H, G, D, capacity, ldqkv, dtype,
                                  split_workspace, split_workspace_bytes, Prefetch{}, stream);
}

int qlinear_decode_attention_rope_prefetch(const void* QKV, const void* table, const int64_t* pos, const int64_t* widx,
                                           void* Kcache, void* Vcache, const float* mask, void* Out, int64_t B, int64_t H,
                                           int64_t G, int64_t D, int64_t capacity, int64_t ldqkv, int dtype,
                                           void* split_workspace, size_t split_workspace_bytes, const void* next_weights,
                                           int next_kind, int64_t next_N, int64_t next_K, void* stream) {
    Prefetch pf{};
    if (next_weights) {
        if (next_N <= 0 || next_K <= 0 || !aligned(next_weights, 16)) return QL_ERR_BAD_SHAPE;
        int64_t wb = 0, sb = 0, soff = 0, blocks = 0;
        if (next_kind == QL_NEXT_W4G32_PACKED) {
            if (next_K % 32 != 0) return QL_ERR_BAD_SHAPE;
            w4_gemv_blocks(next_N, next_K, &wb, &sb, &soff, &blocks);
            pf = Prefetch{{(const char*)next_weights, (const char*)next_weights + soff}, {wb, sb}, (int)blocks};
            // the last workgroup's range may pass the end of a ragged matrix: leave it out
            if (blocks * wb > soff) pf.blocks = (int)(soff / wb);
        } else if (next_kind == QL_NEXT_W8_ROWS) {
            if (next_K % 16 != 0) return QL_ERR_BAD_SHAPE;
            w8_gemv_blocks(next_N, next_K, next_K, &wb, &blocks);
            pf = Prefetch{{(const char*)next_weights, nullptr}, {wb, 0}, (int)(blocks * wb > next_N * next_K ? next_N * next_K / wb : blocks)};
        } else {
            return QL_ERR_UNSUPPORTED;
        }
    }
    return attention_rope_checked(QKV, table, pos, widx, Kcache, Vcache, mask, Out, B, H, G, D, capacity, ldqkv, dtype,
                                  split_workspace, split_workspace_bytes, pf, stream);
}

int qlinear_masked_softmax(const void* scores, const float* mask, void* P, int64_t rows, int64_t T, int64_t mask_rows,
                           int64_t lds, int64_t ldm, int64_t ldp, int dtype, void* stream) {
    if (!scores || !P) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (rows < 0 || !fits_i32(T) || T <= 0 || lds < T || ldp < T || (mask && (mask_rows <= 0 || ldm < T)) ||
        rows > ((int64_t)1 << 33))
        return QL_ERR_BAD_SHAPE;
    if (rows == 0) return 0;
    return masked_softmax(dtype, scores, mask, P, rows, T, mask ? mask_rows : 1, lds, ldm, ldp, (hipStream_t)stream);
}

int qlinear_prefill_attention_tiles(int64_t* q_block, int64_t* k_tile) {
    if (!q_block || !k_tile) return QL_ERR_NULL_POINTER;
    prefill_attention_tiles(q_block, k_tile);
    return 0;
}

int qlinear_prefill_attention(const void* Q, const void* Kcache, const void* Vcache, const float* mask, const uint8_t* tile_flags,
                              void* Out, int64_t B, int64_t S, int64_t T, int64_t H, int64_t G, int64_t D, int64_t capacity,
                              int64_t ldm, int dtype, void* stream) {
    if (!Q || !Kcache || !Vcache || !Out) return QL_ERR_NULL_POINTER;
    if (tile_flags && !mask) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (B < 0 || !fits_i32(S) || !fits_i32(T) || !fits_i32(H) || !fits_i32(G) || !fits_i32(capacity) || T > capacity ||
        (mask && ldm < T) || B * G * ((S + 15) / 16) >= ((int64_t)1 << 31) || capacity * G * D >= ((int64_t)1 << 31))
        return QL_ERR_BAD_SHAPE;
    if (D != 128 || H != 16 * G || dtype == QL_DTYPE_F32) return QL_ERR_UNSUPPORTED;
    if ((((uintptr_t)Q | (uintptr_t)Kcache | (uintptr_t)Vcache | (uintptr_t)Out) & 15) != 0) return QL_ERR_MISALIGNED;
    if (B == 0) return 0;
    return prefill_attention(dtype, Q, Kcache, Vcache, mask, tile_flags, Out, B, S, T, H, G, capacity, ldm, (hipStream_t)stream);
}

int qlinear_silu_mul(const void* In, void* Out, int64_t rows, int64_t hidden, int64_t ldin, int64_t ldo, int dtype,
                     void* stream) {
    if (!In || !Out) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(rows) || !fits_i32(hidden) || ldin < 2 * hidden || ldo < hidden || hidden % 8 != 0) return QL_ERR_BAD_SHAPE;
    if (!aligned(In, 16) || !aligned(Out, 16) || (ldin * esize(dtype)) % 16 || (ldo * esize(dtype)) % 16 ||
        (hidden * esize(dtype)) % 16)
        return QL_ERR_MISALIGNED;
    return silu_mul(dtype, In, Out, rows, hidden, ldin, ldo, (hipStream_t)stream);
}

int qlinear_greedy_advance(const void* logits, int64_t B, int64_t N, int64_t ldl, int64_t* tok, int64_t* write_index,
                           int64_t* pos, float* mask, int64_t capacity, int dtype, void* stream) {
    if (!logits || !tok || !write_index || !pos || !mask) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(B) || !fits_i32(N) || !fits_i32(capacity) || ldl < N) return QL_ERR_BAD_SHAPE;
    return greedy_advance(dtype, logits, B, N, ldl, tok, write_index, pos, mask, capacity, (hipStream_t)stream);
}

int qlinear_top_p_sample(const void* logits, int64_t B, int64_t N, int64_t ldl, int64_t top_k, float top_p, float temperature,
                         const float* dev_params, uint64_t* rng_state, int64_t* tok, int64_t* write_index, int64_t* pos, float* mask,
                         int64_t capacity, float* probs_out, int64_t* index_out, float* u_out, int64_t out_ld, int dtype, void* stream) {
    if (!logits || !tok) return QL_ERR_NULL_POINTER;
    if (!dtype_ok(dtype)) return QL_ERR_BAD_DTYPE;
    if (!fits_i32(B) || !fits_i32(N) || ldl < N) return QL_ERR_BAD_SHAPE;
    const bool book = write_index || pos || mask;
    if (book && !(write_index && pos && mask)) return QL_ERR_NULL_POINTER;
    if (book && !fits_i32(capacity)) return QL_ERR_BAD_SHAPE;
    if ((probs_out == nullptr) != (index_out == nullptr)) return QL_ERR_NULL_POINTER;
    if (probs_out && (out_ld < (top_k < N ? top_k : N) || !fits_i32(out_ld))) return QL_ERR_BAD_SHAPE;
    return top_p_sample(dtype, logits, B, N, ldl, top_k, top_p, temperature, dev_params, rng_state, tok, write_index, pos, mask,
                        capacity, probs_out, index_out, u_out, out_ld, (hipStream_t)stream);
}

}  // extern "C"
