// int8 per-output-channel quantised linear forward kernels for gfx950 (MI355X).
//
// Replaces _dynamic_quant_matmul_kernel (chatglm_q/int8/triton_ops.py:13-84) and adds the
// int8-activation path whose semantic the reference only states in its ONNX symbolic
// (chatglm_q/int8/qlinear.py:56-70) and quantiser (chatglm_q/int8/quantizer.py:11-19).
//
//   w8_generic_kernel        any strides (incl. the (K, N)-contiguous form the reference test uses)
//   w8_gemv_kernel           module layout: W (N, K) row-major, K contiguous; one wave owns 4 output
//                            channels and streams their rows with 16-byte loads (decode shapes)
//   act_quant_rowwise_kernel row-wise symmetric int8 activation quantisation (fp32 arithmetic)
//   w8a8_mfma_kernel         i8 x i8 -> i32 on v_mfma_i32_32x32x32_i8, rank-1 scale epilogue
#include "launch.h"
#include "ql_common.h"

namespace ql {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// =============================================================================================
// generic weight-only: one thread per (m, n), arbitrary weight strides
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void w8_generic_kernel(const T* __restrict__ A, const int8_t* __restrict__ W,
                                                         const T* __restrict__ S, const T* __restrict__ bias,
                                                         T* __restrict__ C, int M, int N, int K, int64_t ldw_k,
                                                         int64_t ldw_n, int64_t lda, int64_t ldc) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.y;
    if (n >= N || m >= M) return;
    const T* a = A + (int64_t)m * lda;
    const int8_t* w = W + (int64_t)n * ldw_n;
    const float s = Act<T>::load(S + n);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const float wq = Act<T>::round((float)w[(int64_t)k * ldw_k] * s);   // b * scale rounded to act dtype
        acc = __builtin_fmaf(Act<T>::load(a + k), wq, acc);
    }
    store_out<T>(C + (int64_t)m * ldc + n, acc, bias ? bias + n : nullptr);
}

// =============================================================================================
// module layout GEMV: wave = 4 output channels, lanes stride K in 16-byte units
// =============================================================================================
// fp16: bytes -> half2 without cvt: b ^ 0x80 = b + 128 (unsigned); 0x6400 | u8 = 1024 + u8; minus 1152
// gives b exactly.  One v_and_or / v_lshr+v_and_or per byte PAIR: (b0, b2) and (b1, b3).
struct BytePairs {
    h2 p02, p13;
};
__device__ __forceinline__ BytePairs byte_pairs_f16(u32 w) {
    const h2 k1152 = {(f16)1152.0f, (f16)1152.0f};
    const u32 t = w ^ 0x80808080u;
    BytePairs r;
    r.p02 = as_h2((t & 0x00FF00FFu) | 0x64006400u) - k1152;
    r.p13 = as_h2(((t >> 8) & 0x00FF00FFu) | 0x64006400u) - k1152;
    return r;
}

template <typename T, int MB>
__global__ __launch_bounds__(256) void w8_gemv_kernel(const T* __restrict__ A, const int8_t* __restrict__ W,
                                                      const T* __restrict__ S, const T* __restrict__ bias,
                                                      T* __restrict__ C, int M, int N, int K, int64_t ldw,
                                                      int64_t lda, int64_t ldc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = (blockIdx.x * 4 + wave) * 4;   // first of the wave's 4 output channels
    if (nb >= N) return;                          // waves are independent: no barriers below
    const int m0 = blockIdx.y * MB;

    float acc[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[m][c] = 0.f;

    const int8_t* wrow[4];
    float sc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = (nb + c < N) ? (nb + c) : (N - 1);
        wrow[c] = W + (int64_t)n * ldw;
        sc[c] = Act<T>::load(S + n);
    }
    const T* arow[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) arow[m] = A + (int64_t)((m0 + m < M) ? (m0 + m) : (M - 1)) * lda;

    const int kvec = K & ~15;
#pragma unroll 2
    for (int k = lane * 16; k < kvec; k += 64 * 16) {
        u32x4 w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[c] + k));

        if constexpr (Act<T>::code == QL_DTYPE_F16) {
            // activations regrouped to match the byte pairs: (a0,a2),(a1,a3) per 4 k
            h2 a02[MB][4], a13[MB][4];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32x4 x = *reinterpret_cast<const u32x4*>(arow[m] + k + 8 * j);
                    a02[m][2 * j + 0] = as_h2((x[0] & 0xFFFFu) | (x[1] << 16));
                    a13[m][2 * j + 0] = as_h2((x[0] >> 16) | (x[1] & 0xFFFF0000u));
                    a02[m][2 * j + 1] = as_h2((x[2] & 0xFFFFu) | (x[3] << 16));
                    a13[m][2 * j + 1] = as_h2((x[2] >> 16) | (x[3] & 0xFFFF0000u));
                }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16 sh = (f16)sc[c];
                const h2 s2 = {sh, sh};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const BytePairs b = byte_pairs_f16(w[c][j]);
                    const h2 w02 = b.p02 * s2, w13 = b.p13 * s2;   // rounded to fp16 (faithful)
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        float v = acc[m][c];
                        v = __builtin_amdgcn_fdot2(w02, a02[m][j], v, false);
                        v = __builtin_amdgcn_fdot2(w13, a13[m][j], v, false);
                        acc[m][c] = v;
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    float a[MB];
#pragma unroll
                    for (int m = 0; m < MB; ++m) a[m] = Act<T>::load(arow[m] + k + 4 * j + b);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int q = (int)(w[c][j] << (24 - 8 * b)) >> 24;   // sign-extended byte b
                        const float wq = Act<T>::round((float)q * sc[c]);
#pragma unroll
                        for (int m = 0; m < MB; ++m) acc[m][c] = __builtin_fmaf(a[m], wq, acc[m][c]);
                    }
                }
        }
    }
    // K tail (K % 16): one element per lane
    for (int k = kvec + lane; k < K; k += 64) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float wq = Act<T>::round((float)wrow[c][k] * sc[c]);
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[m][c] = __builtin_fmaf(Act<T>::load(arow[m] + k), wq, acc[m][c]);
        }
    }

#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[m][c] = wave_sum(acc[m][c]);

    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m0 + m >= M) break;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int n = nb + c;
                if (n < N) store_out<T>(C + (int64_t)(m0 + m) * ldc + n, acc[m][c], bias ? bias + n : nullptr);
            }
        }
    }
}

// =============================================================================================
// row-wise int8 activation quantisation (quantize_int8, chatglm_q/int8/quantizer.py:11-19),
// evaluated in fp32: true divisions and round-half-even so the integers match the oracle exactly.
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void act_quant_rowwise_kernel(const T* __restrict__ A, int8_t* __restrict__ Aq,
                                                                float* __restrict__ a_scale, int K, int64_t lda) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const T* a = A + (int64_t)m * lda;
    float mx = 0.f;
    for (int k = tid; k < K; k += 256) mx = fmaxf(mx, fabsf(Act<T>::load(a + k)));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = mx / 127.0f;
    s = fmaxf(s, 1e-10f);
    if (tid == 0) a_scale[m] = s;
    int8_t* q = Aq + (int64_t)m * K;
    for (int k = tid; k < K; k += 256) {
        float v = rintf(Act<T>::load(a + k) / s);
        v = fminf(fmaxf(v, -127.f), 127.f);
        q[k] = (int8_t)v;
    }
}

// =============================================================================================
// W8A8: C = round(acc_i32 * (a_scale[m] * w_scale[n])) (+ bias), acc = Aq (M,K) . W (N,K)^T
//   v_mfma_i32_32x32x32_i8: lane l supplies row (l & 31) of its operand and the 16 consecutive k
//   bytes of half (l >> 5) of the 32-deep step - both operands are K-contiguous in memory, so each
//   fragment is ONE 16-byte load.  Block = 4 waves = 64 x 64 outputs (wave = 32 x 32).
//   Accumulator map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void w8a8_mfma_kernel(const int8_t* __restrict__ Aq, const float* __restrict__ a_scale,
                                                        const int8_t* __restrict__ W, const T* __restrict__ S,
                                                        const T* __restrict__ bias, T* __restrict__ C, int M, int N,
                                                        int K, int64_t ldc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mt = blockIdx.y * 64 + (wave >> 1) * 32;
    const int nt = blockIdx.x * 64 + (wave & 1) * 32;
    if (mt >= M || nt >= N) return;
    const int r = lane & 31, kh = lane >> 5;
    const int am = (mt + r < M) ? (mt + r) : (M - 1);
    const int bn = (nt + r < N) ? (nt + r) : (N - 1);
    const int8_t* ap = Aq + (int64_t)am * K + kh * 16;
    const int8_t* bp = W + (int64_t)bn * K + kh * 16;

    i32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;

#pragma unroll 4
    for (int k = 0; k < K; k += 32) {
        const i32x4 a = *reinterpret_cast<const i32x4*>(ap + k);
        const i32x4 b = *reinterpret_cast<const i32x4*>(bp + k);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
    }

    const int n = nt + r;
    if (n < N) {
        const float ws = Act<T>::load(S + n);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = mt + (i & 3) + 8 * (i >> 2) + 4 * kh;
            if (m < M) {
                const float comb = a_scale[m] * ws;
                store_out<T>(C + (int64_t)m * ldc + n, (float)acc[i] * comb, bias ? bias + n : nullptr);
            }
        }
    }
}

// =============================================================================================
// launchers
// =============================================================================================
template <typename T>
static int launch_w8_generic(const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t M,
                             int64_t N, int64_t K, int64_t ldw_k, int64_t ldw_n, int64_t lda, int64_t ldc,
                             hipStream_t st) {
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)M);
    w8_generic_kernel<T><<<grid, 256, 0, st>>>((const T*)A, W, (const T*)S, (const T*)bias, (T*)C, (int)M, (int)N,
                                               (int)K, ldw_k, ldw_n, lda, ldc);
    return finish_launch();
}

template <typename T, int MB>
static int launch_w8_gemv_mb(const T* A, const int8_t* W, const T* S, const T* bias, T* C, int M, int N, int K,
                             int64_t ldw, int64_t lda, int64_t ldc, hipStream_t st) {
    const int quads = (N + 3) / 4;
    dim3 grid((unsigned)((quads + 3) / 4), (unsigned)((M + MB - 1) / MB));
    w8_gemv_kernel<T, MB><<<grid, 256, 0, st>>>(A, W, S, bias, C, M, N, K, ldw, lda, ldc);
    return finish_launch();
}

template <typename T>
static int launch_w8_gemv(const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t M,
                          int64_t N, int64_t K, int64_t ldw, int64_t lda, int64_t ldc, hipStream_t st) {
    if (M == 1)
        return launch_w8_gemv_mb<T, 1>((const T*)A, W, (const T*)S, (const T*)bias, (T*)C, 1, (int)N, (int)K, ldw, lda, ldc, st);
    if (M == 2)
        return launch_w8_gemv_mb<T, 2>((const T*)A, W, (const T*)S, (const T*)bias, (T*)C, 2, (int)N, (int)K, ldw, lda, ldc, st);
    return launch_w8_gemv_mb<T, 4>((const T*)A, W, (const T*)S, (const T*)bias, (T*)C, (int)M, (int)N, (int)K, ldw, lda, ldc, st);
}

template <typename T>
static int launch_act_quant(const void* A, int8_t* Aq, float* a_scale, int64_t M, int64_t K, int64_t lda,
                            hipStream_t st) {
    act_quant_rowwise_kernel<T><<<(unsigned)M, 256, 0, st>>>((const T*)A, Aq, a_scale, (int)K, lda);
    return finish_launch();
}

template <typename T>
static int launch_w8a8(const int8_t* Aq, const float* a_scale, const int8_t* W, const void* S, const void* bias,
                       void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, hipStream_t st) {
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    w8a8_mfma_kernel<T><<<grid, 256, 0, st>>>(Aq, a_scale, W, (const T*)S, (const T*)bias, (T*)C, (int)M, (int)N,
                                              (int)K, ldc);
    return finish_launch();
}

#define QL_DISPATCH_DTYPE(dtype, fn, ...)                         \
    switch (dtype) {                                              \
    case QL_DTYPE_F32: return fn<float>(__VA_ARGS__);             \
    case QL_DTYPE_F16: return fn<f16>(__VA_ARGS__);               \
    case QL_DTYPE_BF16: return fn<__bf16>(__VA_ARGS__);           \
    default: return QL_ERR_BAD_DTYPE;                             \
    }

int w8_generic(int dtype, const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t M,
               int64_t N, int64_t K, int64_t ldw_k, int64_t ldw_n, int64_t lda, int64_t ldc, hipStream_t st) {
    QL_DISPATCH_DTYPE(dtype, launch_w8_generic, A, W, S, bias, C, M, N, K, ldw_k, ldw_n, lda, ldc, st)
}
int w8_gemv(int dtype, const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t M,
            int64_t N, int64_t K, int64_t ldw, int64_t lda, int64_t ldc, hipStream_t st) {
    QL_DISPATCH_DTYPE(dtype, launch_w8_gemv, A, W, S, bias, C, M, N, K, ldw, lda, ldc, st)
}
int act_quant_rowwise(int dtype, const void* A, int8_t* Aq, float* a_scale, int64_t M, int64_t K, int64_t lda,
                      hipStream_t st) {
    QL_DISPATCH_DTYPE(dtype, launch_act_quant, A, Aq, a_scale, M, K, lda, st)
}
int w8a8_gemm(int dtype, const int8_t* Aq, const float* a_scale, const int8_t* W, const void* S, const void* bias,
              void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, hipStream_t st) {
    QL_DISPATCH_DTYPE(dtype, launch_w8a8, Aq, a_scale, W, S, bias, C, M, N, K, ldc, st)
}

}  // namespace ql
