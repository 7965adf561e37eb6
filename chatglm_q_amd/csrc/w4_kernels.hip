// int4 group-quantised linear forward kernels for gfx950 (MI355X).
//
// Replaces _dynamic_quant_matmul_s4_kernel (chatglm_q/int4/triton_ops.py:18-87).
//
// Kernels on the reference (canonical) layout, weight-only dequant fused into the contraction,
// fp32 accumulation:
//   w4_generic_kernel      any group size / any N; one thread per output element (robustness path)
//   w4_canon_kernel        (K/2, N) layout, group 32: 128-column x 16-group tiles,
//                          split-K over workgroups + splitk_reduce_kernel
// The derived-layout kernels live in w4_packed.hip.
#include "launch.h"
#include "w4_dequant.h"

namespace ql {

// =============================================================================================
// generic: one thread per (m, n)
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void w4_generic_kernel(const T* __restrict__ A, const uint8_t* __restrict__ Wq,
                                                         const T* __restrict__ S, const T* __restrict__ bias,
                                                         T* __restrict__ C, int M, int N, int K, int group,
                                                         int64_t lda, int64_t ldc) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.y;
    if (n >= N || m >= M) return;
    const T* a = A + (int64_t)m * lda;
    float acc = 0.f;
    for (int k = 0; k < K; k += 2) {
        const u32 b = Wq[(int64_t)(k >> 1) * N + n];
        const float s0 = Act<T>::load(S + (int64_t)(k / group) * N + n);
        const float s1 = Act<T>::load(S + (int64_t)((k + 1) / group) * N + n);
        const float w0 = Act<T>::round(((float)(b & 0xFu) - 8.0f) * s0);
        const float w1 = Act<T>::round(((float)(b >> 4) - 8.0f) * s1);
        acc = __builtin_fmaf(Act<T>::load(a + k), w0, acc);
        acc = __builtin_fmaf(Act<T>::load(a + k + 1), w1, acc);
    }
    store_out<T>(C + (int64_t)m * ldc + n, acc, bias ? bias + n : nullptr);
}

// =============================================================================================
// canonical layout, group 32
//   block (256 thr) = 128 columns x 16 groups (512 k).  lane: jn = lane & 15 -> 8 columns
//   (one 8-byte load per packed row), q = lane >> 4 and wave -> which of the 16 groups.
//   Every 128-byte cache line of the weight matrix is consumed by exactly one wave instruction.
// =============================================================================================
template <typename T, int MB> struct CanonRow;   // per-row FMA step, specialised for f16

// generic (fp32 / bf16) step: 8 columns x 2 k from one 8-byte unit
template <typename T, int MB>
__device__ __forceinline__ void canon_step_generic(u32 wlo, u32 whi, const float (&s)[8], const float (&m8s)[8],
                                                   const float (&a0)[MB], const float (&a1)[MB], float (&acc)[MB][8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const u32 w = c < 4 ? wlo : whi;
        const int cc = c & 3;
        const float w0 = dequant_nibble<T>(w, 2 * cc, s[c], m8s[c]);
        const float w1 = dequant_nibble<T>(w, 2 * cc + 1, s[c], m8s[c]);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            acc[m][c] = __builtin_fmaf(a0[m], w0, acc[m][c]);
            acc[m][c] = __builtin_fmaf(a1[m], w1, acc[m][c]);
        }
    }
}

template <typename T, int MB>
__global__ __launch_bounds__(256) void w4_canon_kernel(const T* __restrict__ A, const uint8_t* __restrict__ Wq,
                                                       const T* __restrict__ S, const T* __restrict__ bias,
                                                       T* __restrict__ C, float* __restrict__ partial, int M, int N,
                                                       int K, int G, int64_t lda, int64_t ldc, int ksplit) {
    __shared__ float red[4][MB][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jn = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 128 + jn * 8;
    const int g = blockIdx.y * 16 + wave * 4 + q;
    const int m0 = blockIdx.z * MB;

    float acc[MB][8];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[m][c] = 0.f;

    if (g < G && n0 < N) {
        // all 16 packed rows of this lane's group: issue every load before the first use
        const uint8_t* wp = Wq + (int64_t)g * 16 * N + n0;
        u32x2 wv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) wv[r] = *reinterpret_cast<const u32x2*>(wp + (int64_t)r * N);

        const T* sp = S + (int64_t)g * N + n0;
        const T* ap[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int mm = (m0 + m < M) ? (m0 + m) : (M - 1);
            ap[m] = A + (int64_t)mm * lda + g * 32;
        }

        if constexpr (sizeof(T) == 2 && Act<T>::code == QL_DTYPE_F16) {
            // scales: 8 halves = one 16-byte load; rearranged to (c0,c2),(c1,c3),(c4,c6),(c5,c7)
            const u32x4 sv = *reinterpret_cast<const u32x4*>(sp);
            const h2 s02 = as_h2((sv[0] & 0xFFFFu) | (sv[1] << 16));
            const h2 s13 = as_h2((sv[0] >> 16) | (sv[1] & 0xFFFF0000u));
            const h2 s46 = as_h2((sv[2] & 0xFFFFu) | (sv[3] << 16));
            const h2 s57 = as_h2((sv[2] >> 16) | (sv[3] & 0xFFFF0000u));
            // activations: 32 halves per row = four 16-byte loads; pair r = (a[2r], a[2r+1])
            h2 av[MB][16];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 t = *reinterpret_cast<const u32x4*>(ap[m] + 8 * j);
                    av[m][4 * j + 0] = as_h2(t[0]);
                    av[m][4 * j + 1] = as_h2(t[1]);
                    av[m][4 * j + 2] = as_h2(t[2]);
                    av[m][4 * j + 3] = as_h2(t[3]);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int hsel = 0; hsel < 2; ++hsel) {
                    // word nibble p <-> (column p/2, k parity p%2):
                    //   e0 = (c0,c2) k even, e1 = (c0,c2) k odd, e2 = (c1,c3) k even, e3 = (c1,c3) k odd
                    const NibblePairs e = nibble_pairs_f16(wv[r][hsel]);
                    const h2 sa = hsel ? s46 : s02, sb = hsel ? s57 : s13;
                    const h2 w_even_a = e.e0 * sa, w_odd_a = e.e1 * sa;   // rounded to fp16 (faithful)
                    const h2 w_even_b = e.e2 * sb, w_odd_b = e.e3 * sb;
                    const int cb = hsel * 4;
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const float ae = (float)av[m][r].x, ao = (float)av[m][r].y;
                        acc[m][cb + 0] = __builtin_fmaf(ae, (float)w_even_a.x, acc[m][cb + 0]);
                        acc[m][cb + 0] = __builtin_fmaf(ao, (float)w_odd_a.x, acc[m][cb + 0]);
                        acc[m][cb + 2] = __builtin_fmaf(ae, (float)w_even_a.y, acc[m][cb + 2]);
                        acc[m][cb + 2] = __builtin_fmaf(ao, (float)w_odd_a.y, acc[m][cb + 2]);
                        acc[m][cb + 1] = __builtin_fmaf(ae, (float)w_even_b.x, acc[m][cb + 1]);
                        acc[m][cb + 1] = __builtin_fmaf(ao, (float)w_odd_b.x, acc[m][cb + 1]);
                        acc[m][cb + 3] = __builtin_fmaf(ae, (float)w_even_b.y, acc[m][cb + 3]);
                        acc[m][cb + 3] = __builtin_fmaf(ao, (float)w_odd_b.y, acc[m][cb + 3]);
                    }
                }
            }
        } else {
            float s[8], m8s[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                s[c] = Act<T>::load(sp + c);
                m8s[c] = -8.0f * s[c];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a0[MB], a1[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    a0[m] = Act<T>::load(ap[m] + 2 * r);
                    a1[m] = Act<T>::load(ap[m] + 2 * r + 1);
                }
                canon_step_generic<T, MB>(wv[r][0], wv[r][1], s, m8s, a0, a1, acc);
            }
        }
    }

    // combine the 4 group-lanes of the wave (lanes l, l^16, l^32, l^48 hold the same columns)
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = acc[m][c];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[m][c] = v;
        }
    if (q == 0) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int c = 0; c < 8; ++c) red[wave][m][jn * 8 + c] = acc[m][c];
    }
    __syncthreads();
    for (int idx = tid; idx < MB * 128; idx += 256) {
        const int m = idx >> 7, c = idx & 127;
        const int n = blockIdx.x * 128 + c;
        if (n < N && m0 + m < M) {
            const float v = (red[0][m][c] + red[1][m][c]) + (red[2][m][c] + red[3][m][c]);
            if (ksplit == 1)
                store_out<T>(C + (int64_t)(m0 + m) * ldc + n, v, bias ? bias + n : nullptr);
            else
                partial[((int64_t)blockIdx.y * M + (m0 + m)) * N + n] = v;
        }
    }
}

// =============================================================================================
// host-side launchers (called from abi.hip)
// =============================================================================================
template <typename T>
static int launch_w4_generic(const void* A, const uint8_t* Wq, const void* S, const void* bias, void* C, int64_t M,
                             int64_t N, int64_t K, int64_t group, int64_t lda, int64_t ldc, hipStream_t st) {
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)M);
    w4_generic_kernel<T><<<grid, 256, 0, st>>>((const T*)A, Wq, (const T*)S, (const T*)bias, (T*)C, (int)M, (int)N,
                                               (int)K, (int)group, lda, ldc);
    return finish_launch(QL_K_W4_GENERIC);
}

template <typename T, int MB>
static int launch_w4_canon_mb(const T* A, const uint8_t* Wq, const T* S, const T* bias, T* C, float* ws, int M, int N,
                              int K, int64_t lda, int64_t ldc, hipStream_t st) {
    const int G = K / 32;
    const int ksplit = (G + 15) / 16;
    dim3 grid((unsigned)((N + 127) / 128), (unsigned)ksplit, (unsigned)((M + MB - 1) / MB));
    w4_canon_kernel<T, MB><<<grid, 256, 0, st>>>(A, Wq, S, bias, C, ws, M, N, K, G, lda, ldc, ksplit);
    int rc = finish_launch(QL_K_W4_CANON);
    if (rc != 0 || ksplit == 1) return rc;
    const int64_t total = (int64_t)M * N;
    splitk_reduce_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(ws, bias, C, M, N, ldc, ksplit);
    return finish_launch(QL_K_SPLITK_REDUCE);
}

size_t w4_canon_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    const int64_t G = K / 32, ksplit = (G + 15) / 16;
    if (ksplit <= 1) return 0;
    const int64_t mc = M < kCanonMChunk ? M : kCanonMChunk;
    return (size_t)(ksplit * mc * N) * sizeof(float);
}

template <typename T>
static int launch_w4_canon(const void* A_, const uint8_t* Wq, const void* S_, const void* bias_, void* C_, void* ws,
                           int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc, hipStream_t st) {
    const T* S = (const T*)S_;
    const T* bias = (const T*)bias_;
    for (int64_t mbase = 0; mbase < M; mbase += kCanonMChunk) {
        const int mc = (int)((M - mbase) < kCanonMChunk ? (M - mbase) : kCanonMChunk);
        const T* A = (const T*)A_ + mbase * lda;
        T* C = (T*)C_ + mbase * ldc;
        int rc;
        if (mc == 1)
            rc = launch_w4_canon_mb<T, 1>(A, Wq, S, bias, C, (float*)ws, mc, (int)N, (int)K, lda, ldc, st);
        else if (mc == 2)
            rc = launch_w4_canon_mb<T, 2>(A, Wq, S, bias, C, (float*)ws, mc, (int)N, (int)K, lda, ldc, st);
        else
            rc = launch_w4_canon_mb<T, 4>(A, Wq, S, bias, C, (float*)ws, mc, (int)N, (int)K, lda, ldc, st);
        if (rc) return rc;
    }
    return 0;
}

#define QL_DISPATCH_DTYPE(dtype, fn, ...)                         \
    switch (dtype) {                                              \
    case QL_DTYPE_F32: return fn<float>(__VA_ARGS__);             \
    case QL_DTYPE_F16: return fn<f16>(__VA_ARGS__);               \
    case QL_DTYPE_BF16: return fn<__bf16>(__VA_ARGS__);           \
    default: return QL_ERR_BAD_DTYPE;                             \
    }

int w4_generic(int dtype, const void* A, const uint8_t* Wq, const void* S, const void* bias, void* C, int64_t M,
               int64_t N, int64_t K, int64_t group, int64_t lda, int64_t ldc, hipStream_t st) {
    QL_DISPATCH_DTYPE(dtype, launch_w4_generic, A, Wq, S, bias, C, M, N, K, group, lda, ldc, st)
}
int w4_canon(int dtype, const void* A, const uint8_t* Wq, const void* S, const void* bias, void* C, void* ws,
             int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc, hipStream_t st) {
    QL_DISPATCH_DTYPE(dtype, launch_w4_canon, A, Wq, S, bias, C, ws, M, N, K, lda, ldc, st)
}
}  // namespace ql
