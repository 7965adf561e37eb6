// int8 per-channel weight-only GEMM for many activation rows (prefill / batched decode), gfx950.
//
//   C[M,N] = A[M,K] . (W^T * s[n])     W (N, K) int8 row-major, fp16 / bf16 activations, fp32 accumulation
//
// Same structure as w4_gemm.hip and the same arithmetic as the reference kernel (chatglm_q/int8/triton_ops.py:
// 62-73): every weight b * s is ROUNDED to the activation dtype in registers, products are exact inside
// v_mfma_f32_32x32x16_{f16,bf16}, accumulation is fp32.  Block = 4 waves x 32 output channels = 128 channels,
// BM = 32 MT rows, K step 64.  Lane (j = lane & 31, kb = lane >> 5) owns channel j and the 32 bytes
// k = 64 kt + 32 kb .. +31 of its weight row per step (two 16-byte loads); MFMA sub-step s uses bytes 8s..8s+7.
// fp16: byte -> half without cvt: (b ^ 0x80) spliced under 0x6400 = 1152 + b, minus 1152, times s; the splice
// yields the byte PAIRS (b0,b2),(b1,b3), so the A tile is staged with every 4 halves regrouped the same way
// (a0,a2,a1,a3) - a consistent K permutation of both operands leaves the sum unchanged.
#include "launch.h"
#include "ql_common.h"

namespace ql {

typedef _Float16 w8_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 w8_bf16x8 __attribute__((ext_vector_type(8)));
typedef float w8_f32x16 __attribute__((ext_vector_type(16)));

// regroup the 8 halves of a 16-byte chunk as (a0,a2),(a1,a3),(a4,a6),(a5,a7)
__device__ __forceinline__ u32x4 w8g_pair_even_odd(u32x4 x) {
    u32x4 y;
    y[0] = (x[0] & 0xFFFFu) | (x[1] << 16);
    y[1] = (x[0] >> 16) | (x[1] & 0xFFFF0000u);
    y[2] = (x[2] & 0xFFFFu) | (x[3] << 16);
    y[3] = (x[2] >> 16) | (x[3] & 0xFFFF0000u);
    return y;
}

template <typename T> struct W8Mma;
template <> struct W8Mma<f16> {
    typedef w8_f16x8 frag;
    static constexpr bool kPairedA = true;     // A staged as (a0,a2,a1,a3 | a4,a6,a5,a7)
    static __device__ __forceinline__ w8_f32x16 mma(frag a, frag b, w8_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    // 8 bytes (two words) -> 8 halves b * s in the paired order
    static __device__ __forceinline__ frag dequant(u32 w0, u32 w1, u32 k_mask, u32 k_magic, float s) {
        const h2 k1152 = {(f16)1152.0f, (f16)1152.0f};
        const f16 sh = (f16)s;
        const h2 s2 = {sh, sh};
        const u32 t0 = w0 ^ 0x80808080u, t1 = w1 ^ 0x80808080u;
        const h2 e0 = (as_h2((t0 & k_mask) | k_magic) - k1152) * s2;              // (b0, b2): exact, ONE rounding in * s
        const h2 e1 = (as_h2(((t0 >> 8) & k_mask) | k_magic) - k1152) * s2;       // (b1, b3)
        const h2 e2 = (as_h2((t1 & k_mask) | k_magic) - k1152) * s2;              // (b4, b6)
        const h2 e3 = (as_h2(((t1 >> 8) & k_mask) | k_magic) - k1152) * s2;       // (b5, b7)
        u32x4 r = {as_u32(e0), as_u32(e1), as_u32(e2), as_u32(e3)};
        return __builtin_bit_cast(frag, r);
    }
};
// the same with one scale PER contraction element (backward: the scale belongs to the contraction index);
// sc holds the octet's 8 scales in the operand's own element order
__device__ __forceinline__ w8_f16x8 w8_dequant_sk(u32 w0, u32 w1, u32 k_mask, u32 k_magic, u32x4 sc, f16) {
    const h2 k1152 = {(f16)1152.0f, (f16)1152.0f};
    const u32 t0 = w0 ^ 0x80808080u, t1 = w1 ^ 0x80808080u;
    u32x4 r;
    r[0] = as_u32((as_h2((t0 & k_mask) | k_magic) - k1152) * as_h2(sc[0]));
    r[1] = as_u32((as_h2(((t0 >> 8) & k_mask) | k_magic) - k1152) * as_h2(sc[1]));
    r[2] = as_u32((as_h2((t1 & k_mask) | k_magic) - k1152) * as_h2(sc[2]));
    r[3] = as_u32((as_h2(((t1 >> 8) & k_mask) | k_magic) - k1152) * as_h2(sc[3]));
    return __builtin_bit_cast(w8_f16x8, r);
}
__device__ __forceinline__ w8_bf16x8 w8_dequant_sk(u32 w0, u32 w1, u32, u32, u32x4 sc, __bf16) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32 w = i < 2 ? w0 : w1;
        const int sh = 16 * (i & 1);
        const float lo = (float)((int)(w << (24 - sh)) >> 24) * u32_as_f32(sc[i] << 16);
        const float hi = (float)((int)(w << (16 - sh)) >> 24) * u32_as_f32(sc[i] & 0xFFFF0000u);
        const bf2 p = {(__bf16)lo, (__bf16)hi};
        r[i] = __builtin_bit_cast(u32, p);
    }
    return __builtin_bit_cast(w8_bf16x8, r);
}

template <> struct W8Mma<__bf16> {
    typedef w8_bf16x8 frag;
    static constexpr bool kPairedA = false;    // natural K order
    static __device__ __forceinline__ w8_f32x16 mma(frag a, frag b, w8_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ frag dequant(u32 w0, u32 w1, u32, u32, float s) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 w = i < 2 ? w0 : w1;
            const int sh = 16 * (i & 1);
            const float lo = (float)((int)(w << (24 - sh)) >> 24) * s;            // byte (2i) of the octet
            const float hi = (float)((int)(w << (16 - sh)) >> 24) * s;            // byte (2i + 1)
            const bf2 p = {(__bf16)lo, (__bf16)hi};                               // one rounding each (v_cvt_pk_bf16_f32)
            r[i] = __builtin_bit_cast(u32, p);
        }
        return __builtin_bit_cast(frag, r);
    }
};

// SK = false: S[n] scales the output channel (forward).  SK = true: S[k] scales the CONTRACTION index (the backward
// product grad_A = grad_out . (W * s[:, None]) on a (K_out, N) row-major copy of the weights): the step's 64 scales go
// through LDS (128 bytes per block) and are read back as broadcast fragments.
template <typename T, int MT, int NW, int DEPTH, bool SK = false, bool TILED = false>
__global__ __launch_bounds__(NW * 64) void w8_gemm_kernel(const T* __restrict__ A, const int8_t* __restrict__ W,
                                                      const T* __restrict__ S, int M, int N, int K, int ldw32, int lda32,
                                                      int per, int nbx, const T* __restrict__ bias, T* __restrict__ C,
                                                      int64_t ldc, float* __restrict__ part) {
    // (argument order: the leading 14 dwords - what the first loads need - are preloaded into SGPRs at wave launch)
    const int64_t ldw = ldw32, lda = lda32;
    constexpr int BM = 32 * MT;
    constexpr int NTHR = NW * 64;
    constexpr int CH = (BM * 8 + NTHR - 1) / NTHR;   // 16-byte A chunks staged per thread per K step
    constexpr bool kAllStage = (BM * 8) % NTHR == 0;
    typedef W8Mma<T> MM;
    __shared__ __attribute__((aligned(16))) char smem[2][BM * 128];
    __shared__ __attribute__((aligned(16))) char ssk[2][SK ? 128 : 16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kb = lane >> 5;
    // 1-D grid, XCD-aware tile order (ql_common.h): tiles sharing an A row tile share an L2
    const TileXY tile = xcd_tile(blockIdx.x, gridDim.x, nbx);
    const int m0 = tile.y * BM;
    const int n_raw = tile.x * (NW * 32) + wave * 32 + j;
    const int n = n_raw < N ? n_raw : N - 1;
    const int ksteps = (K + 63) >> 6;
    const int k0 = blockIdx.z * per;           // split-K: this block's K steps are [k0, k0 + nst)
    const int nst = ksteps - k0 < per ? ksteps - k0 : per;
    auto gstep = [&](int t) { return k0 + (t < nst ? t : nst - 1); };

    u32 k_mask, k_magic;
    asm volatile("s_mov_b32 %0, 0x00FF00FF" : "=s"(k_mask));
    asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(k_magic));

    // TILED: W is the tile-major derived copy (w8_tile_kernel): [column tile][K step][lane][32 bytes], zero padded
    const int ctiles = (N + 31) >> 5;
    const int ct_raw = tile.x * NW + wave;
    const int8_t* wrow = TILED ? W + (int64_t)(ct_raw < ctiles ? ct_raw : ctiles - 1) * ksteps * 2048 + lane * 16
                               : W + (int64_t)n * ldw + kb * 32;
    const float sc = SK ? 0.f : Act<T>::load(S + n);

    const T* a_src[CH];
    int a_dst[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        const int q = tid + u * NTHR;
        const int r = (q >> 3) % BM, c = q & 7;
        const int row = (m0 + r < M) ? (m0 + r) : (M - 1);
        a_src[u] = A + (int64_t)row * lda + c * 8;
        a_dst[u] = (r * 8 + (c ^ ((r >> 1) & 7))) * 16;
    }
    const int kmax_a = K - 8;                  // last in-bounds 8-half chunk start
    const int kmax_w = K - 16;                 // last in-bounds 16-byte weight chunk start

    w8_f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    struct Stage {
        u32x4 a[CH];
        u32x4 w[2];
        u32x4 s;                               // SK: chunk (tid & 7) of the step's 64 scales
    };
    Stage st[DEPTH];
    auto load_stage = [&](int kt, Stage& sg) {
        if constexpr (SK) {
            const int c = tid & 7, k = kt * 64 + c * 8;
            sg.s = *reinterpret_cast<const u32x4*>(S + (k <= kmax_a ? k : kmax_a));
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int c = (tid + u * NTHR) & 7;
            const int k = kt * 64 + c * 8;
            sg.a[u] = *reinterpret_cast<const u32x4*>(a_src[u] + (k <= kmax_a ? kt * 64 : kmax_a - c * 8));   // K tail: clamped
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = kt * 64 + kb * 32 + h * 16;
            if constexpr (TILED) sg.w[h] = *reinterpret_cast<const u32x4*>(wrow + (int64_t)kt * 2048 + h * 1024);   // 1 KB contiguous per wave
            else sg.w[h] = *reinterpret_cast<const u32x4*>(wrow + (k <= kmax_w ? kt * 64 + h * 16 : kmax_w - kb * 32));   // cacheable: re-read by other row tiles
        }
    };
    auto store_a = [&](int buf, const Stage& sg) {
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (kAllStage || tid + u * NTHR < BM * 8)
                *reinterpret_cast<u32x4*>(smem[buf] + a_dst[u]) = MM::kPairedA ? w8g_pair_even_odd(sg.a[u]) : sg.a[u];
        if constexpr (SK) {
            if (tid < 8) *reinterpret_cast<u32x4*>(ssk[buf] + tid * 16) = MM::kPairedA ? w8g_pair_even_odd(sg.s) : sg.s;
        }
    };
    auto mma_step = [&](int buf, int kt, const u32x4 (&w_in)[2]) {
        // bytes of a K tail (k >= K) are replaced by 0: b = 0 contributes nothing whatever the (clamped) activations are
        u32x4 w[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) w[h] = (TILED || kt * 64 + kb * 32 + h * 16 <= kmax_w) ? w_in[h] : u32x4{0u, 0u, 0u, 0u};
        auto read_a = [&](int sub, u32x4 (&fr)[MT]) {
            const int c = kb * 4 + sub;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r = mt * 32 + j;
                fr[mt] = *reinterpret_cast<const u32x4*>(smem[buf] + (r * 8 + (c ^ ((r >> 1) & 7))) * 16);
            }
        };
        auto deq = [&](int sub) {
            if constexpr (SK) {
                const u32x4 fs = *reinterpret_cast<const u32x4*>(ssk[buf] + (kb * 4 + sub) * 16);   // broadcast read
                return w8_dequant_sk(w[sub >> 1][2 * (sub & 1)], w[sub >> 1][2 * (sub & 1) + 1], k_mask, k_magic, fs, T());
            } else {
                return MM::dequant(w[sub >> 1][2 * (sub & 1)], w[sub >> 1][2 * (sub & 1) + 1], k_mask, k_magic, sc);
            }
        };
        u32x4 fa[2][MT];
        typename MM::frag fb[2];
        read_a(0, fa[0]);
        fb[0] = deq(0);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            if (sub < 3) {
                read_a(sub + 1, fa[(sub + 1) & 1]);
                fb[(sub + 1) & 1] = deq(sub + 1);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = MM::mma(__builtin_bit_cast(typename MM::frag, fa[sub & 1][mt]), fb[sub & 1], acc[mt]);
            __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (16 + MT - 1) / MT, 0);
            }
        }
    };

#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_stage(gstep(d), st[d]);
    store_a(0, st[0]);
    __syncthreads();

    int kt = 0;
    for (; kt + DEPTH < nst; kt += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int buf = (kt + d) & 1;
            const u32x4 w_cur[2] = {st[d].w[0], st[d].w[1]};
            load_stage(gstep(kt + d + DEPTH), st[d]);
            mma_step(buf, gstep(kt + d), w_cur);
            store_a(buf ^ 1, st[(d + 1) % DEPTH]);
            __syncthreads();
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        if (kt + d < nst) {
            const int buf = (kt + d) & 1;
            mma_step(buf, gstep(kt + d), st[d].w);
            if (kt + d + 1 < nst) store_a(buf ^ 1, st[(d + 1) % DEPTH]);
            __syncthreads();
        }
    }

    if (!part && (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
        // rounded tiles through the (now idle) A-tile LDS, 16-byte row chunks to global (ql_common.h)
        T* lds_wave = reinterpret_cast<T*>(smem[0]) + wave * 1024;
        static_assert(NW * 2048 <= 2 * BM * 128, "one 2 KB epilogue region per wave inside the A-tile buffers");
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            store_tile_32x32<T>(lds_wave, C, ldc, m0 + mt * 32, n_raw - j, M, N, bias, lane, [&](int i) { return acc[mt][i]; });
        return;
    }
    if (n_raw < N) {
        const T* bn = bias ? bias + n_raw : nullptr;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
                if (m >= M) continue;
                if (part) part[((int64_t)blockIdx.z * M + m) * N + n_raw] = acc[mt][i];   // fp32 slab; summed by splitk_reduce_kernel
                else store_out<T>(C + (int64_t)m * ldc + n_raw, acc[mt][i], bn);
            }
    }
}

template <typename T, int MT, int NW, bool SK = false, bool TILED = false>
static int launch_w8_gemm(const void* A, const int8_t* W, const void* S, const void* bias, void* C, int M, int N, int K,
                          int64_t ldw, int64_t lda, int64_t ldc, const GemmPlan& plan, float* ws, hipStream_t st) {
    if (ldw > 0x7fffffff || lda > 0x7fffffff) return QL_ERR_UNSUPPORTED;     // strides travel as 32 bits
    float* part = plan.ksplit > 1 ? ws : nullptr;
    constexpr int BN = NW * 32;
    const int nbx = (N + BN - 1) / BN, nby = (M + 32 * MT - 1) / (32 * MT);
    dim3 grid((unsigned)(nbx * nby), 1, (unsigned)plan.ksplit);
    w8_gemm_kernel<T, MT, NW, 3, SK, TILED><<<grid, NW * 64, 0, st>>>((const T*)A, W, (const T*)S, M, N, K, (int)ldw, (int)lda, plan.per,
        xcd_order(nbx, nby, (double)M * K * 2, (double)N * K), (const T*)bias, (T*)C, ldc, part);
    const int rc = finish_launch(QL_K_W8_GEMM128);
    if (rc != 0 || !part) return rc;
    const int64_t total = (int64_t)M * N;
    splitk_reduce_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(part, (const T*)bias, (T*)C, M, N, ldc, plan.ksplit);
    return finish_launch(QL_K_SPLITK_REDUCE);
}

template <typename T>
static int launch_w8_gemm_any(const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t M,
                              int64_t N, int64_t K, int64_t ldw, int64_t lda, int64_t ldc, void* ws, size_t ws_bytes,
                              hipStream_t st) {
    const GemmPlan plan = gemm_plan(M, N, (K + 63) / 64, ws && ((uintptr_t)ws & 15) == 0 ? ws_bytes : 0);
    switch (plan.mt) {
    case 4:
        if (((N + 255) / 256) * ((M + 127) / 128) >= 256)   // see w4_gemm.hip
            return launch_w8_gemm<T, 4, 8>(A, W, S, bias, C, (int)M, (int)N, (int)K, ldw, lda, ldc, plan, (float*)ws, st);
        return launch_w8_gemm<T, 4, 4>(A, W, S, bias, C, (int)M, (int)N, (int)K, ldw, lda, ldc, plan, (float*)ws, st);
    case 2: return launch_w8_gemm<T, 2, 4>(A, W, S, bias, C, (int)M, (int)N, (int)K, ldw, lda, ldc, plan, (float*)ws, st);
    default: return launch_w8_gemm<T, 1, 4>(A, W, S, bias, C, (int)M, (int)N, (int)K, ldw, lda, ldc, plan, (float*)ws, st);
    }
}

// backward: C (M, N) = A (M, K) . (W (N, K) with S[k] on the contraction)^T ; no bias, no split-K
template <typename T>
static int launch_w8_gemm_sk(const void* A, const int8_t* W, const void* S, void* C, int64_t M, int64_t N, int64_t K,
                             int64_t ldw, int64_t lda, int64_t ldc, hipStream_t st) {
    const GemmPlan plan{0, 1, (int)((K + 63) / 64)};
    if (M > 64 && ((N + 127) / 128) * ((M + 127) / 128) >= 256)
        return launch_w8_gemm<T, 4, 4, true>(A, W, S, nullptr, C, (int)M, (int)N, (int)K, ldw, lda, ldc, plan, nullptr, st);
    if (M > 32) return launch_w8_gemm<T, 2, 4, true>(A, W, S, nullptr, C, (int)M, (int)N, (int)K, ldw, lda, ldc, plan, nullptr, st);
    return launch_w8_gemm<T, 1, 4, true>(A, W, S, nullptr, C, (int)M, (int)N, (int)K, ldw, lda, ldc, plan, nullptr, st);
}

int w8_gemm_scale_k(int dtype, const void* A, const int8_t* W, const void* S, void* C, int64_t M, int64_t N, int64_t K,
                    int64_t ldw, int64_t lda, int64_t ldc, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_w8_gemm_sk<f16>(A, W, S, C, M, N, K, ldw, lda, ldc, st);
    case QL_DTYPE_BF16: return launch_w8_gemm_sk<__bf16>(A, W, S, C, M, N, K, ldw, lda, ldc, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

// many rows on the tile-major derived copy of the weights (qlinear_w8_tile)
template <typename T>
static int launch_w8_gemm_tiled(const void* A, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M, int64_t N,
                                int64_t K, int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st) {
    const GemmPlan plan = gemm_plan(M, N, (K + 63) / 64, ws && ((uintptr_t)ws & 15) == 0 ? ws_bytes : 0);
    switch (plan.mt) {
    case 4:
        if (((N + 255) / 256) * ((M + 127) / 128) >= 256)
            return launch_w8_gemm<T, 4, 8, false, true>(A, Wm, S, bias, C, (int)M, (int)N, (int)K, K, lda, ldc, plan, (float*)ws, st);
        return launch_w8_gemm<T, 4, 4, false, true>(A, Wm, S, bias, C, (int)M, (int)N, (int)K, K, lda, ldc, plan, (float*)ws, st);
    case 2: return launch_w8_gemm<T, 2, 4, false, true>(A, Wm, S, bias, C, (int)M, (int)N, (int)K, K, lda, ldc, plan, (float*)ws, st);
    default: return launch_w8_gemm<T, 1, 4, false, true>(A, Wm, S, bias, C, (int)M, (int)N, (int)K, K, lda, ldc, plan, (float*)ws, st);
    }
}

int w8_gemm_tiled(int dtype, const void* A, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M, int64_t N,
                  int64_t K, int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st) {
    if ((dtype == QL_DTYPE_F16 || dtype == QL_DTYPE_BF16) && w4_gemm256_supported(M, N, K, lda, A, 2))
        return w8_gemm256(dtype, A, Wm, S, bias, C, M, N, K, lda, ldc, st);     // prefill row counts: 256 x 256 tiles (w4_gemm256.hip)
    switch (dtype) {
    case QL_DTYPE_F16: return launch_w8_gemm_tiled<f16>(A, Wm, S, bias, C, M, N, K, lda, ldc, ws, ws_bytes, st);
    case QL_DTYPE_BF16: return launch_w8_gemm_tiled<__bf16>(A, Wm, S, bias, C, M, N, K, lda, ldc, ws, ws_bytes, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

// tile-major derived copy: Wm[ct][kt][h][lane][16 bytes], lane = 32 kb + j, half h: bytes k = 64 kt + 32 kb + 16 h .. + 15 of
// output channel 32 ct + j; rows past N and bytes past K are 0.  One wave load instruction = the 64 lanes' 16-byte pieces of
// one half = 1 KB contiguous.  (Round 1 stored a lane's two halves side by side, [lane][32 bytes]: every 16-byte load
// instruction then touched 16 cache lines and used half of each - measured in round 2 on the W8A8 GEMM, whose K loop
// turned out to be bound by the CU's vector-memory pipe.)
__global__ __launch_bounds__(256) void w8_tile_kernel(const int8_t* __restrict__ W, int8_t* __restrict__ Wm, int N, int K,
                                                      int64_t ldw, int ksteps, int64_t total16) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // 16-byte piece: ((ct * ksteps + kt) * 2 + h) * 64 + lane
    if (idx >= total16) return;
    const int lane = (int)(idx & 63), h = (int)((idx >> 6) & 1);
    const int64_t step = idx >> 7;
    const int kt = (int)(step % ksteps), ct = (int)(step / ksteps);
    const int n = ct * 32 + (lane & 31), k = kt * 64 + (lane >> 5) * 32 + h * 16;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (n < N && k < K) {
        if (k + 16 <= K) {
            v = *reinterpret_cast<const u32x4*>(W + (int64_t)n * ldw + k);
        } else {
            int8_t tmp[16];
            for (int e = 0; e < 16; ++e) tmp[e] = k + e < K ? W[(int64_t)n * ldw + k + e] : (int8_t)0;
            v = *reinterpret_cast<const u32x4*>(tmp);
        }
    }
    reinterpret_cast<u32x4*>(Wm)[idx] = v;
}

size_t w8_tiled_bytes(int64_t N, int64_t K) { return (size_t)((N + 31) / 32) * ((K + 63) / 64) * 2048; }

int w8_tile(const int8_t* W, int8_t* Wm, int64_t N, int64_t K, int64_t ldw, hipStream_t st) {
    const int64_t ksteps = (K + 63) / 64, total16 = ((N + 31) / 32) * ksteps * 128;
    w8_tile_kernel<<<(unsigned)((total16 + 255) / 256), 256, 0, st>>>(W, Wm, (int)N, (int)K, ldw, (int)ksteps, total16);
    return finish_launch();
}

// ---------------------------------------------------------------------------------------------
// Few rows (3 <= M <= 32) on the tile-major copy: the int8 twin of w4_fewrow_kernel (w4_fewrow.hip) - KW independent
// waves per block share 32 output channels and split K, each stages its own A tile in a private LDS region, no block
// barrier until the partial sums are combined through LDS; optional split over blockIdx.y into fp32 slabs.
// ---------------------------------------------------------------------------------------------
template <typename T, int KW>
__global__ __launch_bounds__(KW * 64) void w8_fewrow_kernel(const T* __restrict__ A, const int8_t* __restrict__ Wm,
                                                            const T* __restrict__ S, int M, int N, int K, int64_t lda, int per,
                                                            const T* __restrict__ bias, T* __restrict__ C, int64_t ldc,
                                                            float* __restrict__ part) {
    // (argument order: the leading 14 dwords - what the first loads need - are preloaded into SGPRs at wave launch)
    typedef W8Mma<T> MM;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // KW x 2 x 4 KB A tiles; reused for the reduction
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kb = lane >> 5;
    const int n_raw = blockIdx.x * 32 + j;
    const int ksteps = (K + 63) >> 6;
    const int slice = blockIdx.y * KW + wave;
    const int k0 = slice * per;
    const int nst = k0 >= ksteps ? 0 : (ksteps - k0 < per ? ksteps - k0 : per);

    u32 k_mask, k_magic;
    asm volatile("s_mov_b32 %0, 0x00FF00FF" : "=s"(k_mask));
    asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(k_magic));

    const int8_t* wtile = Wm + (int64_t)blockIdx.x * ksteps * 2048 + lane * 16;
    const float sc = Act<T>::load(S + (n_raw < N ? n_raw : N - 1));
    char* abuf = smem + wave * 8192;

    const T* a_src[4];
    int a_dst[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = lane + 64 * u, r = q >> 3, c = q & 7;
        a_src[u] = A + (int64_t)(r < M ? r : M - 1) * lda + c * 8;
        a_dst[u] = (r * 8 + (c ^ ((r >> 1) & 7))) * 16;
    }
    const int c_mine = lane & 7;
    const int kmax = K - 8;

    w8_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    struct Stage {
        u32x4 a[4];
        u32x4 w[2];
    };
    auto load_stage = [&](int kt, Stage& sg) {
        const int k = kt * 64 + c_mine * 8;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (8 * u < M)                                     // rows past M are never loaded (block-uniform test)
                sg.a[u] = *reinterpret_cast<const u32x4*>(a_src[u] + (k <= kmax ? kt * 64 : kmax - c_mine * 8));
#pragma unroll
        for (int h = 0; h < 2; ++h)
            sg.w[h] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wtile + (int64_t)kt * 2048 + h * 1024));
    };
    auto store_a = [&](int buf, const Stage& sg) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (8 * u < M)
                *reinterpret_cast<u32x4*>(abuf + buf * 4096 + a_dst[u]) = MM::kPairedA ? w8g_pair_even_odd(sg.a[u]) : sg.a[u];
    };
    auto mma_step = [&](int buf, const u32x4 (&w)[2]) {
        const char* sa = abuf + buf * 4096;
        u32x4 fa[2];
        typename MM::frag fb[2];
        auto read_a = [&](int sub) {
            const int c = kb * 4 + sub;
            return *reinterpret_cast<const u32x4*>(sa + (j * 8 + (c ^ ((j >> 1) & 7))) * 16);
        };
        fa[0] = read_a(0);
        fb[0] = MM::dequant(w[0][0], w[0][1], k_mask, k_magic, sc);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            if (sub < 3) {
                const int s1 = sub + 1;
                fa[s1 & 1] = read_a(s1);
                fb[s1 & 1] = MM::dequant(w[s1 >> 1][2 * (s1 & 1)], w[s1 >> 1][2 * (s1 & 1) + 1], k_mask, k_magic, sc);
            }
            acc = MM::mma(__builtin_bit_cast(typename MM::frag, fa[sub & 1]), fb[sub & 1], acc);
        }
    };

    // bias of the lane's channel: requested behind the first weight loads (late kernel argument), not in the wave's tail
    float bias_j = 0.f;
    if (nst > 0) {
        Stage st0, st1;
        load_stage(k0, st0);
        load_stage(k0 + (nst > 1 ? 1 : 0), st1);
        bias_j = Act<T>::load((bias ? bias : S) + (bias && n_raw < N ? n_raw : 0));   // unconditional: no queue drain
        int t = 0;
        for (; t + 2 < nst; t += 2) {
            store_a(0, st0);
            {
                const u32x4 w[2] = {st0.w[0], st0.w[1]};
                load_stage(k0 + t + 2, st0);
                mma_step(0, w);
            }
            store_a(1, st1);
            {
                const u32x4 w[2] = {st1.w[0], st1.w[1]};
                load_stage(k0 + (t + 3 < nst ? t + 3 : nst - 1), st1);
                mma_step(1, w);
            }
        }
        store_a(0, st0);
        mma_step(0, st0.w);
        if (t + 1 < nst) {
            store_a(1, st1);
            mma_step(1, st1.w);
        }
    }

    if (nst == 0 && bias) bias_j = Act<T>::load(bias + (n_raw < N ? n_raw : 0));

    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[((wave - 1) * 16 + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < KW; ++w)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += red[((w - 1) * 16 + i) * 64 + lane];
    if (n_raw >= N) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = (i & 3) + 8 * (i >> 2) + 4 * kb;
        if (m >= M) continue;
        if (part) {
            part[((int64_t)blockIdx.y * M + m) * N + n_raw] = acc[i];
        } else {                                              // store_out's sequence with the preloaded bias
            float y = Act<T>::round(acc[i]);
            if (bias) y = y + bias_j;
            Act<T>::store(C + (int64_t)m * ldc + n_raw, y);
        }
    }
}

constexpr int kW8FewRowWaves = 4;
struct W8FewRowPlan {
    int ksplit, per;
};
static W8FewRowPlan w8_fewrow_plan(int64_t M, int64_t N, int64_t K, size_t ws_bytes) {
    const int64_t ksteps = (K + 63) / 64, nb = (N + 31) / 32;
    int64_t ks = (2048 + nb * kW8FewRowWaves - 1) / (nb * kW8FewRowWaves);
    if (ks > ksteps / (2 * kW8FewRowWaves)) ks = ksteps / (2 * kW8FewRowWaves);
    while (ks > 1 && (size_t)(ks * M * N) * sizeof(float) > ws_bytes) --ks;
    if (ks < 1) ks = 1;
    const int64_t per = (ksteps + ks * kW8FewRowWaves - 1) / (ks * kW8FewRowWaves);
    return {(int)ks, (int)per};
}

template <typename T>
static int launch_w8_fewrow(const void* A, const int8_t* Wm, const void* S, const void* bias, void* C, int M, int N, int K,
                            int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int KW = kW8FewRowWaves;
    const W8FewRowPlan plan = w8_fewrow_plan(M, N, K, ws && ((uintptr_t)ws & 15) == 0 ? ws_bytes : 0);
    float* part = plan.ksplit > 1 ? (float*)ws : nullptr;
    dim3 grid((unsigned)((N + 31) / 32), (unsigned)plan.ksplit);
    w8_fewrow_kernel<T, KW><<<grid, KW * 64, (size_t)KW * 8192, st>>>((const T*)A, Wm, (const T*)S, M, N, K, lda, plan.per,
                                                                      (const T*)bias, (T*)C, ldc, part);
    const int rc = finish_launch(QL_K_W8_FEWROW);
    if (rc != 0 || !part) return rc;
    const int64_t total = (int64_t)M * N;
    splitk_reduce_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(part, (const T*)bias, (T*)C, M, N, ldc, plan.ksplit);
    return finish_launch(QL_K_SPLITK_REDUCE);
}

// rows >= 3 on the tile-major copy: few-row kernel up to 32 rows, tiled GEMM above
size_t w8_tiled_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (M <= 32) {
        const W8FewRowPlan p = w8_fewrow_plan(M, N, K, (size_t)-1);
        return p.ksplit > 1 ? (size_t)(p.ksplit * M * N) * sizeof(float) : 0;
    }
    return w8_gemm_workspace_bytes(M, N, K);
}

int w8_fwd_tiled(int dtype, const void* A, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M, int64_t N,
                 int64_t K, int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st) {
    if (M > 32) return w8_gemm_tiled(dtype, A, Wm, S, bias, C, M, N, K, lda, ldc, ws, ws_bytes, st);
    switch (dtype) {
    case QL_DTYPE_F16: return launch_w8_fewrow<f16>(A, Wm, S, bias, C, (int)M, (int)N, (int)K, lda, ldc, ws, ws_bytes, st);
    case QL_DTYPE_BF16: return launch_w8_fewrow<__bf16>(A, Wm, S, bias, C, (int)M, (int)N, (int)K, lda, ldc, ws, ws_bytes, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

size_t w8_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) { return gemm_workspace_bytes(M, N, (K + 63) / 64); }

int w8_gemm(int dtype, const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t M, int64_t N,
            int64_t K, int64_t ldw, int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_w8_gemm_any<f16>(A, W, S, bias, C, M, N, K, ldw, lda, ldc, ws, ws_bytes, st);
    case QL_DTYPE_BF16: return launch_w8_gemm_any<__bf16>(A, W, S, bias, C, M, N, K, ldw, lda, ldc, ws, ws_bytes, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

}  // namespace ql
