// int8 per-output-channel quantised linear forward kernels for gfx950 (MI355X).
//
// Replaces _dynamic_quant_matmul_kernel (chatglm_q/int8/triton_ops.py:13-84) and adds the
// int8-activation path whose semantic the reference only states in its ONNX symbolic
// (chatglm_q/int8/qlinear.py:56-70) and quantiser (chatglm_q/int8/quantizer.py:11-19).
//
//   w8_generic_kernel        any strides (incl. the (K, N)-contiguous form the reference test uses)
//   w8_gemv_kernel           module layout: W (N, K) row-major, K contiguous; one wave owns 4 output
//                            channels and streams their rows with 16-byte loads (decode shapes)
//   w8a8_mfma_kernel         i8 x i8 -> i32 on v_mfma_i32_32x32x32_i8, rank-1 scale epilogue, weights in the
//                            module's (N, K) row-major buffer (both operands through LDS); the tile-major
//                            kernel and the activation quantiser live in w8a8.hip
#include <stdlib.h>

// developer ablation switches for tools/microbench/w8a8_ablate.hip (always 0 in the library): 1 no MFMA, 2 no
// steady-state global loads, 4 no LDS stores, 8 no LDS fragment reads, 16 no barriers
#ifndef QL_W8A8_ABLATE
#define QL_W8A8_ABLATE 0
#endif
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

// A wave "tile" = 2 x (64 lanes x 16 B) per output channel = 2048 k: K = 4096 is exactly two tiles,
// both of which are in flight before anything waits (same latency structure as w4_packed.hip).
template <int MB, bool A_LDS>
struct W8Tile {
    u32x4 w[4][2];
    u32x4 a[A_LDS ? 1 : MB][A_LDS ? 1 : 4];
};

// LDS image of the activation row for the fp16 kernel: 16-byte piece j (0/1) of lane-chunk lc
// (k = 16 lc .. 16 lc + 15) lives at piece position 2 lc + (j ^ ((lc >> 3) & 1)) - the XOR makes the
// 32-byte-strided per-lane ds_read_b128 bank-conflict free.  Each piece is stored PRE-PAIRED as
// (a0,a2),(a1,a3),(a4,a6),(a5,a7) so that it lines up with byte_pairs_f16() without per-use shuffles.
__device__ __forceinline__ int w8_piece_pos(int lc, int j) { return 2 * lc + (j ^ ((lc >> 3) & 1)); }

__device__ __forceinline__ u32x4 pair_even_odd(u32x4 x) {
    u32x4 y;
    y[0] = (x[0] & 0xFFFFu) | (x[1] << 16);
    y[1] = (x[0] >> 16) | (x[1] & 0xFFFF0000u);
    y[2] = (x[2] & 0xFFFFu) | (x[3] << 16);
    y[3] = (x[2] >> 16) | (x[3] & 0xFFFF0000u);
    return y;
}

// STRICT: every weight b * s is rounded to fp16 before the dot (the reference's sequence, chatglm_q/int8/
// triton_ops.py:70).  Otherwise ("exact-dequant", default) the spliced offset form 1152 + b feeds v_dot2c as is
// and s * (sum a (1152 + b) - 1152 sum a) is applied once per output channel and 16-k chunk: 6 instead of 10
// VALU ops per 4 weights, closer to real arithmetic, ~2e-4 relative from the reference (see w4_packed.hip).
// KS: the block's 4 waves cover 4/KS channel quads x KS slices of K (combined through LDS at the end), so that a
// 4096 x 4096 layer gives 2 blocks per CU instead of 1 (see w4_packed.hip).
// PRO = PRO_ADDNORM (one row, LDS staging, K % 16 == 0): the residual add + RMSNorm in front of the projection runs
// while the row is staged, and pro.gate_epilogue turns the quad's sums into SiLU(h) * gate - the int8 twin of
// w4_packed.hip's fused decode GEMV (same rounding sequence).
template <int MB, int ACH, int KS, bool STRICT, int PRO = PRO_NONE>   // ACH: 16-byte activation pieces staged per thread; 0 = activations from global
__global__ __launch_bounds__(256) void w8_gemv_f16_kernel(const f16* __restrict__ A, const int8_t* __restrict__ W,
                                                          const void* pro_delta, const void* pro_ln_weight, int N, int K,
                                                          int M, int ldw32, int lda32, const f16* __restrict__ S,
                                                          const f16* __restrict__ bias, f16* __restrict__ C, int64_t ldc,
                                                          void* pro_hout, float pro_eps, int pro_gate,
                                                          const f16* __restrict__ resid = nullptr) {
    // resid (one row, nullable): residual stream added in the epilogue, out = round(y + resid) - see w4_packed.hip
    // (argument order: the leading 14 dwords - all that the first loads need - are preloaded into SGPRs at wave
    // launch, see w4_packed_gemv_16_kernel)
    const int64_t ldw = ldw32, lda = lda32;
    const Prologue pro{pro_delta, pro_ln_weight, pro_hout, pro_eps, pro_gate};
    static_assert(PRO == PRO_NONE || ((PRO == PRO_ADDNORM || PRO == PRO_NORM) && MB == 1 && ACH > 0), "prologue: one LDS-staged row");
    constexpr bool kNorm = PRO == PRO_ADDNORM || PRO == PRO_NORM;
    constexpr bool A_LDS = ACH > 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int QW = 4 / KS;
    const int ks = wave % KS;
    const int nb_raw = (blockIdx.x * QW + wave / KS) * 4;
    const bool wave_active = nb_raw < N;
    const int nb = wave_active ? nb_raw : 0;
    const int m0 = blockIdx.y * MB;
    const int kvec = K & ~15;                  // part of K covered by 16-byte units
    const int nchunks_all = kvec >> 4;         // lane-chunks (16 k each) in a row
    const int cps = (nchunks_all + KS - 1) / KS;                   // lane-chunks per K slice
    const int c_begin = ks * cps;
    const int nchunks = min(nchunks_all, c_begin + cps);           // end of this wave's slice
    const int iters = nchunks > c_begin ? (nchunks - c_begin + 127) >> 7 : 0;    // tiles of 128 lane-chunks
    const int ppr = kvec >> 3;                 // 16-byte activation pieces per row

    const f16* arow[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) arow[m] = A + (int64_t)((m0 + m < M) ? (m0 + m) : (M - 1)) * lda;

    u32x4 areg[A_LDS ? ACH : 1];
    u32x4 xreg[kNorm ? ACH : 1], yreg[PRO == PRO_ADDNORM ? ACH : 1];   // ln weight, residual delta
    if constexpr (A_LDS) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            // unconditional (clamped) load: see w4_packed.hip
            const int c = min(tid + i * 256, MB * ppr - 1);
            const int m = MB == 1 ? 0 : c / ppr, cc = c - m * ppr;
            areg[i] = *reinterpret_cast<const u32x4*>(arow[m] + cc * 8);
            if constexpr (kNorm) xreg[i] = *reinterpret_cast<const u32x4*>((const f16*)pro.ln_weight + cc * 8);
            if constexpr (PRO == PRO_ADDNORM)
                yreg[i] = *reinterpret_cast<const u32x4*>((pro.delta ? (const f16*)pro.delta : arow[0]) + cc * 8);
        }
    }

    const int8_t* wrow[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) wrow[c] = W + (int64_t)((nb + c < N) ? (nb + c) : (N - 1)) * ldw;

    auto load_tile = [&](int it) {
        W8Tile<MB, A_LDS> tl;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int lc = c_begin + it * 128 + u * 64 + lane;
            const int lcc = lc < nchunks ? lc : (nchunks_all > 0 ? nchunks_all - 1 : 0);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                tl.w[c][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[c] + (int64_t)lcc * 16));
            if constexpr (!A_LDS) {
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        tl.a[m][2 * u + j] = *reinterpret_cast<const u32x4*>(arow[m] + (int64_t)lcc * 16 + 8 * j);
            }
        }
        return tl;
    };

    // tile 0 unconditionally (addresses are clamped; K >= 16 is the launcher's business): under `if (iters > 0)` the
    // compiler has to place the staging waits as if no tile load were outstanding, i.e. the activation staging, the
    // RMSNorm barriers and the LDS write all waited for the weight tiles to arrive from HBM
    W8Tile<MB, A_LDS> t0 = load_tile(0), t1;
    // per-channel scales of the quad: requested between the two tiles - behind tile 0 because S is a late kernel
    // argument, in front of tile 1 because tile 0's math needs them (loads return in order).  After the staging
    // barrier - where they used to be - they were a dependent global round trip once all tiles had landed.
    __builtin_amdgcn_sched_barrier(0);         // (the scheduler otherwise hoists them, and the wait for S, to the top)
    f16 s_raw[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) s_raw[c] = S[(nb + c < N) ? (nb + c) : (N - 1)];
    if (iters > 1) t1 = load_tile(1);

    // epilogue operands of the wave's channel quad requested behind the weight tiles (see w4_packed.hip)
    const bool quad_early = MB == 1 && nb + 3 < N && (((uintptr_t)bias | (uintptr_t)resid | (uintptr_t)C) & 7) == 0;
    const int nq = quad_early ? nb : 0;
    u32x2 bias_q = {0u, 0u}, resid_q = {0u, 0u};
    if (bias || resid) {                                      // kernel-uniform: plain forwards issue no extra loads
        bias_q = *reinterpret_cast<const u32x2*>((bias && quad_early ? bias : S) + nq);
        resid_q = *reinterpret_cast<const u32x2*>((resid && quad_early ? resid : S) + nq);
    }


    if constexpr (kNorm) {
        // hnew = round(x + delta), written once (block 0); staged row = round(round(hnew * r) * ln_weight)
        float* nred = reinterpret_cast<float*>(smem + (((size_t)kvec * sizeof(f16) + 15) & ~(size_t)15) + 4 * 4 * sizeof(float));
        float hv[ACH][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            unpack8<f16>(areg[i], hv[i]);
            const int c = tid + i * 256;
            if constexpr (PRO == PRO_ADDNORM) {
                if (pro.delta) {
                    float d[8];
                    unpack8<f16>(yreg[i], d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[i][e] = Act<f16>::round(hv[i][e] + d[e]);
                }
            }
            if (c < ppr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(hv[i][e], hv[i][e], ss);
                if constexpr (PRO == PRO_ADDNORM) {
                    if (blockIdx.x == 0 && pro.hout) *reinterpret_cast<u32x4*>((f16*)pro.hout + c * 8) = pack8<f16>(hv[i]);
                }
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) nred[wave] = ss;
        __syncthreads();
        const float r = rsqrtf(((nred[0] + nred[1]) + (nred[2] + nred[3])) / (float)K + pro.eps);
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            float w[8];
            unpack8<f16>(xreg[i], w);
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[i][e] = Act<f16>::round(hv[i][e] * r) * w[e];
            areg[i] = pack8<f16>(hv[i]);
        }
    }
    if constexpr (A_LDS) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int c = tid + i * 256;
            if (c < MB * ppr) {
                const int m = MB == 1 ? 0 : c / ppr, cc = c - m * ppr;
                *reinterpret_cast<u32x4*>(smem + ((int64_t)m * ppr + w8_piece_pos(cc >> 1, cc & 1)) * 16) =
                    pair_even_odd(areg[i]);
            }
        }
        __syncthreads();
    }

    float acc[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[m][c] = 0.f;

    // splice constants in registers: (t & mask) | magic is then ONE v_and_or_b32
    u32 k_mask, k_magic;
    asm volatile("s_mov_b32 %0, 0x00FF00FF" : "=s"(k_mask));
    asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(k_magic));

    h2 s2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // first use pinned HERE, behind the staging barrier: hoisted in front of it, the wait for these loads - the
        // youngest in the queue - kept the barrier back until every weight tile had landed
        f16 sh = s_raw[c];
        asm volatile("" : "+v"(sh));
        s2[c] = h2{sh, sh};
    }

    auto compute_tile = [&](const W8Tile<MB, A_LDS>& tl, int it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int lc = c_begin + it * 128 + u * 64 + lane;
            const bool valid = lc < nchunks;
            const int lcc = valid ? lc : (nchunks_all > 0 ? nchunks_all - 1 : 0);
            h2 a02[MB][4], a13[MB][4];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    u32x4 y;
                    if constexpr (A_LDS)
                        y = *reinterpret_cast<const u32x4*>(smem + ((int64_t)m * ppr + w8_piece_pos(lcc, j)) * 16);
                    else
                        y = pair_even_odd(tl.a[m][2 * u + j]);
                    a02[m][2 * j + 0] = as_h2(y[0]);
                    a13[m][2 * j + 0] = as_h2(y[1]);
                    a02[m][2 * j + 1] = as_h2(y[2]);
                    a13[m][2 * j + 1] = as_h2(y[3]);
                }
            if constexpr (STRICT) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const h2 sc2 = valid ? s2[c] : h2{(f16)0.f, (f16)0.f};   // out-of-range lanes contribute 0
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const BytePairs b = byte_pairs_f16(tl.w[c][u][j]);
                        const h2 w02 = b.p02 * sc2, w13 = b.p13 * sc2;       // rounded to fp16 (faithful)
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
                const h2 ones = {(f16)1.0f, (f16)1.0f};
                float corr[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    float e = 0.f, o = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        e = __builtin_amdgcn_fdot2(ones, a02[m][j], e, false);
                        o = __builtin_amdgcn_fdot2(ones, a13[m][j], o, false);
                    }
                    corr[m] = 1152.0f * (e + o);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float scf = valid ? (float)s2[c].x : 0.f;          // out-of-range lanes contribute 0
                    float e[MB], o[MB];
#pragma unroll
                    for (int m = 0; m < MB; ++m) e[m] = o[m] = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const u32 t = tl.w[c][u][j] ^ 0x80808080u;
                        const h2 x02 = as_h2((t & k_mask) | k_magic);         // (1152 + b0, 1152 + b2)
                        const h2 x13 = as_h2(((t >> 8) & k_mask) | k_magic);  // (1152 + b1, 1152 + b3)
#pragma unroll
                        for (int m = 0; m < MB; ++m) {
                            e[m] = __builtin_amdgcn_fdot2(x02, a02[m][j], e[m], false);
                            o[m] = __builtin_amdgcn_fdot2(x13, a13[m][j], o[m], false);
                        }
                    }
#pragma unroll
                    for (int m = 0; m < MB; ++m) acc[m][c] = __builtin_fmaf(scf, (e[m] + o[m]) - corr[m], acc[m][c]);
                }
            }
        }
    };

    // steady-state loads unconditional and real, last round peeled (see w4_packed.hip)
    const int full = iters >> 1;
    for (int r = 0; r + 1 < full; ++r) {
        compute_tile(t0, 2 * r);
        t0 = load_tile(2 * r + 2);
        compute_tile(t1, 2 * r + 1);
        t1 = load_tile(2 * r + 3);
    }
    if (full > 0) {
        compute_tile(t0, 2 * full - 2);
        if (iters & 1) t0 = load_tile(2 * full);
        compute_tile(t1, 2 * full - 1);
    }
    if (iters & 1) compute_tile(t0, iters - 1);

    // K tail (K % 16): one element per lane, done by slice 0
    if (ks == 0) {
        for (int k = kvec + lane; k < K; k += 64) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16 wq = (f16)((float)wrow[c][k] * (float)s2[c].x);
#pragma unroll
                for (int m = 0; m < MB; ++m) acc[m][c] = __builtin_fmaf((float)arow[m][k], (float)wq, acc[m][c]);
            }
        }
    }

#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[m][c] = wave_sum(acc[m][c]);

    if constexpr (KS > 1) {
        float* red = reinterpret_cast<float*>(smem + (A_LDS ? (((size_t)MB * kvec * sizeof(f16) + 15) & ~(size_t)15) : 0));
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int c = 0; c < 4; ++c) red[(wave * MB + m) * 4 + c] = acc[m][c];
        }
        __syncthreads();
        if (ks == 0 && lane == 0) {
#pragma unroll
            for (int p = 1; p < KS; ++p)
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[m][c] += red[((wave + p) * MB + m) * 4 + c];
        }
    }

    if (wave_active && ks == 0 && lane == 0) {
        float bq[4] = {0.f, 0.f, 0.f, 0.f}, rq[4] = {0.f, 0.f, 0.f, 0.f};
        if (quad_early) {
            unpack2<f16>(bias_q[0], bq[0], bq[1]);
            unpack2<f16>(bias_q[1], bq[2], bq[3]);
            unpack2<f16>(resid_q[0], rq[0], rq[1]);
            unpack2<f16>(resid_q[1], rq[2], rq[3]);
        }
        if (PRO != PRO_NONE && pro.gate_epilogue) {
            // rows of W come in (h, h, gate, gate) quads: C gets N / 2 values (chatglm_q/model.py:200-201)
            float y[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                y[c] = Act<f16>::round(acc[0][c]);
                if (bias) y[c] = Act<f16>::round(y[c] + (quad_early ? bq[c] : Act<f16>::load(bias + nb + c)));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
                Act<f16>::store(C + (nb >> 1) + i, Act<f16>::round(Act<f16>::round(y[i] / (1.0f + __expf(-y[i]))) * y[i + 2]));
            return;
        }
        if (quad_early) {                                     // one row, whole quad: one 8-byte store
            float y[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                y[c] = Act<f16>::round(acc[0][c]);
                if (bias) y[c] = resid ? Act<f16>::round(y[c] + bq[c]) : y[c] + bq[c];
                if (resid) y[c] = y[c] + rq[c];
            }
            *reinterpret_cast<u32x2*>(C + nb) = u32x2{pack2<f16>(y[0], y[1]), pack2<f16>(y[2], y[3])};
            return;
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m0 + m >= M) break;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int n = nb + c;
                if (n >= N) continue;
                if (resid) {
                    float y = Act<f16>::round(acc[m][c]);
                    if (bias) y = Act<f16>::round(y + Act<f16>::load(bias + n));
                    Act<f16>::store(C + n, y + Act<f16>::load(resid + n));
                } else {
                    store_out<f16>(C + (int64_t)(m0 + m) * ldc + n, acc[m][c], bias ? bias + n : nullptr);
                }
            }
        }
    }
}

// generic dtype (fp32 / bf16): same decomposition, per-byte dequant, activations from global
template <typename T, int MB>
__global__ __launch_bounds__(256) void w8_gemv_kernel(const T* __restrict__ A, const int8_t* __restrict__ W,
                                                      const T* __restrict__ S, const T* __restrict__ bias,
                                                      T* __restrict__ C, int M, int N, int K, int64_t ldw,
                                                      int64_t lda, int64_t ldc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = (blockIdx.x * 4 + wave) * 4;   // first of the wave's 4 output channels
    if (nb >= N) return;                          // no barriers in this kernel
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
    const int nchunks = kvec >> 4;
    const int iters = (nchunks + 63) >> 6;
    auto load_tile = [&](int it) {
        const int lc = it * 64 + lane;
        const int lcc = lc < nchunks ? lc : (nchunks > 0 ? nchunks - 1 : 0);
        struct { u32x4 w[4]; } tl;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            tl.w[c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[c] + (int64_t)lcc * 16));
        return tl;
    };
    auto compute_tile = [&](const auto& tl, int it) {
        const int lc = it * 64 + lane;
        if (lc >= nchunks) return;
        const int k = lc * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float a[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) a[m] = Act<T>::load(arow[m] + k + 4 * j + b);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int q = (int)(tl.w[c][j] << (24 - 8 * b)) >> 24;   // sign-extended byte b
                    const float wq = Act<T>::round((float)q * sc[c]);
#pragma unroll
                    for (int m = 0; m < MB; ++m) acc[m][c] = __builtin_fmaf(a[m], wq, acc[m][c]);
                }
            }
    };
    if (iters > 0) {
        auto t0 = load_tile(0);
        auto t1 = load_tile(iters > 1 ? 1 : 0);
        for (int it = 0; it < iters; it += 2) {
            compute_tile(t0, it);
            if (it + 2 < iters) t0 = load_tile(it + 2);
            if (it + 1 < iters) {
                compute_tile(t1, it + 1);
                if (it + 3 < iters) t1 = load_tile(it + 3);
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
// W8A8: C = round(acc_i32 * (a_scale[m] * w_scale[n])) (+ bias), acc = Aq (M,K) . W (N,K)^T
//   A true int8 x int8 dense GEMM on v_mfma_i32_32x32x32_i8 (exact int32 accumulation).
//   Both operands are K-contiguous int8.  Block = 4 waves side by side in N (128 output channels) x
//   BM = 32 MT rows; K step = 128 bytes.  Both tiles are staged through LDS in full 128-byte lines
//   (global -> registers one step ahead -> ds_write_b128 after the MFMAs, double buffered, one barrier per
//   step); 16-byte chunk c of row r sits at chunk position 8 r + (c ^ ((r >> 1) & 7)) so the fragment
//   reads (ds_read_b128, 128-byte pitch) are bank-conflict free.  Lane (j = lane & 31, kb = lane >> 5)
//   feeds MFMA sub-step s with chunk 4 kb + s of its row for BOTH operands (a consistent K assignment).
//   Accumulator map: column = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5).
// =============================================================================================
template <typename T, int MT, int DEPTH>
__global__ __launch_bounds__(256) void w8a8_mfma_kernel(const int8_t* __restrict__ Aq, const int8_t* __restrict__ W, int M,
                                                        int N, int K, int per, const float* __restrict__ a_scale,
                                                        const T* __restrict__ S, const T* __restrict__ bias,
                                                        T* __restrict__ C, int64_t ldc, int* __restrict__ part) {
    // (argument order: the operands of the first loads lead - preloaded into SGPRs at wave launch; scales, bias and
    // the output are needed in the epilogue only)
    constexpr int BM = 32 * MT;
    constexpr int ACH = BM * 8 / 256;          // 16-byte A chunks staged per thread per K step
    constexpr int WCH = 4;                     // 128 rows x 8 chunks / 256 threads
    __shared__ __attribute__((aligned(16))) char smem_a[2][BM * 128];
    __shared__ __attribute__((aligned(16))) char smem_w[2][128 * 128];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kb = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * 128;
    const int ksteps = (K + 127) >> 7;
    // split-K (few row tiles): this block's K steps are [k0, k0 + nst); the int32 slabs add up exactly
    const int k0 = blockIdx.z * per;
    const int nst = ksteps - k0 < per ? ksteps - k0 : per;
    auto gstep = [&](int t) { return k0 + (t < nst ? t : nst - 1); };

    const int8_t* a_src[ACH];
    int a_dst[ACH];
#pragma unroll
    for (int u = 0; u < ACH; ++u) {
        const int q = tid + u * 256, r = q >> 3, c = q & 7;
        a_src[u] = Aq + (int64_t)((m0 + r < M) ? (m0 + r) : (M - 1)) * K + c * 16;
        a_dst[u] = (r * 8 + (c ^ ((r >> 1) & 7))) * 16;
    }
    const int8_t* w_src[WCH];
    int w_dst[WCH];
#pragma unroll
    for (int u = 0; u < WCH; ++u) {
        const int q = tid + u * 256, r = q >> 3, c = q & 7;
        w_src[u] = W + (int64_t)((n0 + r < N) ? (n0 + r) : (N - 1)) * K + c * 16;
        w_dst[u] = (r * 8 + (c ^ ((r >> 1) & 7))) * 16;
    }
    const int c_mine = tid & 7;                // (tid + u * 256) & 7 is the same for every u
    const int klast = K - 16;                  // last in-bounds 16-byte chunk start

    i32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0;

    // register ring: the global loads of K step kt + DEPTH are issued when step kt starts, so DEPTH - 1 steps of
    // MFMA work cover their latency.  DEPTH 4 when the grid gives ~1 block per CU (512 x 4096 x 4096: nothing
    // else hides the latency), 2 for large grids where co-resident blocks do and registers buy occupancy.
    i32x4 a_st[DEPTH][ACH], w_st[DEPTH][WCH];
    auto load_tiles = [&](int kt, i32x4 (&ar)[ACH], i32x4 (&wr)[WCH]) {
        const int k = kt * 128 + c_mine * 16;
        const bool inb = k <= klast;
        const int off = inb ? kt * 128 : klast - c_mine * 16;      // clamped address for the K tail
#pragma unroll
        for (int u = 0; u < ACH; ++u) ar[u] = *reinterpret_cast<const i32x4*>(a_src[u] + off);
#pragma unroll
        for (int u = 0; u < WCH; ++u) wr[u] = *reinterpret_cast<const i32x4*>(w_src[u] + off);
    };
    auto store_tiles = [&](int buf, int kt, const i32x4 (&ar)[ACH], const i32x4 (&wr)[WCH]) {
        const bool inb = kt * 128 + c_mine * 16 <= klast;          // zero weights make the K tail contribute exactly 0
#pragma unroll
        for (int u = 0; u < ACH; ++u) *reinterpret_cast<i32x4*>(smem_a[buf] + a_dst[u]) = ar[u];
#pragma unroll
        for (int u = 0; u < WCH; ++u) *reinterpret_cast<i32x4*>(smem_w[buf] + w_dst[u]) = inb ? wr[u] : i32x4{0, 0, 0, 0};
    };

    const int wr = wave * 32 + j;              // this lane's weight row inside the block's W tile
    // fragments are read ONE SUB-STEP AHEAD of the MFMAs that consume them (an LDS round trip is ~100+ cycles,
    // an MFMA 32-64): reads of sub-step s + 1 are issued before the MFMAs of sub-step s
    auto read_frags = [&](int buf, int sub, i32x4 (&fa)[MT], i32x4& fb) {
        const int c = kb * 4 + sub;
        if (QL_W8A8_ABLATE & 8) {
            fb = i32x4{c, wr, buf, 1};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[mt] = i32x4{c, mt, buf, j};
            return;
        }
        fb = *reinterpret_cast<const i32x4*>(smem_w[buf] + (wr * 8 + (c ^ ((wr >> 1) & 7))) * 16);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int r = mt * 32 + j;
            fa[mt] = *reinterpret_cast<const i32x4*>(smem_a[buf] + (r * 8 + (c ^ ((r >> 1) & 7))) * 16);
        }
    };
    auto mma_step = [&](int buf) {
        i32x4 fa[2][MT], fb[2];
        read_frags(buf, 0, fa[0], fb[0]);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            if (sub < 3) read_frags(buf, sub + 1, fa[(sub + 1) & 1], fb[(sub + 1) & 1]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (QL_W8A8_ABLATE & 1) acc[mt][sub] += fa[sub & 1][mt][0] ^ fa[sub & 1][mt][1] ^ fa[sub & 1][mt][2] ^ fa[sub & 1][mt][3] ^
                                                        fb[sub & 1][0] ^ fb[sub & 1][1] ^ fb[sub & 1][2] ^ fb[sub & 1][3];
                else acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[sub & 1][mt], fb[sub & 1], acc[mt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, MT + 1, 0);        // next sub-step's DS reads first
            __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);            // then this sub-step's MFMAs
        }
    };

    // prologue: DEPTH steps in flight (a step index past the end re-reads the last step: never stored)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_tiles(gstep(d), a_st[d], w_st[d]);
    store_tiles(0, gstep(0), a_st[0], w_st[0]);
    __syncthreads();

    // steady state: every load is unconditional and real (a load under `if (more)` makes hipcc wait
    // vmcnt(0) - for the loads it just issued - before each ds_write; see w4_packed.hip)
    int kt = 0;
    for (; kt + DEPTH < nst; kt += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int buf = (kt + d) & 1;
            // slot d was stored to LDS last step; past the end this is a harmless re-read of the last step
            // (keeps the load count exact)
            if (!(QL_W8A8_ABLATE & 2)) load_tiles(gstep(kt + d + DEPTH), a_st[d], w_st[d]);
            mma_step(buf);
            if (!(QL_W8A8_ABLATE & 4)) store_tiles(buf ^ 1, gstep(kt + d + 1), a_st[(d + 1) % DEPTH], w_st[(d + 1) % DEPTH]);
            if (!(QL_W8A8_ABLATE & 16)) __syncthreads();
        }
    }
    // tail: at most DEPTH steps, no further loads
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        if (kt + d < nst) {
            const int buf = (kt + d) & 1;
            mma_step(buf);
            if (kt + d + 1 < nst) store_tiles(buf ^ 1, gstep(kt + d + 1), a_st[(d + 1) % DEPTH], w_st[(d + 1) % DEPTH]);
            __syncthreads();
        }
    }

    const int n = n0 + wave * 32 + j;
    if (part) {                                               // split-K: int32 slab, summed and scaled by the reduce kernel
        if (n < N) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int m = m0 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
                    if (m < M) part[((int64_t)blockIdx.z * M + m) * N + n] = acc[mt][i];
                }
        }
        return;
    }
    const float ws = Act<T>::load(S + (n < N ? n : N - 1));
    if constexpr (sizeof(T) == 2) {
        if ((ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
            // rounded tile through the (now idle) A-tile LDS, 16-byte row chunks to global (ql_common.h)
            T* lds_wave = reinterpret_cast<T*>(smem_w[0]) + wave * 1024;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float asc[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int m = m0 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
                    asc[i] = a_scale[m < M ? m : M - 1];
                }
                store_tile_32x32<T>(lds_wave, C, ldc, m0 + mt * 32, n0 + wave * 32, M, N, bias, lane,
                                    [&](int i) { return (float)acc[mt][i] * (asc[i] * ws); });
            }
            return;
        }
    }
    if (n < N) {
        const T* bn = bias ? bias + n : nullptr;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
                if (m >= M) continue;
                const float comb = a_scale[m] * ws;
                store_out<T>(C + (int64_t)m * ldc + n, (float)acc[mt][i] * comb, bn);
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
    return finish_launch(QL_K_W8_GENERIC);
}

struct W8Args {
    bool strict;
    const void* A;
    const int8_t* W;
    const void* S;
    const void* bias;
    void* C;
    int M, N, K;
    int64_t ldw, lda, ldc;
    hipStream_t st;
    const void* resid = nullptr;   // one-row residual epilogue (w8_gemv_residual)
};

template <typename T, int MB>
static int launch_w8_gemv_generic(const W8Args& p) {
    const int quads = (p.N + 3) / 4;
    dim3 grid((unsigned)((quads + 3) / 4), (unsigned)((p.M + MB - 1) / MB));
    w8_gemv_kernel<T, MB><<<grid, 256, 0, p.st>>>((const T*)p.A, p.W, (const T*)p.S, (const T*)p.bias, (T*)p.C, p.M,
                                                  p.N, p.K, p.ldw, p.lda, p.ldc);
    return finish_launch(QL_K_W8_GEMV);
}

template <int MB, int ACH, int KS, bool STRICT>
static int launch_w8_gemv_f16(const W8Args& p) {
    const int quads = (p.N + 3) / 4;
    constexpr int QW = 4 / KS;
    dim3 grid((unsigned)((quads + QW - 1) / QW), (unsigned)((p.M + MB - 1) / MB));
    const size_t lds = (ACH > 0 ? (((size_t)MB * (p.K & ~15) * sizeof(f16) + 15) & ~(size_t)15) : 0) +
                       (KS > 1 ? (size_t)4 * MB * 4 * sizeof(float) : 0);
    if (p.ldw > 0x7fffffff || p.lda > 0x7fffffff) return QL_ERR_UNSUPPORTED;   // strides travel as 32 bits
    w8_gemv_f16_kernel<MB, ACH, KS, STRICT><<<grid, 256, lds, p.st>>>((const f16*)p.A, p.W, nullptr, nullptr, p.N, p.K, p.M, (int)p.ldw,
                                                          (int)p.lda, (const f16*)p.S, (const f16*)p.bias, (f16*)p.C, p.ldc,
                                                          nullptr, 0.f, 0, (const f16*)p.resid);
    return finish_launch(QL_K_W8_GEMV);
}

template <int ACH, int KS, int PRO = PRO_ADDNORM>
static int launch_w8_gemv_fused(const W8Args& p, const Prologue& pro) {
    const int quads = (p.N + 3) / 4;
    constexpr int QW = 4 / KS;
    dim3 grid((unsigned)((quads + QW - 1) / QW), 1);
    // staged row, K-slice sums, the four per-wave partial sums of squares
    const size_t lds = (((size_t)(p.K & ~15) * sizeof(f16) + 15) & ~(size_t)15) + (size_t)4 * 4 * sizeof(float) + 4 * sizeof(float);
    if (p.ldw > 0x7fffffff || p.lda > 0x7fffffff) return QL_ERR_UNSUPPORTED;
    w8_gemv_f16_kernel<1, ACH, KS, false, PRO><<<grid, 256, lds, p.st>>>((const f16*)p.A, p.W, pro.delta, pro.ln_weight, p.N,
                                                                                p.K, 1, (int)p.ldw, (int)p.lda, (const f16*)p.S,
                                                                                (const f16*)p.bias, (f16*)p.C, p.ldc, pro.hout,
                                                                                pro.eps, pro.gate_epilogue);
    return finish_launch(QL_K_W8_GEMV);
}

template <int MB, bool STRICT>
static int launch_w8_gemv_f16_st(const W8Args& p) {
    const int64_t pieces = (int64_t)MB * ((p.K & ~15) / 8);
    const bool lds_ok = (size_t)MB * p.K * sizeof(f16) <= 64 * 1024 && pieces > 0;
    // split K two ways inside the block while the plain grid is small and every lane keeps >= 1 chunk
    const int forced = QL_TUNE("QLINEAR_W8_KSPLIT", 0);
    const int64_t quads = (p.N + 3) / 4;
    const bool split = forced ? forced == 2 : ((p.K >> 4) >= 128 && quads / 2 < 1024);
    if (split) {
        if (lds_ok && pieces <= 2 * 256) return launch_w8_gemv_f16<MB, 2, 2, STRICT>(p);
        if (lds_ok && pieces <= 4 * 256) return launch_w8_gemv_f16<MB, 4, 2, STRICT>(p);
        if (lds_ok && pieces <= 8 * 256) return launch_w8_gemv_f16<MB, 8, 2, STRICT>(p);
        return launch_w8_gemv_f16<MB, 0, 2, STRICT>(p);
    }
    if (lds_ok && pieces <= 2 * 256) return launch_w8_gemv_f16<MB, 2, 1, STRICT>(p);
    if (lds_ok && pieces <= 4 * 256) return launch_w8_gemv_f16<MB, 4, 1, STRICT>(p);
    if (lds_ok && pieces <= 8 * 256) return launch_w8_gemv_f16<MB, 8, 1, STRICT>(p);
    return launch_w8_gemv_f16<MB, 0, 1, STRICT>(p);
}
template <int MB>
static int launch_w8_gemv_f16_mb(const W8Args& p) {
    return p.strict ? launch_w8_gemv_f16_st<MB, true>(p) : launch_w8_gemv_f16_st<MB, false>(p);
}

// how the one-row fp16 kernel walks the weights (Prefetch, launch.h): 4 / KS channel quads of ldw bytes per workgroup
void w8_gemv_blocks(int64_t N, int64_t K, int64_t ldw, int64_t* w_block_bytes, int64_t* blocks) {
    const int forced = QL_TUNE("QLINEAR_W8_KSPLIT", 0);
    const int64_t quads = (N + 3) / 4;
    const bool split = forced ? forced == 2 : ((K >> 4) >= 128 && quads / 2 < 1024);
    const int qw = split ? 2 : 4;
    *w_block_bytes = (int64_t)qw * 4 * ldw;
    *blocks = (quads + qw - 1) / qw;
}

static int launch_w8_gemv_any(int dtype, const W8Args& p) {
    switch (dtype) {
    case QL_DTYPE_F16:
        if (p.K < 16) {                                       // no 16-byte unit in a row: the per-byte kernel
            if (p.M == 1) return launch_w8_gemv_generic<f16, 1>(p);
            if (p.M == 2) return launch_w8_gemv_generic<f16, 2>(p);
            return launch_w8_gemv_generic<f16, 4>(p);
        }
        if (p.M == 1) return launch_w8_gemv_f16_mb<1>(p);
        if (p.M == 2) return launch_w8_gemv_f16_mb<2>(p);
        return launch_w8_gemv_f16_mb<4>(p);
    case QL_DTYPE_F32:
        if (p.M == 1) return launch_w8_gemv_generic<float, 1>(p);
        if (p.M == 2) return launch_w8_gemv_generic<float, 2>(p);
        return launch_w8_gemv_generic<float, 4>(p);
    case QL_DTYPE_BF16:
        if (p.M == 1) return launch_w8_gemv_generic<__bf16, 1>(p);
        if (p.M == 2) return launch_w8_gemv_generic<__bf16, 2>(p);
        return launch_w8_gemv_generic<__bf16, 4>(p);
    default: return QL_ERR_BAD_DTYPE;
    }
}

// split-K epilogue of the W8A8 GEMM: exact int32 sum of the slabs, then the rank-1 scale, one rounding, bias
template <typename T>
__global__ __launch_bounds__(256) void w8a8_splitk_reduce_kernel(const int* __restrict__ part, const float* __restrict__ a_scale,
                                                                 const T* __restrict__ S, const T* __restrict__ bias,
                                                                 T* __restrict__ C, int M, int N, int64_t ldc, int ksplit) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    int v = 0;
    for (int s = 0; s < ksplit; ++s) v += part[((int64_t)s * M + m) * N + n];
    store_out<T>(C + (int64_t)m * ldc + n, (float)v * (a_scale[m] * Act<T>::load(S + n)), bias ? bias + n : nullptr);
}

struct W8A8Plan {
    int mt, ksplit, per;
};
// rows per block: large tiles once they still give >= ~1 block per CU, smaller ones to fill the chip; when even
// those leave most CUs idle, split K (exact: the slabs are int32)
static W8A8Plan w8a8_plan(int64_t M, int64_t N, int64_t K, size_t ws_bytes) {
    const int forced_mt = QL_TUNE("QLINEAR_W8A8_MT", 0), forced_ks = QL_TUNE("QLINEAR_W8A8_KSPLIT", 0);
    const int64_t nb = (N + 127) / 128, ksteps = (K + 127) / 128;
    int mt = (M > 64 && nb * ((M + 127) / 128) >= 256) ? 4 : M > 32 ? 2 : 1;
    if (forced_mt == 1 || forced_mt == 2 || forced_mt == 4) mt = forced_mt;
    const int64_t blocks = nb * ((M + 32 * mt - 1) / (32 * mt));
    // measured: splitting pays only while fewer than half of the CUs have a block (M <= 128 at N = 4096: 27.8 -> 22 us);
    // at 512 x 4096 x 4096 (256 blocks) the launch is bound by operand traffic L2 -> CU, not by the K chain
    int64_t ks = forced_ks > 0 ? forced_ks : (blocks < 128 ? 256 / blocks : 1);
    if (ks > 4 && forced_ks <= 0) ks = 4;
    if (ks > 8) ks = 8;
    if (ks > ksteps / 4) ks = ksteps / 4;
    while (ks > 1 && (size_t)(ks * M * N) * sizeof(int) > ws_bytes) --ks;
    if (ks < 1) ks = 1;
    const int64_t per = (ksteps + ks - 1) / ks;
    ks = (ksteps + per - 1) / per;
    return {mt, (int)ks, (int)per};
}

template <typename T, int MT>
static int launch_w8a8_mt(const int8_t* Aq, const float* a_scale, const int8_t* W, const void* S, const void* bias,
                          void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, const W8A8Plan& plan, int* ws,
                          hipStream_t st) {
    int* part = plan.ksplit > 1 ? ws : nullptr;
    dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 32 * MT - 1) / (32 * MT)), (unsigned)plan.ksplit);
    const int forced_depth = QL_TUNE("QLINEAR_W8A8_DEPTH", 0);
    if (forced_depth ? forced_depth == 4 : (int64_t)grid.x * grid.y * grid.z <= 256)
        w8a8_mfma_kernel<T, MT, 4><<<grid, 256, 0, st>>>(Aq, W, (int)M, (int)N, (int)K, plan.per, a_scale, (const T*)S,
                                                         (const T*)bias, (T*)C, ldc, part);
    else
        w8a8_mfma_kernel<T, MT, 2><<<grid, 256, 0, st>>>(Aq, W, (int)M, (int)N, (int)K, plan.per, a_scale, (const T*)S,
                                                         (const T*)bias, (T*)C, ldc, part);
    const int rc = finish_launch(QL_K_W8A8_ROWMAJOR);
    if (rc != 0 || !part) return rc;
    const int64_t total = M * N;
    w8a8_splitk_reduce_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(part, a_scale, (const T*)S, (const T*)bias,
                                                                                 (T*)C, (int)M, (int)N, ldc, plan.ksplit);
    return finish_launch(QL_K_SPLITK_REDUCE);
}

template <typename T>
static int launch_w8a8(const int8_t* Aq, const float* a_scale, const int8_t* W, const void* S, const void* bias,
                       void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st) {
    const W8A8Plan plan = w8a8_plan(M, N, K, ws && ((uintptr_t)ws & 15) == 0 ? ws_bytes : 0);
    if (plan.mt == 4) return launch_w8a8_mt<T, 4>(Aq, a_scale, W, S, bias, C, M, N, K, ldc, plan, (int*)ws, st);
    if (plan.mt == 2) return launch_w8a8_mt<T, 2>(Aq, a_scale, W, S, bias, C, M, N, K, ldc, plan, (int*)ws, st);
    return launch_w8a8_mt<T, 1>(Aq, a_scale, W, S, bias, C, M, N, K, ldc, plan, (int*)ws, st);
}

size_t w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    const W8A8Plan p = w8a8_plan(M, N, K, (size_t)-1);
    return p.ksplit > 1 ? (size_t)(p.ksplit * M * N) * sizeof(int) : 0;
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
            int64_t N, int64_t K, int64_t ldw, int64_t lda, int64_t ldc, bool strict, hipStream_t st) {
    const W8Args p{strict, A, W, S, bias, C, (int)M, (int)N, (int)K, ldw, lda, ldc, st};
    return launch_w8_gemv_any(dtype, p);
}
int w8_gemv_fused(int dtype, bool gate_epilogue, const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t N,
                  int64_t K, int64_t ldw, const void* delta, const void* ln_weight, void* hout, float eps, hipStream_t st) {
    if (dtype != QL_DTYPE_F16) return QL_ERR_BAD_DTYPE;
    const int64_t pieces = K / 8;
    if (K % 16 != 0 || pieces > 8 * 256 || (size_t)K * sizeof(f16) > 60 * 1024) return QL_ERR_UNSUPPORTED;
    const W8Args p{false, A, W, S, bias, C, 1, (int)N, (int)K, ldw, K, N, st};
    const Prologue pro{delta, ln_weight, hout, eps, gate_epilogue ? 1 : 0};
    const int64_t quads = (N + 3) / 4;
    const bool split = (K >> 4) >= 128 && quads / 2 < 1024;    // the unfused kernel's rule
    const bool plain_norm = !delta && !hout;                   // nothing to add, nothing to write back
#define QL_W8_FUSED(ACH_, KS_) \
    return plain_norm ? launch_w8_gemv_fused<ACH_, KS_, PRO_NORM>(p, pro) : launch_w8_gemv_fused<ACH_, KS_, PRO_ADDNORM>(p, pro);
    if (split) {
        if (pieces <= 2 * 256) { QL_W8_FUSED(2, 2) }
        if (pieces <= 4 * 256) { QL_W8_FUSED(4, 2) }
        QL_W8_FUSED(8, 2)
    }
    if (pieces <= 2 * 256) { QL_W8_FUSED(2, 1) }
    if (pieces <= 4 * 256) { QL_W8_FUSED(4, 1) }
    QL_W8_FUSED(8, 1)
#undef QL_W8_FUSED
}

// one-row fp16 forward whose output is added to the residual stream: C = round(y + resid)
int w8_gemv_residual(int dtype, const void* A, const int8_t* W, const void* S, const void* bias, const void* resid, void* C,
                     int64_t N, int64_t K, int64_t ldw, hipStream_t st) {
    if (dtype != QL_DTYPE_F16) return QL_ERR_BAD_DTYPE;
    W8Args p{false, A, W, S, bias, C, 1, (int)N, (int)K, ldw, K, N, st};
    p.resid = resid;
    return launch_w8_gemv_f16_mb<1>(p);
}
int w8a8_gemm(int dtype, const int8_t* Aq, const float* a_scale, const int8_t* W, const void* S, const void* bias,
              void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st) {
    QL_DISPATCH_DTYPE(dtype, launch_w8a8, Aq, a_scale, W, S, bias, C, M, N, K, ldc, ws, ws_bytes, st)
}

}  // namespace ql
