// int4g32 forward for 2..4 activation rows on the 4x4x4 matrix instruction (gfx950), derived layout part 1.
//
// Why: the VALU GEMV (w4_packed.hip) pays one v_dot2c per weight pair AND ROW - two rows cost 46 % more than one
// (w_in: 20 us against 13.7), four rows 3.5x - and the 32-row MFMA tiles of w4_fewrow.hip pay for 32 rows whatever the
// count (w_in: 20 - 22 us at 3..16 rows).  v_mfma_f32_4x4x4_16b_{f16,bf16} is 16 independent 4 x 4 x 4 products: block
// b = lane / 4; lane 4 b + i supplies row i of A (4 values of k), lane 4 b + j column j of B, and lane 4 b + j receives
// D[0..3][j] - FOUR rows for one instruction, with the weights of column j never leaving the lane that unpacked them
// (mapping probed on the hardware: tools/microbench/mfma4_probe.hip).  Mapping used here: block b <-> one K group
// (16 consecutive groups per wave step), lane j of a block <-> column j of the wave's column quad; per step a lane
// loads ONE 16-byte unit (its column's 32 nibbles of its group) and issues 8 MFMAs (4 k each).
// Arithmetic = the GEMV's default mode (exact dequant: no per-weight rounding, fp32 accumulation, rounded once at the
// end; include/qlinear_hip.h QL_FLAG_STRICT_ROUNDING explains the two modes):
//   fp16: (n - 8) is formed exactly in fp16 from the exponent splice (0x6400 | n) - 1032 resp. (0x6400 | 16 n) / 16 - 72
//   bf16: no packed bf16 arithmetic on this chip - the splice 128 + n goes into the MFMA as it is and a second MFMA with
//         B = 1 accumulates the row's activation sum; (sum a (128 + n)) - 136 (sum a) = sum a (n - 8), in fp32.
// Reference semantic: chatglm_q/int4/triton_ops.py:66-80 (nibble decode, group scale, fp32 accumulate, cast), bias added
// as a second rounded operation (chatglm_q/int4/qlinear.py:90-94).
#include <cstdlib>

#include "launch.h"
#include "ql_common.h"

// Build-time switches (tools/ab).  QL_ROWS4_ABLATE bit 0 = no arithmetic (loaded words are folded into the sums as they are),
// bit 1 = no activation reads from LDS: with both, w_in still takes 14.1 us at 2 rows and 17.6 at 4 (4 waves per block) -
// the kernel is bound by its skeleton, above all by every block staging its own copy of the M activation rows (w_in:
// 1 712 blocks x M x 8 KB = 56 MB at 4 rows, as much as the weights), not by the matrix or vector pipes.  Hence
// QL_ROWS4_WAVES = 8 waves per block sharing one staged copy, and QL_ROWS4_RING = 2 weight steps in flight per wave
// (more costs occupancy).  Measured, us at 2 / 4 rows for qkv, o, w_in (fp16):
//   4 waves, ring 4: 6.5 5.8 17.1 / 7.3 6.5 19.1      8 waves, ring 4: 6.9 5.5 17.4 / 7.3 6.0 18.1     16 waves, ring 2: 6.0 6.0 17.0 / 6.3 6.1 17.2
//   4 waves, ring 2: 6.6 5.6 16.9 / 7.3 6.1 18.4      8 waves, ring 2: 5.9 5.2 15.9 / 6.9 5.5 17.1     4 waves, ring 8: 10.0 6.9 21.2 / 11.6 7.7 23.4
#ifndef QL_ROWS4_ABLATE
#define QL_ROWS4_ABLATE 0
#endif
#ifndef QL_ROWS4_RING
#define QL_ROWS4_RING 2
#endif
#ifndef QL_ROWS4_WAVES
#define QL_ROWS4_WAVES 8          // waves per block: they share ONE staged copy of the activation rows
#endif

namespace ql {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Rows4;
template <> struct Rows4<f16> {
    static __device__ __forceinline__ f32x4 mma(u32 a0, u32 a1, u32 b0, u32 b1, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(f16x4, u32x2{a0, a1}), __builtin_bit_cast(f16x4, u32x2{b0, b1}), c, 0, 0, 0);
    }
};
template <> struct Rows4<__bf16> {
    static __device__ __forceinline__ f32x4 mma(u32 a0, u32 a1, u32 b0, u32 b1, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, u32x2{a0, a1}), __builtin_bit_cast(s16x4, u32x2{b0, b1}), c, 0, 0, 0);
    }
};

// KS: K slices per column quad inside the block (its NW waves = NW / KS quad sets x KS slices), as in the GEMV.
// NQ: column quads per wave: they share the activation fragments (LDS reads per weight / NQ), the staged rows and the
// block's fixed costs (staging, barriers, reduction) are spread over NQ x the columns.
// MR: groups of 4 rows per weight fragment (only 1 is instantiated, see w4_rows4_supported).
// gate (run time): the packed columns come in quads (h_2t, h_2t+1, gate_2t, gate_2t+1) - the gate-interleaved copy of a
// first MLP projection - and C gets N / 2 columns, C[m, 2t+i] = round(round(silu(y_i)) * y_{i+2}), y = rounded sum (+ bias,
// rounded): chatglm_q/model.py:200-201, the rounding sequence of the GEMV's and the few-row kernel's gate epilogues.
// PRO = 1: the rows are staged through the residual add + RMSNorm of the model graph (chatglm_q/model.py:62-73,243-245),
//   hnew = round(A + delta) (delta nullable; written to hout by block 0), staged = round(round(hnew * rsqrt(mean(hnew^2) + eps)) *
//   ln_weight) - with the summation order of rmsnorm_kernel (decode_ops.hip: 256 threads per row, chunks tid + 256 u, wave
//   sums combined as (0 + 1) + (2 + 3)), so that the fused launch is bit-equal to qlinear_add_rmsnorm + the projection.
template <typename T, int KS, int RING, int NQ, int NW, int MR, int PRO>
__global__ __launch_bounds__(NW * 64) void w4_rows4_kernel(const T* __restrict__ A, const u32x4* __restrict__ Wt, const T* __restrict__ Sp,
                                                           int N, int K, int M, int lda32, const T* __restrict__ bias,
                                                           T* __restrict__ C, int64_t ldc, int gate, const T* __restrict__ delta,
                                                           const T* __restrict__ ln_weight, T* __restrict__ hout, float eps) {
    constexpr bool kF16 = Act<T>::code == QL_DTYPE_F16;
    constexpr int QW = NW / KS;                                // quad sets per block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int G = K >> 5, cpr = K >> 3;
    const int64_t lda = lda32;
    const int rowb = K * 2 + 16;                               // staged row + 16 bytes: the 4 rows of a block land in 4 bank groups
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane >> 2, x = lane & 3;                     // x: row index as A supplier, column index as B supplier / D owner
    const int ks = wave % KS;
    const int quads = (N + 3) >> 2;
    const int t0 = (blockIdx.x * QW + wave / KS) * NQ;         // first quad of this wave
    const int gs = (G + KS - 1) / KS, g_begin = ks * gs;
    const int g_end = G < g_begin + gs ? G : g_begin + gs;
    const int iters = g_end > g_begin ? (g_end - g_begin + 15) >> 4 : 0;

    // weight / scale ring first: independent of the activations, RING steps in flight before anything waits
    const u32x4* wp[NQ];
    const T* sp[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const int t = t0 + n < quads ? t0 + n : 0;             // quads past the end shadow quad 0 (stores are masked)
        wp[n] = Wt + (int64_t)(t * 4 + x) * G;
        sp[n] = Sp + (int64_t)t * G * 4 + x;
    }
    auto group_of = [&](int it) {
        const int g = g_begin + it * 16 + b;
        return g < g_end ? g : (g_end > g_begin ? g_end - 1 : 0);
    };
    u32x4 wr[RING][NQ];
    T sr[RING][NQ];
    auto load_w = [&](int it, int slot) {
        const int gc = group_of(it < iters ? it : (iters > 0 ? iters - 1 : 0));
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            wr[slot][n] = __builtin_nontemporal_load(wp[n] + gc);
            sr[slot][n] = sp[n][(int64_t)gc * 4];
        }
    };
#pragma unroll
    for (int r = 0; r < RING; ++r) load_w(r, r);

    // stage the M rows: 16-byte chunk cc of row m (k = 8 cc .. 8 cc + 7) at m * rowb + 16 cc, its dwords in the order
    // (0, 2, 1, 3): the two dwords an MFMA takes as its A operand are then an aligned register pair of the ds_read_b128.
    // No swizzle: the 16 lanes of a ds_read_b128 phase are 4 blocks (64-byte stride) x 4 rows (16-byte row padding).
    if constexpr (PRO == 1) {
        static_assert(NW == 8, "two rows at a time, 256 threads each (rmsnorm_kernel's summation order)");
        constexpr int VPT = 4;                                 // K <= 8192
        float* nred = reinterpret_cast<float*>(smem + (int64_t)M * rowb);
        const int half = tid >> 8, ht = tid & 255;
        for (int mp = 0; mp < M; mp += 2) {                    // block-uniform trip count
            const int m = mp + half;
            const bool live = m < M;
            float v[VPT][8];
            float ss = 0.f;
#pragma unroll
            for (int u = 0; u < VPT; ++u) {
                const int i = ht + u * 256;
                if (live && i < cpr) {
                    unpack8<T>(*reinterpret_cast<const u32x4*>(A + (int64_t)m * lda + i * 8), v[u]);
                    if (delta) {
                        float d[8];
                        unpack8<T>(*reinterpret_cast<const u32x4*>(delta + (int64_t)m * K + i * 8), d);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[u][e] = Act<T>::round(v[u][e] + d[e]);
                        if (blockIdx.x == 0 && hout) *reinterpret_cast<u32x4*>(hout + (int64_t)m * K + i * 8) = pack8<T>(v[u]);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(v[u][e], v[u][e], ss);
                }
            }
            ss = wave_sum(ss);
            if (lane == 0) nred[wave] = ss;
            __syncthreads();
            const float tot = half ? (nred[4] + nred[5]) + (nred[6] + nred[7]) : (nred[0] + nred[1]) + (nred[2] + nred[3]);
            const float r = rsqrtf(tot / (float)K + eps);
#pragma unroll
            for (int u = 0; u < VPT; ++u) {
                const int i = ht + u * 256;
                if (live && i < cpr) {
                    float w[8], y[8];
                    unpack8<T>(*reinterpret_cast<const u32x4*>(ln_weight + i * 8), w);
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = Act<T>::round(v[u][e] * r) * w[e];
                    const u32x4 pk = pack8<T>(y);
                    *reinterpret_cast<u32x4*>(smem + (int64_t)m * rowb + i * 16) = u32x4{pk[0], pk[2], pk[1], pk[3]};
                }
            }
            __syncthreads();
        }
    } else {
        for (int c = tid; c < M * cpr; c += NW * 64) {
            const int m = c / cpr, cc = c - m * cpr;
            const u32x4 v = *reinterpret_cast<const u32x4*>(A + (int64_t)m * lda + cc * 8);
            *reinterpret_cast<u32x4*>(smem + (int64_t)m * rowb + cc * 16) = u32x4{v[0], v[2], v[1], v[3]};
        }
        __syncthreads();
    }

    const char* arow[MR];                                      // rows past M repeat the last one (their sums are dropped)
#pragma unroll
    for (int r = 0; r < MR; ++r) arow[r] = smem + (int64_t)(4 * r + x < M ? 4 * r + x : M - 1) * rowb;
    u32 k_lo, k_hi, k_magic;
    asm volatile("s_mov_b32 %0, 0x000F000F" : "=s"(k_lo));
    asm volatile("s_mov_b32 %0, 0x00F000F0" : "=s"(k_hi));
    asm volatile("v_mov_b32 %0, %1" : "=v"(k_magic) : "i"(kF16 ? 0x64006400 : 0x43004300));
    const h2 k1032 = {(f16)1032.0f, (f16)1032.0f}, kInv16 = {(f16)0.0625f, (f16)0.0625f}, kM72 = {(f16)-72.0f, (f16)-72.0f};
    const u32 kOnes = 0x3F803F80u;                             // bf16 (1, 1)

    f32x4 acc[NQ][MR];
#pragma unroll
    for (int n = 0; n < NQ; ++n)
#pragma unroll
        for (int r = 0; r < MR; ++r) acc[n][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the activation chunks of a step are requested one step AHEAD (two register sets, alternating): issued in front of
    // each MFMA they put a full LDS round trip on every one of them
    auto load_a = [&](int it, u32x4 (&av)[MR][4]) {
        const int gc = group_of(it < iters ? it : (iters > 0 ? iters - 1 : 0));
#pragma unroll
        for (int r = 0; r < MR; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (QL_ROWS4_ABLATE & 2) av[r][q] = u32x4{(u32)gc, (u32)q, 0x3c003c00u, 0x3c003c00u};
                else av[r][q] = *reinterpret_cast<const u32x4*>(arow[r] + gc * 64 + q * 16);
            }
    };
    auto step = [&](int it, int slot, const u32x4 (&av)[MR][4]) {
        const int g = g_begin + it * 16 + b;
        if (QL_ROWS4_ABLATE & 1) {
#pragma unroll
            for (int n = 0; n < NQ; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[n][0][q] += u32_as_f32((wr[slot][n][q] ^ av[0][q][0] ^ av[0][q][1] ^ av[0][q][2] ^ av[0][q][3]) & 0x3fffffffu) + (float)sr[slot][n];
            return;
        }
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 osum[MR];                                        // bf16: the rows' activation sums, shared by the NQ quads
        if constexpr (!kF16) {
#pragma unroll
            for (int r = 0; r < MR; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    osum[r] = Rows4<T>::mma(av[r][q][0], av[r][q][1], kOnes, kOnes, q == 0 ? zero : osum[r]);
                    osum[r] = Rows4<T>::mma(av[r][q][2], av[r][q][3], kOnes, kOnes, osum[r]);
                }
        }
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            const u32x4 wv = wr[slot][n];
            const float sc = g < g_end ? (float)sr[slot][n] : 0.f;       // lanes past the slice contribute 0
            f32x4 e[MR], o[MR];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // word q holds k = 8 q .. 8 q + 7 at nibble positions p(k) = (k >> 1) + 4 (k & 1)  [w4_repack_kernel]:
                // (w & 0x000F000F) = nibbles of k (0, 1), ((w >> 8) & 0x000F000F) = k (4, 5): one MFMA over k {0, 1, 4, 5}
                // with dwords 0 and 2 of the activation chunk (staged as registers 0, 1); the 0x00F000F0 masks give k (2, 3)
                // and (6, 7): dwords 1 and 3 (registers 2, 3)
                const u32 w = wv[q], w8 = w >> 8;
                u32 b0, b1, b2, b3;
                if constexpr (kF16) {
                    b0 = as_u32(as_h2((w & k_lo) | k_magic) - k1032);
                    b1 = as_u32(as_h2((w8 & k_lo) | k_magic) - k1032);
                    b2 = as_u32(as_h2((w & k_hi) | k_magic) * kInv16 + kM72);
                    b3 = as_u32(as_h2((w8 & k_hi) | k_magic) * kInv16 + kM72);
                } else {
                    // bf16: 128 + n as spliced
                    b0 = (w & k_lo) | k_magic, b1 = (w8 & k_lo) | k_magic;
                    b2 = ((w >> 4) & k_lo) | k_magic, b3 = ((w8 >> 4) & k_lo) | k_magic;
                }
#pragma unroll
                for (int r = 0; r < MR; ++r) {
                    e[r] = Rows4<T>::mma(av[r][q][0], av[r][q][1], b0, b1, q == 0 ? zero : e[r]);
                    o[r] = Rows4<T>::mma(av[r][q][2], av[r][q][3], b2, b3, q == 0 ? zero : o[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < MR; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float grp = kF16 ? e[r][i] + o[r][i] : __builtin_fmaf(-136.0f, osum[r][i], e[r][i] + o[r][i]);
                    acc[n][r][i] = __builtin_fmaf(sc, grp, acc[n][r][i]);
                }
        }
    };

    // ring walk: slot r serves steps r, r + RING, ...; the refill is issued right behind the step that freed the slot.
    // Whole rounds only contain unconditional loads (hipcc then counts its vmcnt waits); the last round is peeled.
    static_assert(RING % 2 == 0, "the activation register sets alternate with the step parity");
    u32x4 av[2][MR][4];
    load_a(0, av[0]);
    int it = 0;
    for (; it + 2 * RING <= iters; it += RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            load_a(it + r + 1, av[(r + 1) & 1]);
            step(it + r, r, av[r & 1]);
            load_w(it + r + RING, r);
        }
    }
#pragma unroll
    for (int r = 0; r < 2 * RING - 1; ++r) {
        if (it + r < iters) {
            load_a(it + r + 1, av[(r + 1) & 1]);
            step(it + r, r % RING, av[r & 1]);
            if (it + r + RING < iters) load_w(it + r + RING, r % RING);
        }
    }

    // sum over the 16 blocks of the wave (lanes of equal x): DPP inside each 16-lane row, the 4 rows through LDS -
    // together with the K slices of the other waves
#pragma unroll
    for (int n = 0; n < NQ; ++n)
#pragma unroll
        for (int r = 0; r < MR; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[n][r][i] += dpp_move<0x128>(acc[n][r][i]);  // row_ror:8
                acc[n][r][i] += dpp_move<0x124>(acc[n][r][i]);  // row_ror:4
            }
    __syncthreads();                                            // every wave is done with the staged rows
    float* red = reinterpret_cast<float*>(smem);               // [wave][quad of the wave][row group][16-lane row][x][i]
    if ((lane & 12) == 0) {
#pragma unroll
        for (int n = 0; n < NQ; ++n)
#pragma unroll
            for (int r = 0; r < MR; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) red[((((wave * NQ + n) * MR + r) * 4 + (lane >> 4)) * 4 + x) * 4 + i] = acc[n][r][i];
    }
    __syncthreads();
    auto total = [&](int qs, int n, int r, int j, int i) {      // sum over the K slices and the four 16-lane rows
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
            for (int l = 0; l < 4; ++l) s += red[(((((qs * KS + k) * NQ + n) * MR + r) * 4 + l) * 4 + j) * 4 + i];
        return s;
    };
    for (int o = tid; o < QW * NQ * MR * 16; o += NW * 64) {
        const int i = o & 3, j = (o >> 2) & 3, r = (o >> 4) % MR, n = (o / (16 * MR)) % NQ, qs = o / (16 * MR * NQ);
        const int m = 4 * r + i, quad = (blockIdx.x * QW + qs) * NQ + n, col = quad * 4 + j;
        if (m >= M || col >= N) continue;
        if (!gate) {
            store_out<T>(C + (int64_t)m * ldc + col, total(qs, n, r, j, i), bias ? bias + col : nullptr);
        } else if (j < 2) {
            float yh = Act<T>::round(total(qs, n, r, j, i)), yg = Act<T>::round(total(qs, n, r, j + 2, i));
            if (bias) {
                yh = Act<T>::round(yh + Act<T>::load(bias + col));
                yg = Act<T>::round(yg + Act<T>::load(bias + col + 2));
            }
            Act<T>::store(C + (int64_t)m * ldc + quad * 2 + j, Act<T>::round(Act<T>::round(yh / (1.0f + __expf(-yh))) * yg));
        }
    }
}

struct Rows4Pro {
    const void* delta;
    const void* ln_weight;     // non-null selects the prologue
    void* hout;
    float eps;
};

template <typename T, int KS, int NQ, int MR, int PRO>
static int launch_rows4_ks(const void* A, const void* packed, const void* bias, void* C, int M, int N, int K, int64_t lda,
                           int64_t ldc, bool gate, const Rows4Pro& pro, hipStream_t st) {
    const int64_t G = K / 32, Npad = (N + 3) & ~(int64_t)3;
    const u32x4* Wt = (const u32x4*)packed;
    const T* Sp = (const T*)((const char*)packed + Npad * G * 16);
    constexpr int NW = QL_ROWS4_WAVES, QW = NW / KS, RING = QL_ROWS4_RING;
    const int quads = (int)(Npad / 4), per_block = QW * NQ;
    size_t lds = (size_t)M * (K * 2 + 16) + 64;                                 // + the prologue's eight wave sums
    if (lds < (size_t)NW * NQ * MR * 256) lds = (size_t)NW * NQ * MR * 256;     // the reduction scratch
    static bool attr_set = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&w4_rows4_kernel<T, KS, RING, NQ, NW, MR, PRO>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    }();
    (void)attr_set;
    w4_rows4_kernel<T, KS, RING, NQ, NW, MR, PRO><<<(unsigned)((quads + per_block - 1) / per_block), NW * 64, lds, st>>>(
        (const T*)A, Wt, Sp, N, K, M, (int)lda, (const T*)bias, (T*)C, ldc, gate ? 1 : 0, (const T*)pro.delta, (const T*)pro.ln_weight,
        (T*)pro.hout, pro.eps);
    return finish_launch(QL_K_W4_ROWS4);
}

bool w4_rows4_supported(int dtype, int64_t M, int64_t N, int64_t K, int64_t lda) {
    // the M staged rows (+ 16 bytes each) live in LDS; one block per CU is the floor for the widest layer
    // (two row groups per weight fragment - MR = 2, 5..8 rows - were built and measured: qkv 7.6 - 8.1 us against the few-row
    // kernel's 8.9 - 9.3, but w_in 20.9 - 22.1 against 19.2 - 19.3 and batch-8 decode 1.91 against 1.78 ms per step: not instantiated)
    return (dtype == QL_DTYPE_F16 || dtype == QL_DTYPE_BF16) && M >= 1 && M <= 4 && K % 32 == 0 && K >= 32 && lda <= 0x7fffffff &&
           (size_t)M * (K * 2 + 16) <= 150 * 1024;
}

// ks: K slices per column quad (1, 2 or 4: w4_packed.hip's choose_ksplit); gate: SiLU * gate epilogue on a gate-interleaved
// copy (N % 4 == 0, C gets N / 2 columns); ln_weight != nullptr: residual add + RMSNorm prologue (rows contiguous, K <= 8192)
int w4_rows4(int dtype, int ks, const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
             int64_t lda, int64_t ldc, hipStream_t st, bool gate, const void* delta, const void* ln_weight, void* hout, float eps) {
    const Rows4Pro pro{delta, ln_weight, hout, eps};
#define QL_R4K(T_, NQ_, PRO_)                                                                                               \
    switch (ks) {                                                                                                           \
    case 4: return launch_rows4_ks<T_, 4, NQ_, 1, PRO_>(A, packed, bias, C, (int)M, (int)N, (int)K, lda, ldc, gate, pro, st);   \
    case 2: return launch_rows4_ks<T_, 2, NQ_, 1, PRO_>(A, packed, bias, C, (int)M, (int)N, (int)K, lda, ldc, gate, pro, st);   \
    default: return launch_rows4_ks<T_, 1, NQ_, 1, PRO_>(A, packed, bias, C, (int)M, (int)N, (int)K, lda, ldc, gate, pro, st);  \
    }
#define QL_R4(T_)                                                                                                       \
    if (ln_weight) { QL_R4K(T_, 1, 1) }                                                                                 \
    if (nq == 2) { QL_R4K(T_, 2, 0) }                                                                                   \
    QL_R4K(T_, 1, 0)
    // one quad per wave; two (QLINEAR_ROWS4_NQ=2: shared activation fragments, half the blocks) measured slower on every
    // layer shape but w_out at one row (2 rows: 7.2 / 5.8 / 18.4 / 13.3 us against 6.5 / 5.8 / 17.1 / 11.8)
    const int nq = QL_TUNE("QLINEAR_ROWS4_NQ", 1) == 2 ? 2 : 1;
    if (gate && N % 4 != 0) return QL_ERR_BAD_SHAPE;
    if (ln_weight && (K > 8192 || QL_ROWS4_WAVES != 8)) return QL_ERR_UNSUPPORTED;
    if (dtype == QL_DTYPE_F16) { QL_R4(f16) }
    if (dtype == QL_DTYPE_BF16) { QL_R4(__bf16) }
#undef QL_R4
#undef QL_R4K
    return QL_ERR_BAD_DTYPE;
}

}  // namespace ql
