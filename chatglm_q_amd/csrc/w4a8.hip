// W4A8: int4 group-32 weights x int8-quantised activations on the i8 matrix cores (gfx950).
//
// SURVEY.md 8d config 5 / BASELINE configs[4] ("large-M int4 GEMM, MFMA-eligible after act-quant"): the activation
// semantic is the one of the int8 path - row-wise symmetric quantisation, quantize_int8 (chatglm_q/int8/quantizer.py:
// 11-19), integer matmul + scale epilogue (chatglm_q/int8/qlinear.py:60-62) - the weight decode is the int4 one,
// (nibble - 8) * scale[group, column] (chatglm_q/int4/triton_ops.py:71-73, chatglm_q/int4/qlinear.py:20-33):
//
//      C[m, n] = round( a_scale[m] * sum_g  s[g, n] * ( sum_{k in g} Aq[m, k] * (nib[k, n] - 8) ) )   (+ bias)
//
// A group is 32 deep = ONE v_mfma_i32_32x32x32_i8: every MFMA starts a fresh sum, whose exact int32 result is converted
// and folded into the fp32 accumulator with the group's scale - the price of group-wise scales on an integer
// contraction.  One v_cvt_f32_i32 + one v_fma_f32 per accumulator element would be 32 VALU instructions per MFMA; the
// fold used here starts the MFMA from the bit pattern of 1.5 * 2^23 and needs 8 v_pk_add_f32 + 8 v_pk_fma_f32 (see
// `fold`), still twice the MFMA's own 32 cycles: VALU-bound by construction at half the i8 MFMA peak at best.
// The integer stage is exact; the result differs from the weight-only W4A16 path by the activation quantisation error
// (~1e-2 relative, reported by tests/test_parity_gpu.py, not claimed as parity with the Triton reference).
//
// Derived weight layout "a8" (qlinear_w4a8_pack, built lazily like the other derived layouts):
//   Wa[ct][kt][lane][16 B]  ct = n / 32, kt = 64-deep K step, lane = 32 kb + j: column 32 ct + j;
//                           bytes 0..7  = group 2 kt,     k = 16 kb .. 16 kb + 15 of the group  (two dwords of 8 nibbles)
//                           bytes 8..15 = group 2 kt + 1, same
//                           nibble position p of a dword holds k_local = (p >> 1) + 4 (p & 1), so that
//                             ((w << 4) & 0xF0F0F0F0) ^ 0x80808080  = bytes 16 (n - 8) for k_local 0..3   (3 VALU)
//                             ( w       & 0xF0F0F0F0) ^ 0x80808080  = ... for k_local 4..7                (2 VALU)
//                           are two dwords of the MFMA B operand in natural k order; the factor 16 is folded into the scale.
//   Sa[ct][kt][j][2]        the two groups' scales of column 32 ct + j (activation dtype).
//   Columns past N / a missing last group: nibble 8 (= 0), scale 0.
// Kernel skeleton = w8a8_tiled_kernel (w8a8.hip): 8 waves = two K-parity groups x four column tiles, A through three LDS
// buffers with cross-barrier fragment prefetch, W straight from global into the MFMA operand path.
#include <type_traits>

#include "launch.h"
#include "ql_common.h"

// Build-time switches, each measured on 8192 x {4608, 4096, 27392} x 4096 and 8192 x 4096 x 13696 (tools/ab, W4A8 GEMM alone,
// TOP/s; the f16-MFMA path on the same shapes: 924 / 974 / 995 / 1044):
//   QL_W4A8_PKFOLD  packed fold on a magic-biased accumulator (below)          0: 777 / 784 / 824 / 927   1: 803 / 827 / 825 / 991
//   QL_W4A8_DEPTH4  weight register stages of the 128-row variants (with 1)    2: as above   1: 891 / 916 / 912 / 1016   3: 861 / 883 / 860 / 961
//   QL_W4A8_DA4     A-chunk register stages of the 128-row variants            1 (2 spills: 16 registers per stage)
//   QL_W4A8_ROLL    one fragment buffer refilled in place instead of two       neutral (807 / 840 / 834 / 973 with PKFOLD, depth 2)
// and, at run time, QLINEAR_W4A8_COLG=1: 8-wave blocks of 256 columns sharing one A tile (column groups instead of K-parity
// groups): 807 / 902 / 898 / 1000 - half the A bytes per flop buy nothing, this kernel is not bound by operand delivery.
// PMC (rocprofv3, tools/prof_pmc.sh, 8192 x 4096 x 4096, PKFOLD 1 / DEPTH4 1): 24 VALU instructions per MFMA (16 of them the
// fold), VALU busy 64 % and the matrix pipe 21 % of the cycles at 1.94 GHz; the f16-MFMA kernel on the same shape: 6 VALU per
// (half-K) MFMA, matrix pipe busy 56 % at a clock throttled to 1.63 GHz.
#ifndef QL_W4A8_PKFOLD
#define QL_W4A8_PKFOLD 1
#endif
#ifndef QL_W4A8_ROLL
#define QL_W4A8_ROLL 0
#endif
#ifndef QL_W4A8_DEPTH4
#define QL_W4A8_DEPTH4 1          // register-stage depth (weights) of the 128-row variants
#endif
#ifndef QL_W4A8_DA4
#define QL_W4A8_DA4 1             // register-stage depth of the A chunks of the 128-row variants (16 registers each)
#endif

namespace ql {

constexpr int gcd2(int a, int b) { return b == 0 ? a : gcd2(b, a % b); }
constexpr int lcm3(int a, int b, int c) { return (a * b / gcd2(a, b)) * c / gcd2(a * b / gcd2(a, b), c); }

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct W4A8Layout {
    int64_t ctiles, ksteps;
    size_t off_s, bytes;
};
static inline W4A8Layout w4a8_layout(int64_t N, int64_t K, size_t esize) {
    W4A8Layout L;
    L.ctiles = (N + 31) / 32;
    L.ksteps = (K / 32 + 1) / 2;
    L.off_s = (size_t)(L.ctiles * L.ksteps) * 1024;
    L.bytes = L.off_s + (size_t)(L.ctiles * L.ksteps) * 64 * esize;
    return L;
}
size_t w4a8_packed_bytes(int64_t N, int64_t K, int dtype) { return w4a8_layout(N, K, dtype == QL_DTYPE_F32 ? 4 : 2).bytes; }

// one thread per (ct, kt, lane): 16 bytes of weights; lanes < 32 also write their column's two scales
template <typename T>
__global__ __launch_bounds__(256) void w4a8_pack_kernel(const uint8_t* __restrict__ Wq, const T* __restrict__ S,
                                                        u32x4* __restrict__ Wa, T* __restrict__ Sa, int N, int G, int ksteps,
                                                        int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (ct * ksteps + kt) * 64 + lane
    if (idx >= total) return;
    const int lane = (int)(idx & 63), j = lane & 31, kb = lane >> 5;
    const int64_t step = idx >> 6;
    const int kt = (int)(step % ksteps), ct = (int)(step / ksteps);
    const int n = ct * 32 + j;
    u32 words[4] = {0x88888888u, 0x88888888u, 0x88888888u, 0x88888888u};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int g = 2 * kt + h;
        if (n < N && g < G) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                u32 w = 0;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int k = g * 32 + 16 * kb + 8 * d + (p >> 1) + 4 * (p & 1);
                    const u32 b = Wq[(int64_t)(k >> 1) * N + n];         // canonical: byte [k / 2, n], low nibble = even k
                    w |= ((k & 1) ? (b >> 4) : (b & 0xFu)) << (4 * p);
                }
                words[2 * h + d] = w;
            }
        }
        if (kb == 0) {
            T sc = (T)0.f;
            if (n < N && g < G) sc = S[(int64_t)g * N + n];
            Sa[(step * 32 + j) * 2 + h] = sc;
        }
    }
    Wa[idx] = u32x4{words[0], words[1], words[2], words[3]};
}

template <typename T>
static int launch_w4a8_pack(const uint8_t* Wq, const void* S, void* out, int64_t N, int64_t K, hipStream_t st) {
    const W4A8Layout L = w4a8_layout(N, K, sizeof(T));
    const int64_t total = L.ctiles * L.ksteps * 64;
    w4a8_pack_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(Wq, (const T*)S, (u32x4*)out, (T*)((char*)out + L.off_s),
                                                                         (int)N, (int)(K / 32), (int)L.ksteps, total);
    return finish_launch();
}
int w4a8_pack(int dtype, const uint8_t* Wq, const void* S, void* out, int64_t N, int64_t K, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F32: return launch_w4a8_pack<float>(Wq, S, out, N, K, st);
    case QL_DTYPE_F16: return launch_w4a8_pack<f16>(Wq, S, out, N, K, st);
    case QL_DTYPE_BF16: return launch_w4a8_pack<__bf16>(Wq, S, out, N, K, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

// =============================================================================================
// GEMM.  One loop iteration of a K-parity group covers 128 bytes of K = 4 groups = 2 tile-major units.
// =============================================================================================
// NG: 4-wave groups per block (2: one 8-wave block per CU; 1: two independent 4-wave blocks).  The two groups of an 8-wave
// block are K-parity groups as in w8a8_tiled_kernel (same 128 columns, alternate K chunks, fp32 partial sums exchanged at
// the end) or, with COLG, COLUMN groups: the block covers 256 columns, both groups walk the same K chunks and share ONE
// A tile in LDS - half the A bytes per flop from L2 / the fabric, no exchange.
template <typename T, int MT, int DEPTH, int NG, bool COLG = false>
__global__ __launch_bounds__(NG * 256, NG == 1 ? 2 : 1) void w4a8_kernel(const int8_t* __restrict__ Aq, const u32x4* __restrict__ Wa,
                                                   const T* __restrict__ Sa, int M, int N, int K, int nbx, int super_rows,
                                                   const float* __restrict__ a_scale, const T* __restrict__ bias,
                                                   T* __restrict__ C, int64_t ldc) {
    constexpr int BM = 32 * MT;
    constexpr int BK = 128;
    constexpr int CPR = 8;                             // 16-byte chunks per tile row
    constexpr int KP = COLG ? 1 : NG;                  // K-parity groups
    constexpr int BN = COLG ? 128 * NG : 128;          // columns per block
    constexpr int STG = COLG ? NG * 256 : 256;         // threads staging one A tile
    constexpr int NCH = BM * CPR;
    constexpr int ACH = (NCH + STG - 1) / STG;
    constexpr bool kAllStage = NCH % STG == 0;
    constexpr int BUF = BM * BK;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 groups][3 buffers][BUF]; reused by the epilogue

    const int tid = threadIdx.x, lane = tid & 63, tg = COLG ? tid : tid & 255;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = NG == 1 ? 0 : wave >> 2, wv = wave & 3;
    const int j = lane & 31, kb = lane >> 5;
    const TileXY tile = super_rows ? xcd_tile_super(blockIdx.x, gridDim.x, nbx, super_rows) : xcd_tile(blockIdx.x, gridDim.x, nbx);
    const int m0 = tile.y * BM, n0 = tile.x * BN;
    const int G = K >> 5;
    const int ksteps = (G + 1) >> 1;                   // 64-deep units per column tile
    const int nchunks = (K + BK - 1) / BK;
    const int niter = KP == 1 ? nchunks : (nchunks + 1) >> 1;
    const int kgrp = COLG ? 0 : grp;                   // K-parity index of this group
    const int ctiles = (N + 31) >> 5;
    const int ct_raw = tile.x * (BN / 32) + (COLG ? grp * 4 : 0) + wv;
    const int ct = ct_raw < ctiles ? ct_raw : ctiles - 1;
    const u32x4* wbase = Wa + (int64_t)ct * ksteps * 64 + lane;
    const T* sbase = Sa + ((int64_t)ct * ksteps * 32 + j) * 2;
    char* lds_a = smem + kgrp * (3 * BUF);

    const int8_t* a_src[ACH];
    int a_dst[ACH];
#pragma unroll
    for (int u = 0; u < ACH; ++u) {
        const int q = tg + u * STG, r = (q >> 3) % BM, c = q & 7;
        a_src[u] = Aq + (int64_t)((m0 + r < M) ? (m0 + r) : (M - 1)) * K + c * 16;
        a_dst[u] = (r * 8 + (c ^ ((r >> 1) & 7))) * 16;
    }
    const int c_mine = tg & 7;
    const int klast = K - 16;
    // fragment read offsets: group q of the chunk (0..3), lane half kb: chunk 2 q + kb of row mt * 32 + j
    int a_rd[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a_rd[q] = (j * 8 + ((2 * q + kb) ^ ((j >> 1) & 7))) * 16;

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    // register stages: weights DEPTH iterations ahead, A chunks DA + 2 iterations ahead (two of them sit in LDS)
    constexpr int DA = MT == 4 ? QL_W4A8_DA4 : DEPTH;
    struct StageA {
        i32x4 a[ACH];                                  // A chunk of (slot's iteration + DA + 2)
    };
    struct Stage {
        u32x4 w[2];                                    // two units = four groups of this lane's column
        u32 s[2];                                      // their scales (two per unit)
    };
    Stage st[DEPTH];
    StageA sa[DA];
    auto chunk_of = [&](int i) { return KP * (i < niter ? i : niter - 1) + kgrp; };
    auto load_w = [&](int i, Stage& sg) {
        int t = chunk_of(i);
        t = t < nchunks ? t : nchunks - 1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int ua = 2 * t + u;
            ua = ua < ksteps ? ua : ksteps - 1;
            sg.w[u] = wbase[(int64_t)ua * 64];
            if constexpr (sizeof(T) == 2) sg.s[u] = *reinterpret_cast<const u32*>(sbase + (int64_t)ua * 64);
        }
        if constexpr (sizeof(T) == 4) {                // fp32 scales: handled by the (rare) fp32 instantiation below
            sg.s[0] = sg.s[1] = 0;
        }
    };
    auto load_a = [&](int i, i32x4 (&dst)[ACH]) {
        int t = chunk_of(i);
        t = t < nchunks ? t : nchunks - 1;
        const int off = t * BK + c_mine * 16 <= klast ? t * BK : klast - c_mine * 16;
#pragma unroll
        for (int u = 0; u < ACH; ++u) dst[u] = *reinterpret_cast<const i32x4*>(a_src[u] + off);
    };
    auto store_a = [&](int buf, const i32x4 (&src)[ACH]) {
#pragma unroll
        for (int u = 0; u < ACH; ++u)
            if (kAllStage || tg + u * STG < NCH) *reinterpret_cast<i32x4*>(lds_a + buf * BUF + a_dst[u]) = src[u];
    };
    auto read_a = [&](int buf, int q, i32x4 (&fr)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fr[mt] = *reinterpret_cast<const i32x4*>(lds_a + buf * BUF + mt * 32 * BK + a_rd[q]);
    };
    // nibbles -> int8 operand (bytes 16 (n - 8)); scale (activation dtype) -> fp32 / 16
    u32 k_hi, k_sign;
    asm volatile("s_mov_b32 %0, 0xF0F0F0F0" : "=s"(k_hi));
    asm volatile("s_mov_b32 %0, 0x80808080" : "=s"(k_sign));
    auto unpack = [&](u32 w0, u32 w1) {
        return i32x4{(int)(((w0 << 4) & k_hi) ^ k_sign), (int)((w0 & k_hi) ^ k_sign), (int)(((w1 << 4) & k_hi) ^ k_sign),
                     (int)((w1 & k_hi) ^ k_sign)};
    };
    auto scale_of = [&](u32 packed, int h) {
        float lo, hi;
        unpack2<T>(packed, lo, hi);
        return (h ? hi : lo) * 0.0625f;
    };
    // One group = MT MFMAs from a zero accumulator; each exact int32 result is folded into the fp32 accumulator with the
    // group's scale (16 v_cvt_f32_i32 + 16 v_fma_f32).  Software pipeline of depth one: the fold of MFMA e - 1 is issued
    // behind MFMA e, so the VALU work runs in the shadow of the matrix pipe; the sched_barrier pins that pairing - left
    // alone, hipcc hoists all 4 MT MFMAs of a chunk ahead of the folds (256 result registers: 1 108 spills at MT = 4).
#if QL_W4A8_PKFOLD
    // The MFMAs start from the integer 0x4B400000 instead of 0: the bits of the fp32 value 1.5 * 2^23, whose ulp is 1, so the
    // int32 result READ AS A FLOAT is exactly 12582912 + sum (|sum| <= 32 * 127 * 128 < 2^22).  One packed subtraction
    // gives float(sum) exactly - v_cvt_f32_i32 has no packed form, v_pk_add_f32 / v_pk_fma_f32 handle two accumulators
    // per instruction: 8 + 8 VALU instructions per MFMA instead of 16 + 16, same bits.
    // The accumulators live as 8 register PAIRS per row tile (not as one 16-wide vector: insert / extract on that made
    // hipcc rotate the whole accumulator through fresh registers at every fold); the 8 subtractions are issued ahead of
    // the 8 fmas (a dependent v_pk_add -> v_pk_fma pair costs an s_nop).
    f32x2 accp[MT][8];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < 8; ++p) accp[mt][p] = f32x2{0.f, 0.f};
    auto fold = [&](const i32x16& r, float sc, int mt) {
        const f32x2 s2 = {sc, sc}, m2 = {-12582912.0f, -12582912.0f};
        f32x2 f[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) f[p] = f32x2{i32_as_f32(r[2 * p]), i32_as_f32(r[2 * p + 1])} + m2;
#pragma unroll
        for (int p = 0; p < 8; ++p) accp[mt][p] = __builtin_elementwise_fma(f[p], s2, accp[mt][p]);
    };
    i32x16 magic;
#pragma unroll
    for (int i = 0; i < 16; ++i) magic[i] = 0x4B400000;
    auto zero16 = [&] { return magic; };
#else
    auto fold = [&](const i32x16& r, float sc, int mt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = __builtin_fmaf((float)r[i], sc, acc[mt][i]);
    };
    auto zero16 = [] {
        i32x16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0;
        return z;
    };
#endif
    i32x4 fa[2][MT];
    // full chunk: groups 0..3; fragments one group ahead, across the iteration boundary (three LDS buffers)
    auto mma_chunk_full = [&](int buf, int nbuf, const Stage& sg) {
        i32x16 pend = zero16();
        float sc_prev = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#if !QL_W4A8_ROLL
            if (q + 1 < 4) read_a(buf, q + 1, fa[(q + 1) & 1]);
            else read_a(nbuf, 0, fa[0]);
#endif
            const u32x4& w = sg.w[q >> 1];
            const i32x4 b = unpack(w[2 * (q & 1)], w[2 * (q & 1) + 1]);
            const float sc = scale_of(sg.s[q >> 1], q & 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#if QL_W4A8_ROLL
                // one fragment buffer refilled in place: (next group, row tile mt) is requested right behind the MFMA that
                // consumed fa[0][mt] and has MT MFMAs + folds to arrive
                const i32x16 r = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[0][mt], b, zero16(), 0, 0, 0);
                fa[0][mt] = *reinterpret_cast<const i32x4*>(lds_a + (q + 1 < 4 ? buf : nbuf) * BUF + mt * 32 * BK + a_rd[(q + 1) & 3]);
#else
                const i32x16 r = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[q & 1][mt], b, zero16(), 0, 0, 0);
#endif
                if (q > 0 || mt > 0) fold(pend, mt > 0 ? sc : sc_prev, (mt + MT - 1) % MT);
                pend = r;
                __builtin_amdgcn_sched_barrier(0);
            }
            sc_prev = sc;
        }
        fold(pend, sc_prev, MT - 1);
    };
    auto mma_chunk_partial = [&](int buf, int nbuf, const Stage& sg, int groups) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < groups) {                                              // wave-uniform
                i32x4 f[MT];
                read_a(buf, q, f);
                const u32x4& w = sg.w[q >> 1];
                const i32x4 b = unpack(w[2 * (q & 1)], w[2 * (q & 1) + 1]);
                const float sc = scale_of(sg.s[q >> 1], q & 1);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const i32x16 r = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[mt], b, zero16(), 0, 0, 0);
                    fold(r, sc, mt);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        read_a(nbuf, 0, fa[0]);
    };

    {
        i32x4 a0[ACH], a1[ACH];
        load_a(0, a0);
        load_w(0, st[0]);
        load_a(1, a1);
#pragma unroll
        for (int d = 1; d < DEPTH; ++d) load_w(d, st[d]);
#pragma unroll
        for (int d = 0; d < DA; ++d) load_a(d + 2, sa[d].a);
        store_a(0, a0);
        store_a(1, a1);
    }
    __syncthreads();
    read_a(0, 0, fa[0]);

    constexpr int U = lcm3(3, DEPTH, DA);
    constexpr int AHEAD = DA + 2 > DEPTH ? DA + 2 : DEPTH;
    int it = 0;
    for (; it + U + AHEAD <= niter; it += U) {
#pragma unroll
        for (int d = 0; d < U; ++d) {
            const int slot = d % DEPTH, aslot = d % DA, buf = d % 3, nbuf = (d + 1) % 3, wbuf = (d + 2) % 3;
            store_a(wbuf, sa[aslot].a);
            mma_chunk_full(buf, nbuf, st[slot]);
            load_w(it + d + DEPTH, st[slot]);
            load_a(it + d + DA + 2, sa[aslot].a);
            __syncthreads();
        }
    }
#pragma unroll
    for (int d = 0; d < U + AHEAD - 1; ++d) {
        const int i = it + d;
        if (i < niter) {
            const int slot = d % DEPTH, aslot = d % DA, buf = d % 3, nbuf = (d + 1) % 3, wbuf = (d + 2) % 3;
            if (i + 2 < niter) store_a(wbuf, sa[aslot].a);
            const int t = KP * i + kgrp;
            int groups = G - 4 * t;
            groups = t >= nchunks ? 0 : groups;
            if (groups >= 4) mma_chunk_full(buf, nbuf, st[slot]);
            else mma_chunk_partial(buf, nbuf, st[slot], groups);
            if (i + DEPTH < niter) load_w(i + DEPTH, st[slot]);
            if (i + DA + 2 < niter) load_a(i + DA + 2, sa[aslot].a);
            __syncthreads();
        }
    }

#if QL_W4A8_PKFOLD
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            acc[mt][2 * p] = accp[mt][p][0];
            acc[mt][2 * p + 1] = accp[mt][p][1];
        }
#endif
    // epilogue operands requested before the exchange (their round trip overlaps it)
    const int nw = n0 + (COLG ? grp * 128 : 0) + wv * 32;     // first column of this wave
    const int n = nw + j;
    constexpr int OWN = KP == 1 ? MT : (MT == 1 ? 1 : MT / 2);
    auto owner_of = [](int mt) { return (KP == 1 || MT == 1) ? 0 : (mt & 1); };
    auto owns = [&](int mt) { return KP == 1 || owner_of(mt) == grp; };
    auto own_slot = [](int mt) { return KP == 1 ? mt : (mt >> 1); };
    float asc[OWN][16];
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
        const int mt = KP == 1 ? o : (MT == 1 ? 0 : 2 * o + grp);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
            asc[o][i] = a_scale[m < M ? m : M - 1];
        }
    }
    // combine the two K-parity groups (fp32 partial sums): row tile mt is finished by group mt & 1 (MT == 1: group 0)
    if constexpr (KP == 2) {
        constexpr int SLOTS = (MT + 1) / 2;
        f32x4* xch = reinterpret_cast<f32x4*>(smem);
    #pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int owner = owner_of(mt);
            if (owner != grp) {
                const int slot = mt >> 1;
    #pragma unroll
                for (int q = 0; q < 4; ++q)
                    xch[(((owner * 4 + wv) * SLOTS + slot) * 4 + q) * 64 + lane] =
                        f32x4{acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]};
            }
        }
        __syncthreads();
    #pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int owner = owner_of(mt);
            if (owner == grp) {
                const int slot = mt >> 1;
    #pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 o = xch[(((grp * 4 + wv) * SLOTS + slot) * 4 + q) * 64 + lane];
    #pragma unroll
                    for (int e = 0; e < 4; ++e) acc[mt][4 * q + e] += o[e];
                }
            }
        }
        __syncthreads();
    }

    const bool wide = (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    T* lds_wave = reinterpret_cast<T*>(smem) + (grp * 4 + wv) * 1024;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (!owns(mt)) continue;
        const int o = own_slot(mt);
        if constexpr (sizeof(T) == 2) {
            if (wide) {
                store_tile_32x32<T>(lds_wave, C, ldc, m0 + mt * 32, nw, M, N, bias, lane,
                                    [&](int i) { return acc[mt][i] * asc[o][i]; });
                continue;
            }
        }
        if (n < N) {
            const T* bn = bias ? bias + n : nullptr;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
                if (m < M) store_out<T>(C + (int64_t)m * ldc + n, acc[mt][i] * asc[o][i], bn);
            }
        }
    }
}

template <typename T, int MT, int DEPTH, int NG, bool COLG = false>
static int launch_w4a8_mt(const int8_t* Aq, const float* a_scale, const void* packed, const void* bias, void* C, int64_t M,
                          int64_t N, int64_t K, int64_t ldc, hipStream_t st) {
    constexpr int BM = 32 * MT;
    const W4A8Layout L = w4a8_layout(N, K, sizeof(T));
    constexpr int BN = COLG ? 128 * NG : 128;
    const int nbx = (int)((N + BN - 1) / BN), nby = (int)((M + BM - 1) / BM);
    constexpr int kTiles = (COLG ? 1 : NG) * 3 * BM * 128;
    constexpr int kLds = kTiles < NG * 8192 ? NG * 8192 : kTiles;
    static bool attr_set = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&w4a8_kernel<T, MT, DEPTH, NG, COLG>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kLds) == hipSuccess;
    }();
    (void)attr_set;
    const bool no_super = QL_TUNE("QLINEAR_GEMM_SUPER", 1) == 0;
    const int sy = NG == 1 ? 8 : 4;
    const bool super = !no_super && nbx % 8 == 0 && nby % sy == 0 && nby >= 2 * sy;    // ql_common.h: xcd_tile_super
    w4a8_kernel<T, MT, DEPTH, NG, COLG><<<(unsigned)(nbx * nby), NG * 256, kLds, st>>>(
        Aq, (const u32x4*)packed, (const T*)((const char*)packed + L.off_s), (int)M, (int)N, (int)K,
        super ? nbx : xcd_order(nbx, nby, (double)M * K, (double)N * K / 2), super ? sy : 0, a_scale, (const T*)bias, (T*)C, ldc);
    return finish_launch();
}

template <typename T>
static int launch_w4a8(const int8_t* Aq, const float* a_scale, const void* packed, const void* bias, void* C, int64_t M,
                       int64_t N, int64_t K, int64_t ldc, hipStream_t st) {
    const int forced_mt = QL_TUNE("QLINEAR_W4A8_MT", 0), forced_ng = QL_TUNE("QLINEAR_W4A8_NG", 0);
    const int64_t nb = (N + 127) / 128;
    int mt = 1;
    for (int t = 4; t > 1; t >>= 1)
        if (M > 16 * t && nb * ((M + 32 * t - 1) / (32 * t)) >= 256) { mt = t; break; }
    if (mt == 1 && M > 32) mt = 2;
    if (forced_mt == 1 || forced_mt == 2 || forced_mt == 4) mt = forced_mt;
    int ng = (mt == 4 && nb * ((M + 127) / 128) >= 512) ? 1 : 2;
    if (forced_ng == 1 || forced_ng == 2) ng = forced_ng;
    const int colg = QL_TUNE("QLINEAR_W4A8_COLG", 0);
    if (mt == 4 && colg == 1) return launch_w4a8_mt<T, 4, QL_W4A8_DEPTH4, 2, true>(Aq, a_scale, packed, bias, C, M, N, K, ldc, st);
    if (mt == 4 && ng == 1) return launch_w4a8_mt<T, 4, QL_W4A8_DEPTH4, 1>(Aq, a_scale, packed, bias, C, M, N, K, ldc, st);
    if (mt == 4) return launch_w4a8_mt<T, 4, QL_W4A8_DEPTH4, 2>(Aq, a_scale, packed, bias, C, M, N, K, ldc, st);
    if (mt == 2) return launch_w4a8_mt<T, 2, 3, 2>(Aq, a_scale, packed, bias, C, M, N, K, ldc, st);
    return launch_w4a8_mt<T, 1, 3, 2>(Aq, a_scale, packed, bias, C, M, N, K, ldc, st);
}

int w4a8_gemm(int dtype, const int8_t* Aq, const float* a_scale, const void* packed, const void* bias, void* C, int64_t M,
              int64_t N, int64_t K, int64_t ldc, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_w4a8<f16>(Aq, a_scale, packed, bias, C, M, N, K, ldc, st);
    case QL_DTYPE_BF16: return launch_w4a8<__bf16>(Aq, a_scale, packed, bias, C, M, N, K, ldc, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

}  // namespace ql
