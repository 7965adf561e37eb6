// Transposed int4g32 product on the CANONICAL layout (backward of the quantized matmul), gfx950.
//
//   C[M, Nout] = A[M, Kc] . dequant(W)^T      A = grad_out, Kc = the forward's N, Nout = the forward's K
//   W (Nout/2, Kc) uint8: byte [p][k] = rows 2p (low nibble) and 2p+1 (high nibble), nibble = q + 8
//   S (Nout/32, Kc): dequant(W)[n][k] = (nibble - 8) * S[n/32][k], rounded to the activation dtype
//
// This is the reference's `dynamic_quant_matmul_transposed_s4` (chatglm_q/int4/triton_ops.py:142-264, called from
// DynamicQuantizeMatMul.backward, chatglm_q/int4/qlinear.py:53-64) with the same arithmetic as its kernel
// (triton_ops.py:191-195): every weight dequantised and ROUNDED to the activation dtype, exact products, fp32
// accumulation, one output rounding.  No derived layout is needed: the canonical bytes are contiguous along the
// contraction index here, so lane (j, kb) of a wave takes 8 consecutive bytes of byte-row j per MFMA sub-step and
// gets TWO B fragments out of them - the low nibbles are output column 2j, the high nibbles column 2j+1.
// The scales vary along the contraction (8 per fragment, shared by the 32 output columns of a group): each wave
// stages the 2 groups x 64 k of its 64 columns through LDS per K step (256 bytes) and reads them back as
// broadcast ds_read_b128.  A tile, register ring and fragment timing as in w4_gemm.hip / w8_gemm.hip.
// fp16: nibble -> half by exponent splice on the byte PAIRS (b0,b2),(b1,b3) of a word, so the A tile and the
// scale tile are staged with every 4 halves regrouped the same way (a consistent K permutation).
#include "launch.h"
#include "ql_common.h"

namespace ql {

typedef _Float16 t4_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 t4_bf16x8 __attribute__((ext_vector_type(8)));
typedef float t4_f32x16 __attribute__((ext_vector_type(16)));

// regroup the 8 halves of a 16-byte chunk as (a0,a2),(a1,a3),(a4,a6),(a5,a7)
__device__ __forceinline__ u32x4 t4_pair_even_odd(u32x4 x) {
    u32x4 y;
    y[0] = (x[0] & 0xFFFFu) | (x[1] << 16);
    y[1] = (x[0] >> 16) | (x[1] & 0xFFFF0000u);
    y[2] = (x[2] & 0xFFFFu) | (x[3] << 16);
    y[3] = (x[2] >> 16) | (x[3] & 0xFFFF0000u);
    return y;
}

template <typename T> struct TMma;
template <> struct TMma<f16> {
    typedef t4_f16x8 frag;
    static constexpr bool kPaired = true;
    static constexpr u32 kMagic = 0x64006400u;
    static __device__ __forceinline__ t4_f32x16 mma(frag a, frag b, t4_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    // 8 bytes (w0, w1) + their 8 scales (paired order) -> fragment of the even column (low nibbles) and of the
    // odd column (high nibbles): 0x6400 | n = 1024 + n, 0x6400 | (n << 4) = 1024 + 16 n
    static __device__ __forceinline__ void dequant(u32 w0, u32 w1, u32x4 sc, u32 m_lo, u32 m_hi, u32 magic, frag& even, frag& odd) {
        const h2 k1032 = {(f16)1032.0f, (f16)1032.0f};
        const h2 kInv16 = {(f16)0.0625f, (f16)0.0625f};
        const h2 kM72 = {(f16)-72.0f, (f16)-72.0f};
        u32x4 e, o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32 w = h ? w1 : w0, w8 = w >> 8;
            const h2 s02 = as_h2(sc[2 * h]), s13 = as_h2(sc[2 * h + 1]);
            e[2 * h] = as_u32((as_h2((w & m_lo) | magic) - k1032) * s02);               // exact n - 8, ONE rounding in * s
            e[2 * h + 1] = as_u32((as_h2((w8 & m_lo) | magic) - k1032) * s13);
            o[2 * h] = as_u32((as_h2((w & m_hi) | magic) * kInv16 + kM72) * s02);
            o[2 * h + 1] = as_u32((as_h2((w8 & m_hi) | magic) * kInv16 + kM72) * s13);
        }
        even = __builtin_bit_cast(frag, e);
        odd = __builtin_bit_cast(frag, o);
    }
};
template <> struct TMma<__bf16> {
    typedef t4_bf16x8 frag;
    static constexpr bool kPaired = false;     // natural K order
    static constexpr u32 kMagic = 0;
    static __device__ __forceinline__ t4_f32x16 mma(frag a, frag b, t4_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void dequant(u32 w0, u32 w1, u32x4 sc, u32, u32, u32, frag& even, frag& odd) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        u32x4 e, o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                       // byte pair (2i, 2i+1) of the octet, scales (s_2i, s_2i+1)
            const u32 w = (i < 2 ? w0 : w1) >> (16 * (i & 1));
            const float s0 = u32_as_f32(sc[i] << 16), s1 = u32_as_f32(sc[i] & 0xFFFF0000u);
            const float l0 = ((float)(w & 0xFu) - 8.0f) * s0, l1 = ((float)((w >> 8) & 0xFu) - 8.0f) * s1;
            const float h0 = ((float)((w >> 4) & 0xFu) - 8.0f) * s0, h1 = ((float)((w >> 12) & 0xFu) - 8.0f) * s1;
            const bf2 pe = {(__bf16)l0, (__bf16)l1}, po = {(__bf16)h0, (__bf16)h1};   // one rounding each
            e[i] = __builtin_bit_cast(u32, pe);
            o[i] = __builtin_bit_cast(u32, po);
        }
        even = __builtin_bit_cast(frag, e);
        odd = __builtin_bit_cast(frag, o);
    }
};

template <typename T, int MT, int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void w4_tgemm_kernel(const T* __restrict__ A, const uint8_t* __restrict__ Wq,
                                                           const T* __restrict__ S, T* __restrict__ C, int M, int R,
                                                           int Kc, int G, int64_t lda, int64_t ldw, int64_t lds,
                                                           int64_t ldc, int nbx) {
    constexpr int BM = 32 * MT;
    constexpr int NTHR = NW * 64;
    constexpr int CH = (BM * 8 + NTHR - 1) / NTHR;   // 16-byte A chunks staged per thread per K step
    constexpr bool kAllStage = (BM * 8) % NTHR == 0;
    typedef TMma<T> MM;
    __shared__ __attribute__((aligned(16))) char smem[2][BM * 128];
    __shared__ __attribute__((aligned(16))) char ssc[2][NW * 256];   // per wave: 2 groups x 8 chunks x 16 B

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kb = lane >> 5;
    const TileXY tile = xcd_tile(blockIdx.x, gridDim.x, nbx);
    const int m0 = tile.y * BM;
    const int p_base = tile.x * (NW * 32) + wave * 32;        // first byte row of the wave (a multiple of 32)
    const int p_raw = p_base + j;
    const int p = p_raw < R ? p_raw : R - 1;                  // clamped: loads stay in bounds, stores are masked
    const int ksteps = (Kc + 63) >> 6;

    u32 m_lo, m_hi, k_magic;
    asm volatile("s_mov_b32 %0, 0x000F000F" : "=s"(m_lo));
    asm volatile("s_mov_b32 %0, 0x00F000F0" : "=s"(m_hi));
    asm volatile("v_mov_b32 %0, %1" : "=v"(k_magic) : "i"(MM::kMagic));

    const uint8_t* wrow = Wq + (int64_t)p * ldw + kb * 32;
    // scale staging: lane l & 15 -> group (l >> 3) & 1 of the wave's two, 16-byte chunk l & 7 of the step
    const int sg_l = (lane >> 3) & 1, sc_l = lane & 7;
    const int g_want = p_base / 16 + sg_l, g_mine = g_want < G ? g_want : G - 1;
    const T* srow = S + (int64_t)g_mine * lds + sc_l * 8;
    const int s_dst = wave * 256 + (sg_l * 8 + sc_l) * 16;

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
    const int kmax_a = Kc - 8;                 // last in-bounds 8-half chunk start
    const int kmax_w = Kc - 16;                // last in-bounds 16-byte weight chunk start

    t4_f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.f;

    struct Stage {
        u32x4 a[CH];
        u32x4 w[2];
        u32x4 s;
    };
    Stage st[DEPTH];
    auto load_stage = [&](int kt, Stage& sg) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int c = (tid + u * NTHR) & 7;
            const int k = kt * 64 + c * 8;
            sg.a[u] = *reinterpret_cast<const u32x4*>(a_src[u] + (k <= kmax_a ? kt * 64 : kmax_a - c * 8));   // K tail: clamped
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = kt * 64 + kb * 32 + h * 16;
            sg.w[h] = *reinterpret_cast<const u32x4*>(wrow + (k <= kmax_w ? kt * 64 + h * 16 : kmax_w - kb * 32));
        }
        const int ks = kt * 64 + sc_l * 8;
        sg.s = *reinterpret_cast<const u32x4*>(srow + (ks <= kmax_a ? kt * 64 : kmax_a - sc_l * 8));
    };
    auto store_tiles = [&](int buf, const Stage& sg) {
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (kAllStage || tid + u * NTHR < BM * 8)
                *reinterpret_cast<u32x4*>(smem[buf] + a_dst[u]) = MM::kPaired ? t4_pair_even_odd(sg.a[u]) : sg.a[u];
        if (lane < 16) *reinterpret_cast<u32x4*>(ssc[buf] + s_dst) = MM::kPaired ? t4_pair_even_odd(sg.s) : sg.s;
    };
    auto mma_step = [&](int buf, int kt, const u32x4 (&w_in)[2]) {
        // bytes of a K tail (k >= Kc) become 0x88: both nibbles dequantise to exactly 0
        u32x4 w[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
            w[h] = (kt * 64 + kb * 32 + h * 16 <= kmax_w) ? w_in[h] : u32x4{0x88888888u, 0x88888888u, 0x88888888u, 0x88888888u};
        auto read_ops = [&](int sub, u32x4 (&fr)[MT], u32x4& fs) {
            const int c = kb * 4 + sub;
            fs = *reinterpret_cast<const u32x4*>(ssc[buf] + wave * 256 + ((j >> 4) * 8 + c) * 16);   // broadcast read
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r = mt * 32 + j;
                fr[mt] = *reinterpret_cast<const u32x4*>(smem[buf] + (r * 8 + (c ^ ((r >> 1) & 7))) * 16);
            }
        };
        u32x4 fa[2][MT], fs[2];
        typename MM::frag fe[2], fo[2];
        read_ops(0, fa[0], fs[0]);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            // B fragments of this sub-step (their scales were read one sub-step ahead), reads of the next one
            MM::dequant(w[sub >> 1][2 * (sub & 1)], w[sub >> 1][2 * (sub & 1) + 1], fs[sub & 1], m_lo, m_hi, k_magic,
                        fe[sub & 1], fo[sub & 1]);
            if (sub < 3) read_ops(sub + 1, fa[(sub + 1) & 1], fs[(sub + 1) & 1]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt][0] = MM::mma(__builtin_bit_cast(typename MM::frag, fa[sub & 1][mt]), fe[sub & 1], acc[mt][0]);
                acc[mt][1] = MM::mma(__builtin_bit_cast(typename MM::frag, fa[sub & 1][mt]), fo[sub & 1], acc[mt][1]);
            }
        }
    };

#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_stage(d < ksteps ? d : ksteps - 1, st[d]);
    store_tiles(0, st[0]);
    __syncthreads();

    int kt = 0;
    for (; kt + DEPTH < ksteps; kt += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int buf = (kt + d) & 1;
            const u32x4 w_cur[2] = {st[d].w[0], st[d].w[1]};
            load_stage(kt + d + DEPTH < ksteps ? kt + d + DEPTH : ksteps - 1, st[d]);
            mma_step(buf, kt + d, w_cur);
            store_tiles(buf ^ 1, st[(d + 1) % DEPTH]);
            __syncthreads();
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        if (kt + d < ksteps) {
            const int buf = (kt + d) & 1;
            mma_step(buf, kt + d, st[d].w);
            if (kt + d + 1 < ksteps) store_tiles(buf ^ 1, st[(d + 1) % DEPTH]);
            __syncthreads();
        }
    }

    // C/D map of the 32x32 MFMA: column = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5); the lane's two
    // accumulators are the adjacent output columns 2p and 2p+1: one 4-byte store
    if (p_raw < R) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
                if (m >= M) continue;
                T pair[2];
                Act<T>::store(&pair[0], acc[mt][0][i]);
                Act<T>::store(&pair[1], acc[mt][1][i]);
                *reinterpret_cast<u32*>(C + (int64_t)m * ldc + 2 * p_raw) = *reinterpret_cast<const u32*>(pair);
            }
    }
}

template <typename T, int MT>
static int launch_tgemm(const void* A, const uint8_t* Wq, const void* S, void* C, int M, int R, int Kc, int G, int64_t lda,
                        int64_t ldw, int64_t lds, int64_t ldc, hipStream_t st) {
    constexpr int NW = 4;
    const int nbx = (R + NW * 32 - 1) / (NW * 32), nby = (M + 32 * MT - 1) / (32 * MT);
    w4_tgemm_kernel<T, MT, NW, 3><<<(unsigned)(nbx * nby), NW * 64, 0, st>>>(
        (const T*)A, Wq, (const T*)S, (T*)C, M, R, Kc, G, lda, ldw, lds, ldc,
        xcd_order(nbx, nby, (double)M * Kc * 2, (double)R * Kc));
    return finish_launch();
}

template <typename T>
static int launch_tgemm_any(const void* A, const uint8_t* Wq, const void* S, void* C, int64_t M, int64_t R, int64_t Kc,
                            int64_t G, int64_t lda, int64_t ldw, int64_t lds, int64_t ldc, hipStream_t st) {
    const int64_t nb = (R + 127) / 128;
    if (M > 32 && nb * ((M + 63) / 64) >= 256)
        return launch_tgemm<T, 2>(A, Wq, S, C, (int)M, (int)R, (int)Kc, (int)G, lda, ldw, lds, ldc, st);
    return launch_tgemm<T, 1>(A, Wq, S, C, (int)M, (int)R, (int)Kc, (int)G, lda, ldw, lds, ldc, st);
}

int w4_tgemm(int dtype, const void* A, const uint8_t* Wq, const void* S, void* C, int64_t M, int64_t Nout, int64_t Kc,
             int64_t lda, int64_t ldw, int64_t lds, int64_t ldc, hipStream_t st) {
    const int64_t R = Nout / 2, G = Nout / 32;
    switch (dtype) {
    case QL_DTYPE_F16: return launch_tgemm_any<f16>(A, Wq, S, C, M, R, Kc, G, lda, ldw, lds, ldc, st);
    case QL_DTYPE_BF16: return launch_tgemm_any<__bf16>(A, Wq, S, C, M, R, Kc, G, lda, ldw, lds, ldc, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

}  // namespace ql
