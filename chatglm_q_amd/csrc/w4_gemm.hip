// int4g32 GEMM for many activation rows (prefill / batched decode) on the derived layout, gfx950.
//
//   C[M,N] = A[M,K] . dequant(W)      fp16 / bf16 activations, fp32 accumulation on MFMA
//
// This IS a dense GEMM (the reference itself runs it on tl.dot, chatglm_q/int4/triton_ops.py:75), and it
// keeps the reference's arithmetic exactly: every weight is dequantised to (n - 8) * s ROUNDED to the
// activation dtype (triton_ops.py:72-73) in registers and fed to v_mfma_f32_32x32x16_{f16,bf16}, whose
// products are exact and whose accumulation is fp32.
//
// Decomposition: block = 4 waves side by side in N (128 columns) x BM = 32 MT rows.  A wave owns 32
// columns: lane (j = lane & 31, kb = lane >> 5) loads ONE 16-byte unit per 64-deep K step - the 32
// nibbles of column j, group 2 kt + kb - and turns word s of it into the B fragment of MFMA sub-step s
// (the two half-waves supply the two groups of the step, so an MFMA's K = 16 is octet s of group 2kt
// and octet s of group 2kt+1; the A fragment uses the same assignment, any consistent K permutation
// gives the same sum).  The 13 VALU ops that build a B fragment are reused by MT MFMAs.
// The A tile (BM x 64 halves) is staged through LDS by all 4 waves (global -> registers one step ahead ->
// ds_write_b128 after the math), double buffered, one barrier per K step; 16-byte chunk c of row r lives
// at chunk position 8 r + (c ^ ((r >> 1) & 7)), which makes the 128-byte-pitch ds_read_b128 fragment
// reads bank-conflict free (16-lane service groups, 64-bank rows).
#include "launch.h"
#include "w4_dequant.h"

namespace ql {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T> struct Mma;
template <> struct Mma<f16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    // word (8 nibbles, positions (p, p+4) = k pair (2p, 2p+1)) -> 8 dequantised halves in k order
    static __device__ __forceinline__ frag dequant(u32 w, u32 k_mask_lo, u32 k_mask_hi, u32 k_magic, h2 s2) {
        const h2 k1032 = {(f16)1032.0f, (f16)1032.0f};
        const h2 kInv16 = {(f16)0.0625f, (f16)0.0625f};
        const h2 kM72 = {(f16)-72.0f, (f16)-72.0f};
        const u32 w8 = w >> 8;
        const h2 e0 = (as_h2((w & k_mask_lo) | k_magic) - k1032) * s2;              // exact (n-8), ONE rounding in * s
        const h2 e1 = (as_h2((w & k_mask_hi) | k_magic) * kInv16 + kM72) * s2;
        const h2 e2 = (as_h2((w8 & k_mask_lo) | k_magic) - k1032) * s2;
        const h2 e3 = (as_h2((w8 & k_mask_hi) | k_magic) * kInv16 + kM72) * s2;
        u32x4 r = {as_u32(e0), as_u32(e1), as_u32(e2), as_u32(e3)};
        return __builtin_bit_cast(frag, r);
    }
    static __device__ __forceinline__ h2 scale_pair(const f16* p, bool valid) {
        const f16 s = valid ? *p : (f16)0.f;
        return h2{s, s};
    }
    static constexpr u32 kMagic = 0x64006400u;
};
template <> struct Mma<__bf16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    // bf16: 0x4300 | n = 128 + n; (128 + n) * s - 136 s is exact in fp32, then ONE rounding to bf16
    static __device__ __forceinline__ frag dequant(u32 w, u32 k_mask_lo, u32, u32 k_magic, float s) {
        const float m136s = -136.0f * s;
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 x = ((w >> (4 * i)) & k_mask_lo) | k_magic;           // nibble pair i = k pair (2i, 2i+1)
            const float lo = __builtin_fmaf(u32_as_f32(x << 16), s, m136s);
            const float hi = __builtin_fmaf(u32_as_f32(x & 0xFFFF0000u), s, m136s);
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            const bf2 p = {(__bf16)lo, (__bf16)hi};                         // v_cvt_pk_bf16_f32
            r[i] = __builtin_bit_cast(u32, p);
        }
        return __builtin_bit_cast(frag, r);
    }
    static __device__ __forceinline__ float scale_pair(const __bf16* p, bool valid) { return valid ? (float)*p : 0.f; }
    static constexpr u32 kMagic = 0x43004300u;
};

template <typename T, int MT>
__global__ __launch_bounds__(256) void w4_packed_gemm_kernel(const T* __restrict__ A, const u32x4* __restrict__ Wt,
                                                             const T* __restrict__ Sp, const T* __restrict__ bias,
                                                             T* __restrict__ C, int M, int N, int K, int G,
                                                             int64_t lda, int64_t ldc) {
    constexpr int BM = 32 * MT;
    constexpr int CH = BM * 8 / 256;           // 16-byte A chunks staged per thread per K step
    typedef Mma<T> MM;
    __shared__ __attribute__((aligned(16))) char smem[2][BM * 128];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kb = lane >> 5;
    const int m0 = blockIdx.y * BM;
    const int n_raw = blockIdx.x * 128 + wave * 32 + j;
    const int n = n_raw < N ? n_raw : N - 1;   // clamped column: loads stay in bounds, stores are masked
    const int ksteps = (G + 1) >> 1;           // 64 k (two groups) per step

    u32 k_mask_lo, k_mask_hi, k_magic;
    asm volatile("s_mov_b32 %0, 0x000F000F" : "=s"(k_mask_lo));
    asm volatile("s_mov_b32 %0, 0x00F000F0" : "=s"(k_mask_hi));
    asm volatile("v_mov_b32 %0, %1" : "=v"(k_magic) : "i"(MM::kMagic));

    const u32x4* wcol = Wt + (int64_t)n * G;
    const T* scol = Sp + ((int64_t)(n >> 2) * G) * 4 + (n & 3);

    // A staging: thread -> CH chunks; chunk q: row q / 8, 16-byte column q % 8
    const T* a_src[CH];
    int a_dst[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        const int q = tid + u * 256;
        const int r = q >> 3, c = q & 7;
        const int row = (m0 + r < M) ? (m0 + r) : (M - 1);
        a_src[u] = A + (int64_t)row * lda + c * 8;
        a_dst[u] = (r * 8 + (c ^ ((r >> 1) & 7))) * 16;
    }
    const int kmax = K - 8;                    // last in-bounds 8-half chunk start

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    // Register ring, DEPTH K-steps deep: the global loads (A chunks, this lane's weight unit AND its scale) of
    // step kt + DEPTH are issued when step kt starts, so DEPTH - 1 steps of MFMA work cover their latency.  In
    // the steady-state loop every load is unconditional and real, so hipcc's vmcnt waits are exact (a load
    // under `if (more)` made it drain the queue every step; the per-step scale load sat on the critical path).
    constexpr int DEPTH = 3;
    typedef decltype(MM::scale_pair(scol, true)) scale_t;
    struct Stage {
        u32x4 a[CH];
        u32x4 w;
        T s;
    };
    Stage st[DEPTH];
    auto load_stage = [&](int kt, Stage& sg) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int c = (tid + u * 256) & 7;
            const int k = kt * 64 + c * 8;
            // a K tail (odd group count) reads a clamped chunk; its weights are zeroed at use
            sg.a[u] = *reinterpret_cast<const u32x4*>(a_src[u] + (k <= kmax ? kt * 64 : kmax - c * 8));
        }
        const int g = 2 * kt + kb;
        const int gc = g < G ? g : G - 1;
        sg.w = __builtin_nontemporal_load(wcol + gc);
        sg.s = scol[(int64_t)gc * 4];
    };
    auto store_a = [&](int buf, const Stage& sg) {
#pragma unroll
        for (int u = 0; u < CH; ++u) *reinterpret_cast<u32x4*>(smem[buf] + a_dst[u]) = sg.a[u];
    };
    auto mma_step = [&](int buf, int kt, u32x4 w_cur, T s_raw) {
        const int g = 2 * kt + kb;
        const T s_eff = g < G ? s_raw : (T)0.f;              // the missing half of an odd last step contributes 0
        const scale_t s = MM::scale_pair(&s_eff, true);
        // A fragments are read ONE SUB-STEP AHEAD of the MFMAs that consume them: with the reads issued right in
        // front of each MFMA (what the straightforward loop compiled to) every MFMA waited out a full LDS round
        // trip (~100+ cycles against its own 32) and the matrix pipe idled 70 % of the time
        auto read_a = [&](int sub, u32x4 (&fr)[MT]) {
            const int c = kb * 4 + sub;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r = mt * 32 + j;
                fr[mt] = *reinterpret_cast<const u32x4*>(smem[buf] + (r * 8 + (c ^ ((r >> 1) & 7))) * 16);
            }
        };
        u32x4 fa[2][MT];
        typename MM::frag fb[2];
        read_a(0, fa[0]);
        fb[0] = MM::dequant(w_cur[0], k_mask_lo, k_mask_hi, k_magic, s);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            // software pipeline inside the step: while the MFMAs of sub-step `sub` run, issue the A-fragment reads
            // and build the B fragment of sub-step sub + 1 (VALU slots under the 32-cycle MFMA shadows)
            if (sub < 3) {
                read_a(sub + 1, fa[(sub + 1) & 1]);
                fb[(sub + 1) & 1] = MM::dequant(w_cur[sub + 1], k_mask_lo, k_mask_hi, k_magic, s);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = MM::mma(__builtin_bit_cast(typename MM::frag, fa[sub & 1][mt]), fb[sub & 1], acc[mt]);
            // schedule shape per sub-step: the MT reads first, then each MFMA followed by a slice of the VALU work
            __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);            // DS reads of the next sub-step
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, (16 + MT - 1) / MT, 0);   // its share of the dequant VALU
            }
        }
    };

#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_stage(d < ksteps ? d : ksteps - 1, st[d]);
    store_a(0, st[0]);
    __syncthreads();

    int kt = 0;
    for (; kt + DEPTH < ksteps; kt += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int buf = (kt + d) & 1;
            const u32x4 w_cur = st[d].w;
            const T s_cur = st[d].s;
            // slot d: its A chunks went to LDS last step, its weight unit / scale were just copied out
            load_stage(kt + d + DEPTH < ksteps ? kt + d + DEPTH : ksteps - 1, st[d]);
            mma_step(buf, kt + d, w_cur, s_cur);
            store_a(buf ^ 1, st[(d + 1) % DEPTH]);
            __syncthreads();
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        if (kt + d < ksteps) {
            const int buf = (kt + d) & 1;
            mma_step(buf, kt + d, st[d].w, st[d].s);
            if (kt + d + 1 < ksteps) store_a(buf ^ 1, st[(d + 1) % DEPTH]);
            __syncthreads();
        }
    }

    // C/D map of 32x32 MFMA: column = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    if (n_raw < N) {
        const T* bn = bias ? bias + n_raw : nullptr;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
                if (m < M) store_out<T>(C + (int64_t)m * ldc + n_raw, acc[mt][i], bn);
            }
    }
}

template <typename T, int MT>
static int launch_gemm(const void* A, const void* packed, const void* bias, void* C, int M, int N, int K, int64_t lda,
                       int64_t ldc, hipStream_t st) {
    const int64_t G = K / 32, Npad = (N + 3) & ~(int64_t)3;
    const u32x4* Wt = (const u32x4*)packed;
    const T* Sp = (const T*)((const char*)packed + Npad * G * 16);
    dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 32 * MT - 1) / (32 * MT)));
    w4_packed_gemm_kernel<T, MT><<<grid, 256, 0, st>>>((const T*)A, Wt, Sp, (const T*)bias, (T*)C, M, N, K, (int)G, lda, ldc);
    return finish_launch();
}

template <typename T>
static int launch_gemm_any(const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N,
                           int64_t K, int64_t lda, int64_t ldc, hipStream_t st) {
    if (M > 64) return launch_gemm<T, 4>(A, packed, bias, C, (int)M, (int)N, (int)K, lda, ldc, st);
    if (M > 32) return launch_gemm<T, 2>(A, packed, bias, C, (int)M, (int)N, (int)K, lda, ldc, st);
    return launch_gemm<T, 1>(A, packed, bias, C, (int)M, (int)N, (int)K, lda, ldc, st);
}

int w4_packed_gemm(int dtype, const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N,
                   int64_t K, int64_t lda, int64_t ldc, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_gemm_any<f16>(A, packed, bias, C, M, N, K, lda, ldc, st);
    case QL_DTYPE_BF16: return launch_gemm_any<__bf16>(A, packed, bias, C, M, N, K, lda, ldc, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

}  // namespace ql
