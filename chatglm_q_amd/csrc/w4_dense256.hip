// Many-row int4g32 / int8 weight-only GEMM as TWO launches (round 4): the weights of the call are dequantised ONCE into a 16-bit
// fragment-major image (w4_expand_kernel: (n - 8) * s, int8: b * s, ROUNDED to the activation dtype - the reference's per-weight
// rounding, chatglm_q/int4/triton_ops.py:71-73, chatglm_q/int8/triton_ops.py:62-73), then a dense 16-bit GEMM streams that image
// (dense256_kernel: the int8 x int8 ring kernel's structure, w8a8_gemm256.hip, on v_mfma_f32_16x16x32_{f16,bf16}).
//
// Why: at 8192 rows the fused kernel (w4_gemm256.hip) dequantises every weight in each of its 32 row blocks - 416 VALU wave
// instructions and 32 KB of LDS stores per 64-deep tile and block beside 1 024 pipe cycles of MFMA - and lands at 83 % of the vendor's
// DENSE f16 GEMM, power bound whatever its loop looks like (profiles/r04_gemm_power.txt).  Dequantising once per CALL costs one
// memory-bound pass (4096 x 4096: 9 MB in, 32 MB out) and leaves the GEMM with no VALU and no LDS stores in its loop.
//
// Image layout (what a stage of the GEMM wants in LDS, so its LDS-DMA is linear): for block column bx (256 columns) and stage t (32 k =
// one int4 group) 16 KB = [tile nt (16)][lane 16 q + i (64)][16 B]; slot (nt, q, i) holds the 8 weights k = 32 t + 8 q .. + 7 of column
// 256 bx + 32 (nt >> 1) + 8 (i >> 2) + 4 (nt & 1) + (i & 3) - the column mapping that leaves a lane of the GEMM with EIGHT CONSECUTIVE
// columns of one output row per tile pair (weight fragment = first MFMA operand), i.e. 16-byte row stores straight from registers.
#include "launch.h"
#include "vmq.h"
#include "w4_dequant.h"
#include "w4_mma.h"

namespace ql {

typedef float f32x4d __attribute__((ext_vector_type(4)));
constexpr int kD256Stage = 32768;                  // A 16 KB | B 16 KB
constexpr int kD256Lds = 4 * kD256Stage;

// fp16 int8 dequant of 8 bytes in natural k order (w4_gemm256.hip has the same helper for its fused path)
template <typename T>
__device__ __forceinline__ u32x4 w8_dequant8(u32 w0, u32 w1, float s) {
    if constexpr (Act<T>::code == QL_DTYPE_F16) {
        const h2 k1152 = {(f16)1152.0f, (f16)1152.0f};
        const f16 sh = (f16)s;
        const h2 s2 = {sh, sh};
        const u32 t0 = w0 ^ 0x80808080u, t1 = w1 ^ 0x80808080u, k64 = 0x64646464u;
        const h2 e0 = (as_h2(__builtin_amdgcn_perm(k64, t0, 0x04010400u)) - k1152) * s2;
        const h2 e1 = (as_h2(__builtin_amdgcn_perm(k64, t0, 0x04030402u)) - k1152) * s2;
        const h2 e2 = (as_h2(__builtin_amdgcn_perm(k64, t1, 0x04010400u)) - k1152) * s2;
        const h2 e3 = (as_h2(__builtin_amdgcn_perm(k64, t1, 0x04030402u)) - k1152) * s2;
        return u32x4{as_u32(e0), as_u32(e1), as_u32(e2), as_u32(e3)};
    } else {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 w = i < 2 ? w0 : w1;
            const int sh = 16 * (i & 1);
            const float lo = (float)((int)(w << (24 - sh)) >> 24) * s, hi = (float)((int)(w << (16 - sh)) >> 24) * s;
            const bf2 pr = {(__bf16)lo, (__bf16)hi};
            r[i] = __builtin_bit_cast(u32, pr);
        }
        return r;
    }
}

// One thread = one unit of the tile-major copy: int4 (W8 = false): Wm[ct][kt][lane] (column 32 ct + j, group 2 kt + kb) + its scale;
// int8: [ct][kt][half h][lane 32 kb + j][16 B] - the thread takes both halves of (ct, kt, kb, j): 32 bytes = one 32-deep stage.
template <typename T, bool W8>
__global__ __launch_bounds__(256) void w4_expand_kernel(const u32x4* __restrict__ Wm, const T* __restrict__ Sm, u32x4* __restrict__ img, int N,
                                                        int ksteps, int stages, int64_t total) {
    typedef Mma<T> MM;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (ct * ksteps + kt) * 64 + lane
    if (idx >= total) return;
    const int lane = (int)(idx & 63), j = lane & 31, kb = lane >> 5;
    const int64_t step = idx >> 6;
    const int kt = (int)(step % ksteps), ct = (int)(step / ksteps);
    const int t = 2 * kt + kb;                                             // the stage (32 k) of this unit
    if (t >= stages) return;
    u32x4 f[4];
    if constexpr (W8) {
        const int n = 32 * ct + j;
        const float s = Act<T>::load(Sm + (n < N ? n : N - 1));
        const u32x4 lo = Wm[(step * 2 + 0) * 64 + lane], hi = Wm[(step * 2 + 1) * 64 + lane];
        f[0] = w8_dequant8<T>(lo[0], lo[1], s);
        f[1] = w8_dequant8<T>(lo[2], lo[3], s);
        f[2] = w8_dequant8<T>(hi[0], hi[1], s);
        f[3] = w8_dequant8<T>(hi[2], hi[3], s);
    } else {
        const u32x4 unit = Wm[idx];
        const auto sc = MM::scale_pair(Sm + idx, true);
        u32 k_mask_lo = 0x000F000Fu, k_mask_hi = 0x00F000F0u, k_magic = MM::kMagic;
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = __builtin_bit_cast(u32x4, MM::dequant(unit[q], k_mask_lo, k_mask_hi, k_magic, sc));
    }
    const int bx = ct >> 3, p = ct & 7, nt = 2 * p + ((j >> 2) & 1), i = 4 * (j >> 3) + (j & 3);
    u32x4* dst = img + (((int64_t)bx * stages + t) * 16 + nt) * 64 + i;
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[16 * q] = f[q];
}

template <typename T>
__device__ __forceinline__ void mfma16_d(f32x4d& acc, const u32x4& w, const u32x4& a) {
    // accumulators pinned to the accumulation registers (w8a8_gemm256.hip: left to itself hipcc moves the 64 tiles between files)
    if constexpr (Act<T>::code == QL_DTYPE_F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}

// 4 waves as 2 x 2, wave tile 128 x 128 = 8 x 8 tiles of 16 x 16; ring of four stages; step t: barrier (stage t + 1 landed), 64 MFMAs with
// the 16 fragment reads of stage t + 1 and the 8 requests of stage t + 4 between them; vmcnt(16) at the barrier.
template <typename T, bool GATE>
__global__ __launch_bounds__(256) void dense256_kernel(const T* __restrict__ A, const char* __restrict__ img, int M, int N, int steps, int64_t lda,
                                                       int nbx, int super_rows, const T* __restrict__ bias, T* __restrict__ C, int64_t ldc,
                                                       const T* __restrict__ resid, int64_t ldr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // stage[4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int c16 = lane & 15, kq = lane >> 4;
    const TileXY tile = super_rows ? xcd_tile_super(blockIdx.x, gridDim.x, nbx, super_rows) : xcd_tile(blockIdx.x, gridDim.x, nbx);
    const int m0 = tile.y * 256, n0 = tile.x * 256;

    unsigned a_off[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int r = 16 * (4 * wave + n) + (lane >> 2), cp = lane & 3;
        const int row = (m0 + r < M) ? (m0 + r) : (M - 1);
        a_off[n] = (unsigned)row * (unsigned)(lda * (int64_t)sizeof(T)) + (unsigned)((cp ^ (((r >> 3) & 1) << 1)) * 16);
    }
    const unsigned long long a_base = sgpr64((unsigned long long)(uintptr_t)A);
    const unsigned long long b_base = sgpr64((unsigned long long)(uintptr_t)img + ((unsigned long long)tile.x * (unsigned long long)steps * 16ull +
                                                                                  (unsigned long long)(4 * wave)) * 1024ull);
    const unsigned b_voff = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)smem);
    const unsigned a_dma = lds0 + (unsigned)(4 * wave) * 1024u;
    const unsigned b_dma = lds0 + 16384u + (unsigned)(4 * wave) * 1024u;
    const int a_rd0 = (128 * wr + c16) * 64 + ((kq ^ (((c16 >> 3) & 1) << 1)) * 16);    // + mt * 1024
    const int b_rd0 = 16384 + (8 * wc) * 1024 + lane * 16;                               // + nt * 1024
    const char* a_rd[2] = {smem + a_rd0, smem + a_rd0 + 2 * kD256Stage};
    const char* b_rd[2] = {smem + b_rd0, smem + b_rd0 + 2 * kD256Stage};

    f32x4d acc[8][8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = f32x4d{0.f, 0.f, 0.f, 0.f};

    auto issue_piece = [&](int t, int buf, int q) {    // q = 0 .. 3: A pieces, 4 .. 7: the wave's four 1 KB tiles of the B stage
        const int k = t < steps ? t : steps - 1;       // past the end: the last stage again (never read; keeps the queue counts fixed)
        if (q < 4) glds16(a_dma + (unsigned)(buf * kD256Stage + q * 1024), a_off[q], sgpr64(a_base + (unsigned long long)k * 64ull));
        else glds16(b_dma + (unsigned)(buf * kD256Stage + (q - 4) * 1024), b_voff, sgpr64(b_base + ((unsigned long long)k * 16ull + (unsigned long long)(q - 4)) * 1024ull));
    };
    u32x4 fa[2][8], fb[2][8];
    auto read_a = [&](int buf, int mt, u32x4& x) { x = *reinterpret_cast<const u32x4*>(a_rd[buf >> 1] + (buf & 1) * kD256Stage + mt * 1024); };
    auto read_b = [&](int buf, int nt, u32x4& x) { x = *reinterpret_cast<const u32x4*>(b_rd[buf >> 1] + (buf & 1) * kD256Stage + nt * 1024); };

#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int q = 0; q < 8; ++q) issue_piece(st, st, q);
    vm_wait_imm<24>();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) { read_b(0, i, fb[0][i]); read_a(0, i, fa[0][i]); }

    auto step = [&](int t, auto bufc) {
        constexpr int BUF = decltype(bufc)::value, NXT = (BUF + 1) & 3, cur = BUF & 1, nxt = cur ^ 1;
        vm_wait_imm<16>();                             // this wave's pieces of stage t + 1 have landed (two younger stages in flight)
        __syncthreads();
        static_for<64>([&](auto qc) {
            constexpr int q = decltype(qc)::value, mt = q >> 3, nt = (mt & 1) ? 7 - (q & 7) : (q & 7);
            mfma16_d<T>(acc[mt][nt], fb[cur][nt], fa[cur][mt]);
            if constexpr (q < 32 && (q & 1) == 0) {
                constexpr int i = q >> 1;
                if constexpr (i < 8) read_b(NXT, i, fb[nxt][i]);
                else read_a(NXT, i - 8, fa[nxt][i - 8]);
            }
            if constexpr ((q & 7) == 5) issue_piece(t + 4, BUF, q >> 3);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    int t = 0;
    for (; t + 4 <= steps; t += 4) {
        step(t, std::integral_constant<int, 0>{});
        step(t + 1, std::integral_constant<int, 1>{});
        step(t + 2, std::integral_constant<int, 2>{});
        step(t + 3, std::integral_constant<int, 3>{});
    }
    if (t < steps) step(t, std::integral_constant<int, 0>{});
    if (t + 1 < steps) step(t + 1, std::integral_constant<int, 1>{});
    if (t + 2 < steps) step(t + 2, std::integral_constant<int, 2>{});
    vm_wait_imm<0>();

    // ---- epilogue: 8 consecutive columns of one row per lane and tile pair -> one 16-byte store -------------------------------------------
    const int mw = m0 + 128 * wr, nw = n0 + 128 * wc;
    const bool wide = GATE || ((ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
                               (!resid || ((ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(resid) & 15) == 0)));
    static_for<4>([&](auto ppc) {
        constexpr int pp = decltype(ppc)::value;
        const int nb = nw + 32 * pp + 8 * kq;          // first of the lane's 8 columns
        float bs[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) bs[r] = (bias && nb + r < N) ? Act<T>::load(bias + nb + r) : 0.f;
        static_for<8>([&](auto mtc) {
            constexpr int mt = decltype(mtc)::value;
            const int m = mw + 16 * mt + c16;
            float y[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                y[r] = Act<T>::round(r < 4 ? acc[mt][2 * pp][r & 3] : acc[mt][2 * pp + 1][r & 3]);
                if (bias) y[r] = Act<T>::round(y[r] + bs[r]);
            }
            if (m >= M || nb >= N) return;
            if constexpr (GATE) {                      // two (h0, h1, gate0, gate1) quads -> out[nb / 2 + 0..3] = round(round(silu(h)) * gate)
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float hv = y[4 * (i >> 1) + (i & 1)], gv = y[4 * (i >> 1) + 2 + (i & 1)];
                    o[i] = Act<T>::round(hv / (1.0f + __expf(-hv))) * gv;
                }
                T* dst = C + (int64_t)m * ldc + (nb >> 1);
                if (nb + 8 <= N) *reinterpret_cast<u32x2*>(dst) = u32x2{pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])};
                else *reinterpret_cast<u32*>(dst) = pack2<T>(o[0], o[1]);
            } else {
                T* dst = C + (int64_t)m * ldc + nb;
                if (wide && nb + 8 <= N) {
                    if (resid) {
                        float r[8];
                        unpack8<T>(*reinterpret_cast<const u32x4*>(resid + (int64_t)m * ldr + nb), r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) y[e] += r[e];
                    }
                    *reinterpret_cast<u32x4*>(dst) = pack8<T>(y);
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        if (nb + r < N) Act<T>::store(dst + r, resid ? y[r] + Act<T>::load(resid + (int64_t)m * ldr + nb + r) : y[r]);
                }
            }
        });
    });
}

size_t dense256_image_bytes(int64_t N, int64_t K) { return (size_t)((N + 255) / 256) * (size_t)(K / 32) * 16384; }

// the dense path needs whole 32-deep stages (four at least), 16-byte aligned rows and 32-bit byte offsets into A
bool dense256_can_run(int64_t M, int64_t N, int64_t K, int64_t lda, const void* A) {
    return M > 0 && N > 0 && K % 32 == 0 && K >= 128 && (lda * 2) % 16 == 0 && ((uintptr_t)A & 15) == 0 && M * lda * 2 < ((int64_t)1 << 31);
}

template <typename T, bool W8>
static int launch_expand(const void* tiled, const void* S, void* image, int64_t N, int64_t K, hipStream_t st) {
    const int64_t ctiles = (N + 31) / 32, ksteps = (K / 32 + 1) / 2, total = ctiles * ksteps * 64;
    const u32x4* Wm = (const u32x4*)tiled;
    const T* Sm = W8 ? (const T*)S : (const T*)((const char*)tiled + (size_t)total * 16);      // int4: Sm follows Wm in part 2 (launch.h)
    w4_expand_kernel<T, W8><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(Wm, Sm, (u32x4*)image, (int)N, (int)ksteps, (int)(K / 32), total);
    return finish_launch(QL_K_OTHER);
}

int dense256_expand(int dtype, bool w8, const void* tiled, const void* S, void* image, int64_t N, int64_t K, hipStream_t st) {
    if (dtype == QL_DTYPE_F16) return w8 ? launch_expand<f16, true>(tiled, S, image, N, K, st) : launch_expand<f16, false>(tiled, S, image, N, K, st);
    if (dtype == QL_DTYPE_BF16) return w8 ? launch_expand<__bf16, true>(tiled, S, image, N, K, st) : launch_expand<__bf16, false>(tiled, S, image, N, K, st);
    return QL_ERR_BAD_DTYPE;
}

template <typename T, bool GATE>
static int launch_dense(const void* A, const void* image, const void* bias, const void* resid, void* C, int64_t M, int64_t N, int64_t K,
                        int64_t lda, int64_t ldc, int64_t ldr, hipStream_t st) {
    static bool attr = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&dense256_kernel<T, GATE>), hipFuncAttributeMaxDynamicSharedMemorySize, kD256Lds) == hipSuccess;
    }();
    (void)attr;
    const int nbx = (int)((N + 255) / 256), nby = (int)((M + 255) / 256);
    const bool super = nby >= 2 && (nbx % 8 == 0 || nbx >= 32);
    dense256_kernel<T, GATE><<<(unsigned)(nbx * nby), 256, kD256Lds, st>>>(
        (const T*)A, (const char*)image, (int)M, (int)N, (int)(K / 32), lda, super ? nbx : xcd_order(nbx, nby, (double)M * K * 2, (double)N * K * 2),
        super ? 4 : 0, (const T*)bias, (T*)C, ldc, (const T*)resid, ldr);
    return finish_launch(QL_K_OTHER);
}

int dense256(int dtype, bool gate, const void* A, const void* image, const void* bias, const void* resid, void* C, int64_t M, int64_t N,
             int64_t K, int64_t lda, int64_t ldc, int64_t ldr, hipStream_t st) {
    if (dtype == QL_DTYPE_F16)
        return gate ? launch_dense<f16, true>(A, image, bias, nullptr, C, M, N, K, lda, ldc, 0, st)
                    : launch_dense<f16, false>(A, image, bias, resid, C, M, N, K, lda, ldc, ldr, st);
    if (dtype == QL_DTYPE_BF16)
        return gate ? launch_dense<__bf16, true>(A, image, bias, nullptr, C, M, N, K, lda, ldc, 0, st)
                    : launch_dense<__bf16, false>(A, image, bias, resid, C, M, N, K, lda, ldc, ldr, st);
    return QL_ERR_BAD_DTYPE;
}

}  // namespace ql
