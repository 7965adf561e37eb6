// fp32 activations, MANY rows, on the fp32 matrix instruction (gfx950) - round 5.
//
//   C[M,N] = A[M,K] . dequant(W)        int4g32: (n - 8) * s   (chatglm_q/int4/triton_ops.py:66-80 with fp32 operands: tl.dot serves fp32 at any M;
//                                       the reference's own kernel tests are fp32, tests/test_triton_ops_int4.py:11-22, atol = rtol = 1e-4)
//                                       int8 per channel: b * s[n]   (chatglm_q/int8/triton_ops.py:62-73)
//   every weight dequantised in fp32 with ONE rounding (the product of a small integer and an fp32 scale), products and sums in fp32 on
//   v_mfma_f32_32x32x2_f32 (exact fp32 FMAs: no reduced-precision path is involved), bias added after the sum.
//
// Why: fp32 activations with more than 4 rows ran on the canonical-layout VALU split-K kernel (w4_kernels.hip) - correct, and ~20x
// under what the matrix cores do with the same operands (missing 2).  `torch_dtype: "float32"` is a valid load
// configuration of the reference (chatglm_q/loader.py:16-38).
//
// Structure (the plain one: at 64 cycles per MFMA the loop is matrix-pipe bound with time to spare around it):
//   * reads the CANONICAL buffers - no derived copy for fp32: int4 (K / 2, N) bytes + (K / 32, N) fp32 scales; int8 (N, K) rows (K contiguous);
//   * block = 256 threads = 4 waves as 2 x 2 on a (32 MT x 2) x 128 output tile, wave tile (32 MT) x 64, K tile = 32 = one int4 group;
//   * both operand tiles live in LDS as [row or column][32 k] fp32 with a 36-float pitch (16-byte fragment reads, conflict-free);
//     a lane (index i = lane & 31, half kb = lane >> 5) reads k = 8 q + 4 kb + 0..3 in one ds_read_b128 and feeds four MFMA steps
//     with them - step t contracts k = 8 q + t and 8 q + 4 + t, the same pairing on both operands;
//   * global loads of tile kt + 1 are issued before the MFMAs of tile kt and written to the other LDS buffer behind them: one barrier per tile;
//   * 73 KB of LDS per block: two blocks per CU overlap each other's barriers.
#include "launch.h"
#include "ql_common.h"

namespace ql {

typedef float f32x16f __attribute__((ext_vector_type(16)));

constexpr int kF32Pitch = 36;                      // floats per LDS row (32 k + 4 of padding: 144 bytes)
constexpr int kF32BTile = 128 * kF32Pitch * 4;     // B tile bytes: 128 columns
constexpr int kF32Fold = 16;                       // K tiles per first-level sum (512 k)

// W8 = false: Wq = canonical int4 (K / 2, N) bytes, S = (K / 32, N) fp32.  W8 = true: Wq = int8 (N, K) rows with row stride ldw, S = (N) fp32.
template <bool W8, int MT>
__global__ __launch_bounds__(256, 2) void wq_gemm_f32_kernel(const float* __restrict__ A, const uint8_t* __restrict__ Wq, const float* __restrict__ S,
                                                             int M, int N, int K, int64_t lda, int64_t ldw, const float* __restrict__ bias,
                                                             float* __restrict__ C, int64_t ldc, int nbx) {
    constexpr int BM = 64 * MT;                        // block rows
    constexpr int kATile = BM * kF32Pitch * 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // A[2] | B[2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i32 = lane & 31, kb = lane >> 5;
    const TileXY tile = xcd_tile(blockIdx.x, gridDim.x, nbx);
    const int m0 = tile.y * BM, n0 = tile.x * 128;
    const int ktiles = K >> 5;

    // ---- staging share: A: BM rows x 8 16-byte chunks = 8 BM chunks, 2 MT per thread; B: column c = tid & 127, half h = tid >> 7 (16 k) ----
    f32x4 ar[2 * MT];                                 // (ext_vector_type: the HIP struct type kept these arrays in scratch)
    const int bc = tid & 127, bh = tid >> 7;
    const int bn = n0 + bc < N ? n0 + bc : N - 1;      // clamped column: loads stay in bounds, stores are masked
    u32 wraw[4];                                       // int4: 8 bytes = 16 nibbles (k = 16 bh ..); int8: 16 bytes
    float bs = 0.f;
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int u = 0; u < 2 * MT; ++u) {
            const int q = tid + 256 * u, r = q >> 3, c4 = q & 7;
            const int row = m0 + r < M ? m0 + r : M - 1;
            ar[u] = *reinterpret_cast<const f32x4*>(A + (int64_t)row * lda + kt * 32 + 4 * c4);
        }
        if constexpr (W8) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(Wq + (int64_t)bn * ldw + kt * 32 + 16 * bh);
            wraw[0] = v[0]; wraw[1] = v[1]; wraw[2] = v[2]; wraw[3] = v[3];
        } else {
            const uint8_t* wp = Wq + ((int64_t)kt * 16 + 8 * bh) * N + bn;   // byte rows 8 bh .. 8 bh + 7 of the group: k = 16 bh + 2 r, + 1
            u32 lo = 0, hi = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lo |= (u32)wp[(int64_t)r * N] << (8 * r);
                hi |= (u32)wp[(int64_t)(r + 4) * N] << (8 * r);
            }
            wraw[0] = lo; wraw[1] = hi;
            bs = S[(int64_t)kt * N + bn];
        }
    };
    if constexpr (W8) bs = S[bn];
    auto store_tile = [&](int buf) {
        char* a_lds = smem + buf * kATile;
        char* b_lds = smem + 2 * kATile + buf * kF32BTile;
#pragma unroll
        for (int u = 0; u < 2 * MT; ++u) {
            const int q = tid + 256 * u, r = q >> 3, c4 = q & 7;
            *reinterpret_cast<f32x4*>(a_lds + (r * kF32Pitch + 4 * c4) * 4) = ar[u];
        }
        float w[16];
        if constexpr (W8) {
#pragma unroll
            for (int e = 0; e < 16; ++e) w[e] = (float)((int)(wraw[e >> 2] << (24 - 8 * (e & 3))) >> 24) * bs;     // b * s: one rounding
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {              // byte r: low nibble = k 16 bh + 2 r, high nibble = the next k   (int4/quantizer.py:24-28)
                const u32 b = (wraw[r >> 2] >> (8 * (r & 3))) & 0xFFu;
                w[2 * r] = (float)((int)(b & 0xFu) - 8) * bs;                                                       // (n - 8) * s: one rounding
                w[2 * r + 1] = (float)((int)(b >> 4) - 8) * bs;
            }
        }
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4)
            *reinterpret_cast<f32x4*>(b_lds + (bc * kF32Pitch + 16 * bh + 4 * v4) * 4) = f32x4{w[4 * v4], w[4 * v4 + 1], w[4 * v4 + 2], w[4 * v4 + 3]};
    };

    // two-level sums: the MFMA chain adds K-tile after K-tile into `acc`, sequentially; every kF32Fold K tiles (512 k) the partial sums
    // are folded into `tot` and restarted - the rounding noise of a 4096- or 13696-term fp32 chain (~sqrt(K) ulp of the running sum,
    // 3.5e-4 absolute at 512 x 4096 x 4096 with int8 weights: over the reference's 1e-4 bar) drops to that of 512-term chains
    f32x16f acc[MT][2], tot[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = tot[mt][nt][e] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) load_tile(kt + 1);        // in flight behind this tile's 32 MT MFMAs
        const char* a_lds = smem + cur * kATile + ((32 * MT * wr + i32) * kF32Pitch + 4 * kb) * 4;
        const char* b_lds = smem + 2 * kATile + cur * kF32BTile + ((64 * wc + i32) * kF32Pitch + 4 * kb) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                  // k = 8 q + 4 kb + 0..3
            f32x4 fa[MT], fb[2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const f32x4*>(a_lds + (mt * 32 * kF32Pitch + 8 * q) * 4);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) fb[nt] = *reinterpret_cast<const f32x4*>(b_lds + (nt * 32 * kF32Pitch + 8 * q) * 4);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][t], fb[nt][t], acc[mt][nt], 0, 0, 0);
        }
        if (kt + 1 < ktiles) store_tile(cur ^ 1);      // the other buffer: its readers passed the barrier of tile kt - 1
        if ((kt & (kF32Fold - 1)) == kF32Fold - 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    tot[mt][nt] += acc[mt][nt];
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] += tot[mt][nt];

    // ---- epilogue: a lane holds column i32 and 16 rows of each 32 x 32 tile: 128 contiguous bytes per row and store instruction ----
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + 64 * wc + 32 * nt + i32;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + 32 * MT * wr + 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * kb;
                if (m < M) C[(int64_t)m * ldc + n] = bias ? acc[mt][nt][e] + bv : acc[mt][nt][e];
            }
    }
}

template <bool W8, int MT>
static int launch_f32(const float* A, const uint8_t* Wq, const float* S, const float* bias, float* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                      int64_t ldw, int64_t ldc, hipStream_t st) {
    constexpr int lds = 2 * (64 * MT) * kF32Pitch * 4 + 2 * kF32BTile;
    static bool attr_set = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&wq_gemm_f32_kernel<W8, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    }();
    (void)attr_set;
    const int nbx = (int)((N + 127) / 128), nby = (int)((M + 64 * MT - 1) / (64 * MT));
    wq_gemm_f32_kernel<W8, MT><<<(unsigned)(nbx * nby), 256, lds, st>>>(A, Wq, S, (int)M, (int)N, (int)K, lda, ldw, bias, C, ldc,
                                                                         xcd_order(nbx, nby, (double)M * K * 4, (double)N * K * (W8 ? 1.0 : 0.5)));
    return finish_launch(W8 ? QL_K_W8_GEMM128 : QL_K_W4_GEMM128);
}

// rows from which the matrix-core kernel is taken for fp32 activations (below: the one-to-four-row passes / the canonical-layout kernel).
// Measured (profiles/r05_f32_rows.txt, 4096 x 4096): a block walks all of K by itself, ~200 us at K = 4096 whatever M is - the VALU kernels
// (split K over workgroups) are ahead up to 96 rows (8: 28 vs 190 us, 32: 66 vs 191, 96: 178 vs 216), behind from 128 (231 vs 205), 3.7 x at 512, 5.5 x at 2048
bool wq_gemm_f32_serves(int64_t M, int64_t N, int64_t K) { return M >= 128 && N >= 32 && K % 32 == 0 && K >= 64; }

int w4_gemm_f32(const void* A, const uint8_t* Wq, const void* S, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                int64_t ldc, hipStream_t st) {
    // 64-row tiles while 128-row tiles would leave most of the 512 block slots (256 CUs x 2) empty
    if (((N + 127) / 128) * ((M + 127) / 128) < 384)
        return launch_f32<false, 1>((const float*)A, Wq, (const float*)S, (const float*)bias, (float*)C, M, N, K, lda, N, ldc, st);
    return launch_f32<false, 2>((const float*)A, Wq, (const float*)S, (const float*)bias, (float*)C, M, N, K, lda, N, ldc, st);
}

int w8_gemm_f32(const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t ldw,
                int64_t lda, int64_t ldc, hipStream_t st) {
    if (((N + 127) / 128) * ((M + 127) / 128) < 384)
        return launch_f32<true, 1>((const float*)A, (const uint8_t*)W, (const float*)S, (const float*)bias, (float*)C, M, N, K, lda, ldw, ldc, st);
    return launch_f32<true, 2>((const float*)A, (const uint8_t*)W, (const float*)S, (const float*)bias, (float*)C, M, N, K, lda, ldw, ldc, st);
}

}  // namespace ql
