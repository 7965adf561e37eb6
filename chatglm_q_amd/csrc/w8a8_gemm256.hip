// int8 activations x int8 weights for MANY rows (prefill of an int8 model with act_quant) on 256 x 256 output tiles, gfx950 - round 3.
//
//   acc_i32 = Aq (M,K) . W (N,K)^T on v_mfma_i32_32x32x32_i8, C = acc * a_scale[m] * w_scale[n] (+ bias)
//   (chatglm_q/int8/qlinear.py:56-62; the integer stage is exact)
//
// The structure of w4_gemm256.hip with nothing to dequantise: 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA tiles, a K
// tile of 128 bytes = 4 sub-steps = 32 MFMAs per wave, two LDS buffers per operand (128 KB), ONE block barrier per K tile placed in
// front of the tile's last sub-step, every load an LDS-DMA (vmq.h):
//   * A tile (256 rows x 128 bytes of the row-major int8 rows): 32 pieces of 1 KB, swizzle in the SOURCE address (chunk c of row r
//     at position 8 r + (c ^ ((r >> 1) & 7)): conflict-free ds_read_b128 at a 128-byte pitch);
//   * W tile: the tile-major derived copy (qlinear_w8_tile: [column tile][64-deep K step][half][lane][16 B]) IS fragment-major -
//     the four 1 KB units of a column tile and K tile are 4 KB contiguous in memory and land in LDS as they lie;
//   * no VALU in the loop at all: fragment reads, MFMAs, eight LDS-DMA requests per wave and K tile spread behind the MFMAs.
// Round 2's many-row kernel (w8a8_tiled_kernel, 128 x 128 tiles, W global -> VGPR, A through ds_write) reaches 1.6 - 1.7 POP/s at
// 8192 rows; DESIGN.md 4a named 256 x 256 tiles as its next step.
#include "launch.h"
#include "vmq.h"

#ifndef QL_I256_ABLATE
#define QL_I256_ABLATE 0
#endif

namespace ql {

typedef int i32x16 __attribute__((ext_vector_type(16)));

#ifdef QL_I256_STAMPS                                // developer build: lane 0 of waves 0 / 4 stamps the K loop's shader cycles
__device__ unsigned long long ql_i256_stamps[8192 * 2 * 4];
#endif
constexpr int kI256Tile = 256 * 128;               // one operand tile: 256 rows (columns) x 128 bytes of K
constexpr int kI256Lds = 4 * kI256Tile;            // A[2] | W[2]

#ifdef QL_DEV_TUNING                                // round 3's kernel: developer library only (QLINEAR_I256_RING=0), the A/B partner of the ring kernel below
// NW = 8: 2 (M) x 4 (N) waves, wave tile 128 x 64, two waves per SIMD (round 3).
// NW = 4: 2 x 2 waves, wave tile 128 x 128 = 4 x 4 MFMA tiles (256 accumulator registers), ONE wave per SIMD - the geometry of the
//         vendor's i8 kernel on this chip (rocprof: MT256x256x128, 256 threads; profiles/r04_vendor_gemm_diff.txt): half the
//         fragment reads per MFMA (8 per 16 instead of 6 per 8), a 4-wave barrier, no second wave competing for the SIMD's issue.
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void w8a8_gemm256_kernel(const int8_t* __restrict__ Aq, const int8_t* __restrict__ Wm, int M, int N, int K,
                                                               int nbx, int super_rows, const float* __restrict__ a_scale,
                                                               const T* __restrict__ Sc, const T* __restrict__ bias, T* __restrict__ C, int64_t ldc) {
    static_assert(NW == 8 || NW == 4, "8 waves (2 x 4) or 4 waves (2 x 2)");
    constexpr int NWN = NW / 2;                        // waves side by side in N
    constexpr int NT = 8 / NWN;                        // 32-column MFMA tiles per wave
    constexpr int PW = 32 / NW;                        // A pieces (and W units) a wave stages per K tile
    constexpr int CTW = 8 / NW;                        // W column tiles a wave stages (4 units each)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // A[2] | W[2]; reused by the epilogue
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / NWN, wc = wave % NWN;
    const int j = lane & 31, kb = lane >> 5;
    const TileXY tile = super_rows ? xcd_tile_super(blockIdx.x, gridDim.x, nbx, super_rows) : xcd_tile(blockIdx.x, gridDim.x, nbx);
    const int m0 = tile.y * 256, n0 = tile.x * 256;
    const int ktiles = K >> 7, ksteps64 = K >> 6;

    // ---- this wave's share of the staging work: A pieces PW wave .. PW wave + PW - 1 (8 rows each), W column tiles CTW wave .. ----
    unsigned a_off[PW];
#pragma unroll
    for (int n = 0; n < PW; ++n) {
        const int q = 64 * (PW * wave + n) + lane, r = q >> 3, cp = q & 7;
        const int row = (m0 + r < M) ? (m0 + r) : (M - 1);
        a_off[n] = (unsigned)row * (unsigned)K + (unsigned)((cp ^ ((r >> 1) & 7)) * 16);
    }
    const int ctiles = (N + 31) >> 5;
    const unsigned long long a_base = sgpr64((unsigned long long)(uintptr_t)Aq);
    unsigned long long w_base[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int ct_raw = tile.x * 8 + CTW * wave + c;
        const int ct = ct_raw < ctiles ? ct_raw : ctiles - 1;      // clamped: loads stay in bounds, stores are masked
        w_base[c] = sgpr64((unsigned long long)(uintptr_t)Wm + (unsigned long long)ct * (unsigned long long)ksteps64 * 2048ull);
    }
    const unsigned w_voff = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)smem);
    const unsigned a_dma = lds0 + (unsigned)(PW * wave) * 1024u;                         // + buffer * tile + n * 1024
    const unsigned w_dma = lds0 + (unsigned)(2 * kI256Tile) + (unsigned)(CTW * 4 * wave) * 1024u;
    char* w_lds = smem + 2 * kI256Tile;
    // fragment read offsets: sub-step s = (unit u, half h) of the K tile reads chunk 4 u + 2 kb + h of row 128 wr + 32 mt + j
    // (the K assignment of the tile-major weights, w8a8.hip) and unit (column tile NT wc + nt, s) of the W tile
    int a_rd[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a_rd[s] = ((128 * wr + j) * 8 + ((4 * (s >> 1) + 2 * kb + (s & 1)) ^ ((j >> 1) & 7))) * 16;
    const int w_rd = ((NT * wc) * 4 * 64 + lane) * 16;             // + nt * 4096 + s * 1024

    i32x16 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0;

    auto issue_piece = [&](int kt, int buf, int q) {   // q = 0 .. PW - 1: A pieces, PW .. 2 PW - 1: W units
#if QL_I256_ABLATE & 2                              // energy ablation: no requests in the loop (tiles 0 / 1 stay in LDS)
        if (kt >= 2) return;
#endif
        const int k = kt < ktiles ? kt : ktiles - 1;   // past the end: the last tile again (never read; keeps the queue counts fixed)
        if (q < PW) glds16(a_dma + (unsigned)(buf * kI256Tile + q * 1024), a_off[q], sgpr64(a_base + (unsigned long long)k * 128ull));
        else {
            const int u = q - PW;                      // unit u & 3 of the wave's column tile u >> 2
            glds16(w_dma + (unsigned)(buf * kI256Tile + u * 1024), w_voff, sgpr64(w_base[u >> 2] + (unsigned long long)(4 * k + (u & 3)) * 1024ull));
        }
    };
    i32x4 fa[2][4], fb[2][NT];
    auto read_frags = [&](int buf, int s, i32x4 (&xa)[4], i32x4 (&xb)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xb[nt] = *reinterpret_cast<const i32x4*>(w_lds + buf * kI256Tile + w_rd + nt * 4096 + s * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#if QL_I256_ABLATE & 1                              // energy ablation (results wrong): half the A fragment reads, the other half reused
            if (mt >= 2) { xa[mt] = xa[mt - 2]; continue; }
#endif
            xa[mt] = *reinterpret_cast<const i32x4*>(smem + buf * kI256Tile + mt * 4096 + a_rd[s]);
        }
    };

    // ---- prologue: tiles 0 and 1 requested; tile 0 landed ------------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < 2 * PW; ++q) issue_piece(0, 0, q);
#pragma unroll
    for (int q = 0; q < 2 * PW; ++q) issue_piece(1, 1, q);
    vm_wait_imm<2 * PW>();
    __syncthreads();
    read_frags(0, 0, fa[0], fb[0]);

    // ---- K loop (w4_gemm256.hip): sub-steps 0..2, barrier (tile kt + 1 complete, tile kt's last fragments in registers), then the
    // requests of tile kt + 2 one behind each MFMA of sub-step 3, with the first fragment reads of tile kt + 1 in front
    auto mma_sub = [&](int s) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s & 1][mt], fb[s & 1][nt], acc[mt][nt], 0, 0, 0);
    };
#ifdef QL_I256_STAMPS
    unsigned long long t_vm = 0, t_bar = 0;
#endif
    auto k_tile = [&](int kt, auto curc) {
        constexpr int cur = decltype(curc)::value, nxt = cur ^ 1;
        static_for<3>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            read_frags(cur, s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);         // the next fragments are requested before these MFMAs; sub-steps do not mix
            mma_sub(s);
            __builtin_amdgcn_sched_barrier(0);
        });
#if defined(QL_I256_STAMPS) && QL_I256_STAMPS > 1
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
        vm_wait_imm<0>();                              // tile kt + 1 has landed (this wave's pieces)
#if defined(QL_I256_STAMPS) && QL_I256_STAMPS > 1
        const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#if defined(QL_I256_STAMPS) && QL_I256_STAMPS > 1
        const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
        t_vm += ts1 - ts0;
        t_bar += ts2 - ts1;
#endif
        read_frags(nxt, 0, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        static_for<4 * NT>([&](auto qc) {              // 4 NT MFMAs, 2 PW = 64 / NW = 4 NT requests: one behind each
            constexpr int q = decltype(qc)::value, mt = q / NT, nt = q % NT;
            acc[mt][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[1][mt], fb[1][nt], acc[mt][nt], 0, 0, 0);
            issue_piece(kt + 2, cur, q);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    static_assert(4 * NT == 2 * PW, "one request per MFMA of the last sub-step");
#ifdef QL_I256_STAMPS
    const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#endif
    int kt = 0;
    for (; kt + 1 < ktiles; kt += 2) {
        k_tile(kt, std::integral_constant<int, 0>{});
        k_tile(kt + 1, std::integral_constant<int, 1>{});
    }
    if (kt < ktiles) k_tile(kt, std::integral_constant<int, 0>{});
#ifdef QL_I256_STAMPS
    if (lane == 0 && (wave % NWN) == 0 && blockIdx.x < 8192) {
        unsigned long long* o = ql_i256_stamps + ((size_t)blockIdx.x * 2 + wr) * 4;
        o[0] = t_vm; o[1] = t_bar; o[2] = t_loop0; o[3] = __builtin_amdgcn_s_memtime() - t_loop0;
    }
#endif
    vm_wait_imm<0>();                                  // the queue is empty before LDS is reused
    __syncthreads();                                   // ... and every wave is past its last fragment read

    // ---- epilogue: rank-1 scales, rounded 32 x 32 tiles through 2 KB of LDS per wave, 16-byte row chunks to global ----------------
    // (row tile outermost: its 16 activation scales are fetched once and live in 16 registers - with the column tile outermost
    // hipcc kept all 64 of them and spilled 252 bytes per lane, rocprof scratch column, round 3)
    const int mw = m0 + 128 * wr, nw = n0 + (256 / NWN) * wc;
    const bool wide = (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    T* lds_wave = reinterpret_cast<T*>(smem) + wave * 1024;
    float ws[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nw + 32 * nt + j;
        ws[nt] = Act<T>::load(Sc + (n < N ? n : N - 1));
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float asc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = mw + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
            asc[i] = a_scale[m < M ? m : M - 1];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nw + 32 * nt + j;
            if (wide) {
                store_tile_32x32<T>(lds_wave, C, ldc, mw + mt * 32, nw + 32 * nt, M, N, bias, lane,
                                    [&](int i) { return (float)acc[mt][nt][i] * (asc[i] * ws[nt]); });
            } else if (n < N) {
                const T* bn = bias ? bias + n : nullptr;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int m = mw + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
                    if (m < M) store_out<T>(C + (int64_t)m * ldc + n, (float)acc[mt][nt][i] * (asc[i] * ws[nt]), bn);
                }
            }
        }
    }
}

#endif  // QL_DEV_TUNING

// ---- round 4: the same GEMM as a RING OF FOUR 64-byte K stages on v_mfma_i32_16x16x64_i8, one wave per SIMD ----------------------------
// What the measurements of the kernel above said (profiles/r04_gemm_power.txt, tools/i256_timeline.py): its loop waits ~10 cycles per K
// tile for data and ~115 at the barrier, yet takes 2 850 cycles per 2 048 of MFMA issue at ~1.0 - 1.2 GHz - the chip is POWER bound, and
// (a) on random operands the 16x16 MFMA shapes sustain 12 - 17 % more than the 32x32 ones under the same cap (MFMA-only loops: 3.84 vs
// 3.29 POP/s; a 32x32 MFMA moves 4 KB of accumulators in and out per 64 K ops, a 16x16x64 one 1 KB per 32 K), (b) with ONE wave per
// SIMD everything between two MFMAs costs issue slots, and the 16 LDS-DMA requests (M0 save / set / restore each) sat behind the 16
// MFMAs of one sub-step, the 8 fragment reads in front of each sub-step.  Here:
//   * 4 waves as 2 x 2, wave tile 128 x 128 = 8 x 8 tiles of 16 x 16 (256 accumulator registers), one MFMA k-step (64 bytes) per stage;
//   * LDS = 4 stages x (A 256 rows x 64 B | W 256 columns x 64 B) = 128 KB.  Step t: MFMAs on the fragments of stage t (in registers),
//     fragment reads of stage t + 1 and the 8 requests of stage t + 4 (into the buffer stage t just left) spread BETWEEN its 64 MFMAs;
//     ONE barrier per step, vmcnt(16) in front of it: two stages stay in flight across every barrier;
//   * A stage image: row r = 64 bytes = 4 chunks, chunk c at position c ^ (2 (r >> 3 & 1)) - conflict-free for the four NON-contiguous
//     16-lane groups a ds_read_b128 is serviced in ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS; the first choice, c ^ (r >> 2 & 3),
//     is conflict-free for contiguous groups and measured SQ_LDS_BANK_CONFLICT = 1 / 3 of the LDS cycles), swizzle in the SOURCE
//     address of the LDS-DMA; W: the tile-major copy lands as it lies ([column tile][half][lane]) and a
//     lane (index i, k quarter q) of an MFMA tile reads unit half q & 1, lane 32 (q >> 1) + its column (mapping at w_rd0 below).
//     K order inside the dot product: quarter q of the MFMA = chunk q of both operands' 64 bytes;
//   * the WEIGHT fragment is the MFMA's first operand (D[i][j]: i = output column, j = row), and an MFMA tile's 16 columns are chosen
//     so that a lane ends up with EIGHT CONSECUTIVE columns of one row per tile pair: the epilogue stores 16-byte row pieces straight
//     from registers, 64 contiguous bytes per row and instruction - no LDS transposition.
typedef int i32x4v __attribute__((ext_vector_type(4)));
constexpr int kR4Stage = 32768;                    // A 16 KB | W 16 KB
constexpr int kR4Lds = 4 * kR4Stage;

// GATE: the weight copy is gate-interleaved (a first MLP projection: columns in (h, h, gate, gate) quads) and the epilogue applies
// SiLU * gate - the lane's 8 consecutive columns are two quads, 4 outputs, one 8-byte store; C has N / 2 columns.
template <typename T, bool GATE = false>
__global__ __launch_bounds__(256) void w8a8_gemm256_r4_kernel(const int8_t* __restrict__ Aq, const int8_t* __restrict__ Wm, int M, int N, int K,
                                                              int nbx, int super_rows, const float* __restrict__ a_scale,
                                                              const T* __restrict__ Sc, const T* __restrict__ bias, T* __restrict__ C, int64_t ldc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // stage[4]; reused by the epilogue
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int c16 = lane & 15, kq = lane >> 4;
    const TileXY tile = super_rows ? xcd_tile_super(blockIdx.x, gridDim.x, nbx, super_rows) : xcd_tile(blockIdx.x, gridDim.x, nbx);
    const int m0 = tile.y * 256, n0 = tile.x * 256;
    const int steps = K >> 6;                          // 64-byte K stages

    // ---- staging share of this wave: A pieces 4 wave .. 4 wave + 3 (16 rows x 64 B each), W column tiles 2 wave, 2 wave + 1 (2 KB each) ----
    unsigned a_off[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int r = 16 * (4 * wave + n) + (lane >> 2), cp = lane & 3;
        const int row = (m0 + r < M) ? (m0 + r) : (M - 1);
        a_off[n] = (unsigned)row * (unsigned)K + (unsigned)((cp ^ (((r >> 3) & 1) << 1)) * 16);
    }
    const int ctiles = (N + 31) >> 5;
    const unsigned long long a_base = sgpr64((unsigned long long)(uintptr_t)Aq);
    unsigned long long w_base[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int ct_raw = tile.x * 8 + 2 * wave + c;
        const int ct = ct_raw < ctiles ? ct_raw : ctiles - 1;      // clamped: loads stay in bounds, stores are masked
        w_base[c] = sgpr64((unsigned long long)(uintptr_t)Wm + (unsigned long long)ct * (unsigned long long)steps * 2048ull);
    }
    const unsigned w_voff = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)smem);
    const unsigned a_dma = lds0 + (unsigned)(4 * wave) * 1024u;                  // + stage * kR4Stage + n * 1024
    const unsigned w_dma = lds0 + 16384u + (unsigned)(4 * wave) * 1024u;         // + stage * kR4Stage + u * 1024
    // fragment read addresses: stages 0 / 1 through the 16-bit immediate of one base, stages 2 / 3 of a second one
    const int a_rd0 = (128 * wr + c16) * 64 + ((kq ^ (((c16 >> 3) & 1) << 1)) * 16);    // + mt * 1024
    // MFMA tile nt of the wave (column tile ct = 4 wc + (nt >> 1) of the block, b = nt & 1) does NOT cover 16 consecutive columns: its
    // index i = 4 q + e is column 8 q + 4 (b ^ (q >> 1)) + e of the 32, so that the lane holding D rows 4 q .. 4 q + 3 of tiles 2 p and
    // 2 p + 1 holds the EIGHT CONSECUTIVE columns 8 q .. 8 q + 7 of its row (one 16-byte store; which tile holds the lower four flips
    // with q >> 1) - and the 16 lanes of a fragment read still cover 256 distinct bytes modulo the bank row (conflict-free:
    // 8 (q & 1) + 4 (b ^ (q >> 1)) + e takes all 16 values).
    const int qd = c16 >> 2, ed = c16 & 3;
    const int w_rd0 = 16384 + (4 * wc) * 2048 + (kq & 1) * 1024 + (32 * (kq >> 1) + 8 * qd + 4 * (0 ^ (qd >> 1)) + ed) * 16;   // even tiles, + (nt >> 1) * 2048
    const int w_rd1 = 16384 + (4 * wc) * 2048 + (kq & 1) * 1024 + (32 * (kq >> 1) + 8 * qd + 4 * (1 ^ (qd >> 1)) + ed) * 16;   // odd tiles
    const char* a_rd[2] = {smem + a_rd0, smem + a_rd0 + 2 * kR4Stage};
    const char* w_rd[2][2] = {{smem + w_rd0, smem + w_rd0 + 2 * kR4Stage}, {smem + w_rd1, smem + w_rd1 + 2 * kR4Stage}};

    i32x4v acc[8][8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = i32x4v{0, 0, 0, 0};

    auto issue_piece = [&](int t, int buf, int q) {    // q = 0 .. 3: A pieces, 4 .. 7: W units (column tile (q - 4) >> 1, half (q - 4) & 1)
        const int k = t < steps ? t : steps - 1;       // past the end: the last stage again (never read; keeps the queue counts fixed)
        if (q < 4) glds16(a_dma + (unsigned)(buf * kR4Stage + q * 1024), a_off[q], sgpr64(a_base + (unsigned long long)k * 64ull));
        else {
            const int u = q - 4;
            glds16(w_dma + (unsigned)(buf * kR4Stage + u * 1024), w_voff, sgpr64(w_base[u >> 1] + (unsigned long long)(2 * k + (u & 1)) * 1024ull));
        }
    };
    i32x4 fa[2][8], fb[2][8];
    auto read_a = [&](int buf, int mt, i32x4& x) { x = *reinterpret_cast<const i32x4*>(a_rd[buf >> 1] + (buf & 1) * kR4Stage + mt * 1024); };
    auto read_b = [&](int buf, int nt, i32x4& x) {
        x = *reinterpret_cast<const i32x4*>(w_rd[nt & 1][buf >> 1] + (buf & 1) * kR4Stage + (nt >> 1) * 2048);
    };

    // ---- prologue: stages 0 .. 3 requested, stage 0 landed, its fragments read ------------------------------------------------------
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int q = 0; q < 8; ++q) issue_piece(st, st, q);
    vm_wait_imm<24>();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) { read_b(0, i, fb[0][i]); read_a(0, i, fa[0][i]); }

    // ---- step t (buffer t & 3 = BUF): barrier (stage t + 1 landed everywhere, every wave's fragments of stage t are in registers), then
    // 64 MFMAs with the 16 fragment reads of stage t + 1 behind the first 32 and the 8 requests of stage t + 4 behind every 8th
    auto step = [&](int t, auto bufc) {
        constexpr int BUF = decltype(bufc)::value, NXT = (BUF + 1) & 3, cur = BUF & 1, nxt = cur ^ 1;
        vm_wait_imm<16>();                             // this wave's pieces of stage t + 1 have landed (two younger stages in flight)
        __syncthreads();
        static_for<64>([&](auto qc) {
            constexpr int q = decltype(qc)::value, mt = q >> 3, nt = (mt & 1) ? 7 - (q & 7) : (q & 7);   // serpentine: B fragment reused across the turn
            // accumulators pinned to the accumulation registers ("+a": left to its own allocation hipcc kept part of the 64 tiles in
            // VGPRs and moved them back and forth around every step, with spills).  The operands come from ds_read_b128 (hipcc
            // places the lgkmcnt waits for the "v" inputs itself); nothing reads an accumulator before the epilogue.
            // The WEIGHT fragment is the first operand: D[i][j] has i = output column, j = row, so a lane ends up with FOUR CONSECUTIVE
            // COLUMNS of one row (column 4 (lane >> 4) + r, row lane & 15) and the epilogue stores 8-byte row pieces straight from
            // registers - no LDS transposition (the 32x32 kernels' epilogue: 16 two-byte LDS stores per tile and lane).
            asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(fb[cur][nt]), "v"(fa[cur][mt]));
            if constexpr (q < 32 && (q & 1) == 0) {
                constexpr int i = q >> 1;              // 16 reads: B 0..7 first (all needed by the next step's first row of tiles), then A
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
    vm_wait_imm<0>();                                  // the queue is empty before LDS is reused
    __syncthreads();                                   // ... and every wave is past its last fragment read

    // ---- epilogue: rank-1 scales, 8 consecutive columns of one row per lane and tile pair -> one 16-byte store --------------------------
    const int mw = m0 + 128 * wr, nw = n0 + 128 * wc;
    const bool wide = (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    const bool flip = (kq >> 1) != 0;                  // quads 2, 3: the odd tile of a pair holds the lower four columns
    float asc[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int m = mw + 16 * mt + c16;
        asc[mt] = a_scale[m < M ? m : M - 1];
    }
    static_for<4>([&](auto ppc) {
        constexpr int pp = decltype(ppc)::value;
        const int nb = nw + 32 * pp + 8 * kq;          // first of the lane's 8 columns
        float ws[8], bs[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int n = nb + r < N ? nb + r : N - 1;
            ws[r] = Act<T>::load(Sc + n);
            bs[r] = bias ? Act<T>::load(bias + n) : 0.f;
        }
        static_for<8>([&](auto mtc) {
            constexpr int mt = decltype(mtc)::value;
            const int m = mw + 16 * mt + c16;
            float y[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int a0 = acc[mt][2 * pp][r & 3], a1 = acc[mt][2 * pp + 1][r & 3];
                const int av = ((r < 4) != flip) ? a0 : a1;
                float p = (float)av * (asc[mt] * ws[r]);
                asm volatile("" : "+v"(p));            // the fp32 product exists (the reference's Cast, Mul, then the output dtype:
                y[r] = Act<T>::round(p);               // chatglm_q/int8/qlinear.py:60-62); left fusable hipcc rounds product -> f16 once
                if (bias) y[r] = GATE ? Act<T>::round(y[r] + bs[r]) : y[r] + bs[r];     // (the plain store rounds the sum itself)
            }
            if constexpr (GATE) {
                if (m < M && nb < N) {                 // out[nb / 2 + i] = round(round(silu(h_i)) * gate_i)   (chatglm_q/model.py:200-201)
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float hv = y[4 * (i >> 1) + (i & 1)], gv = y[4 * (i >> 1) + 2 + (i & 1)];
                        o[i] = Act<T>::round(hv / (1.0f + __expf(-hv))) * gv;
                    }
                    T* dst = C + (int64_t)m * ldc + (nb >> 1);
                    const u32x2 packed = {pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])};
                    if (nb + 8 <= N) __builtin_nontemporal_store(packed, reinterpret_cast<u32x2*>(dst));
                    else *reinterpret_cast<u32*>(dst) = packed[0];                             // N % 8 == 4: the last quad alone
                }
            } else if (m < M && nb < N) {
                T* dst = C + (int64_t)m * ldc + nb;
                if (wide && nb + 8 <= N)                // non-temporal: the output is read by the NEXT launch at the earliest (interleaved A/B, 8192 x 4096 x
                    __builtin_nontemporal_store(pack8<T>(y), reinterpret_cast<u32x4*>(dst));   // 4096: 128.8 -> 123.6 us, 2.13 -> 2.22 POP/s)
                else
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        if (nb + r < N) Act<T>::store(dst + r, y[r]);
            }
        });
    });
}

template <typename T, bool GATE = false>
static int launch_i256_r4(const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M,
                          int64_t N, int64_t K, int64_t ldc, hipStream_t st) {
    static bool attr_set = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_gemm256_r4_kernel<T, GATE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kR4Lds) == hipSuccess;
    }();
    (void)attr_set;
    const int nbx = (int)((N + 255) / 256), nby = (int)((M + 255) / 256);
    const bool no_super = QL_TUNE("QLINEAR_GEMM_SUPER", 1) == 0;
    const int sy = QL_TUNE("QLINEAR_GEMM_SY", 4);
    // grouped order (ql_common.h: xcd_tile_super) for column counts that are a multiple of 8 or wide (w_in: 107 column tiles, int8 x int8
    // 1.80 -> 2.05 POP/s, int4g32 +1.5 %); 18 column tiles (qkv_proj) measured better in whole rows (tools/ab/run_sy_sweep.sh: 1.78 vs 1.68 POP/s)
    const bool super = !no_super && nby >= 2 && (nbx % 8 == 0 || nbx >= 32);
    w8a8_gemm256_r4_kernel<T, GATE><<<(unsigned)(nbx * nby), 256, kR4Lds, st>>>(
        Aq, Wm, (int)M, (int)N, (int)K, super ? nbx : xcd_order(nbx, nby, (double)M * K, (double)N * K), super ? sy : 0, a_scale,
        (const T*)S, (const T*)bias, (T*)C, ldc);
    return finish_launch(QL_K_W8A8_GEMM256);
}

#ifdef QL_DEV_TUNING
template <typename T, int NW>
static int launch_i256_nw(const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M,
                          int64_t N, int64_t K, int64_t ldc, hipStream_t st) {
    static bool attr_set = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_gemm256_kernel<T, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kI256Lds) == hipSuccess;
    }();
    (void)attr_set;
    const int nbx = (int)((N + 255) / 256), nby = (int)((M + 255) / 256);
    const bool no_super = QL_TUNE("QLINEAR_GEMM_SUPER", 1) == 0;
    const int sy = QL_TUNE("QLINEAR_GEMM_SY", 4);
    // grouped order (ql_common.h: xcd_tile_super) for column counts that are a multiple of 8 or wide (w_in: 107 column tiles, int8 x int8
    // 1.80 -> 2.05 POP/s, int4g32 +1.5 %); 18 column tiles (qkv_proj) measured better in whole rows (tools/ab/run_sy_sweep.sh: 1.78 vs 1.68 POP/s)
    const bool super = !no_super && nby >= 2 && (nbx % 8 == 0 || nbx >= 32);
    w8a8_gemm256_kernel<T, NW><<<(unsigned)(nbx * nby), NW * 64, kI256Lds, st>>>(
        Aq, Wm, (int)M, (int)N, (int)K, super ? nbx : xcd_order(nbx, nby, (double)M * K, (double)N * K), super ? sy : 0, a_scale,
        (const T*)S, (const T*)bias, (T*)C, ldc);
    return finish_launch(QL_K_W8A8_GEMM256);
}

#endif
#ifndef QL_I256_NW
#define QL_I256_NW 4
#endif
#ifndef QL_I256_RING
#define QL_I256_RING 1
#endif
template <typename T>
static int launch_i256(const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M,
                       int64_t N, int64_t K, int64_t ldc, hipStream_t st) {
#ifdef QL_DEV_TUNING
    if (!QL_TUNE("QLINEAR_I256_RING", QL_I256_RING)) {
        if (QL_TUNE("QLINEAR_I256_NW", QL_I256_NW) != QL_I256_NW)
            return launch_i256_nw<T, 12 - QL_I256_NW>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
        return launch_i256_nw<T, QL_I256_NW>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
    }
#endif
    return launch_i256_r4<T>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
}

// what the kernel needs: 16-bit outputs, whole 128-byte K tiles (two at least), 32-bit byte offsets into Aq
bool w8a8_gemm256_can_run(int dtype, int64_t M, int64_t N, int64_t K, const void* Aq) {
    return (dtype == QL_DTYPE_F16 || dtype == QL_DTYPE_BF16) && M > 0 && N > 0 && K % 128 == 0 && K >= 256 && ((uintptr_t)Aq & 15) == 0 &&
           M * K < ((int64_t)1 << 31);
}
// ... and when the launcher of w8a8.hip takes it: the grid pays in whole rounds of 256 blocks (one block per CU at a time)
bool w8a8_gemm256_supported(int dtype, int64_t M, int64_t N, int64_t K, const void* Aq) {
    const int min_blocks = QL_TUNE("QLINEAR_W8A8_256_MIN_BLOCKS", 0);   // tuning sweeps (developer build)
    if ((dispatch_flags() & QL_D_NO256) || !w8a8_gemm256_can_run(dtype, M, N, K, Aq)) return false;
    const int64_t blocks = ((N + 255) / 256) * ((M + 255) / 256);
    if (min_blocks > 0) return blocks >= min_blocks;
    const int64_t cus = cu_count(), rounds = (blocks + cus - 1) / cus;
    return blocks >= cus && blocks * 10 >= rounds * cus * 7;
}

// gate-interleaved copy of a first MLP projection, SiLU * gate in the epilogue: C is (M, N / 2)
int w8a8_gemm256_gated(int dtype, const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M,
                       int64_t N, int64_t K, int64_t ldc, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_i256_r4<f16, true>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
    case QL_DTYPE_BF16: return launch_i256_r4<__bf16, true>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

int w8a8_gemm256(int dtype, const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M,
                 int64_t N, int64_t K, int64_t ldc, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_i256<f16>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
    case QL_DTYPE_BF16: return launch_i256<__bf16>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

}  // namespace ql

#if defined(QL_I256_STAMPS) && defined(QL_DEV_TUNING)
extern "C" int qlinear_i256_stamps_read(unsigned long long* out, int blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ql::ql_i256_stamps), sizeof(unsigned long long) * 8 * blocks);
}
#endif
