// int8 activations x int8 weights on the matrix cores (gfx950): BASELINE config 3's path, round 2.
//
//   act_quant_rows_kernel     row-wise (or per-tensor) symmetric int8 activation quantisation in ONE pass over the
//                             row: 16-byte loads, the row stays in registers between the max and the quantise,
//                             16-byte stores (quantize_int8, chatglm_q/int8/quantizer.py:11-19; per-tensor:
//                             DynamicQuantizeMatMul.symbolic branch 2, chatglm_q/int8/qlinear.py:64-70)
//   w8a8_tiled_kernel         acc_i32 = Aq (M,K) . W (N,K)^T on v_mfma_i32_32x32x32_i8, rank-1 scale epilogue
//                             (chatglm_q/int8/qlinear.py:60-62), with the weights read from the tile-major derived
//                             copy (qlinear_w8_tile): a lane's 16 bytes ARE its MFMA B fragment, so W never touches
//                             LDS - only the (shared) activation tile does.
//
// Why this shape (numbers for 512 x 4096 x 4096, the configuration the metric is quoted on): one 64 x 128 output tile
// per CU is the only decomposition that fills 256 CUs without split-K slabs (8 MB of int32 each); it needs
// (64 + 128) x 4096 = 768 KB per CU through the CU's 64 B/clk vector-memory path, i.e. >= 12.3 k cycles against 8.2 k
// cycles of MFMA work - the kernel is bound by operand delivery, and everything that is not a load (LDS traffic,
// barriers, address arithmetic) has to hide behind it.  Round 1's kernel staged BOTH operands through LDS with one
// wave per SIMD and a block barrier per 128-byte K step (20.7 us).  Here:
//   * W goes global -> VGPR -> MFMA, 4 KB contiguous per wave and K step, 3-4 steps in flight;
//   * 8 waves per block = two K-parity groups of 4 waves: group g owns the K steps t = g, g + 2, ...  (int32 partial
//     sums add exactly; they are combined through LDS once, at the end).  Two waves per SIMD at different phases of
//     their step fill each other's barrier / LDS-latency gaps - the same total MFMA work, twice the latency hiding;
//   * the A tile is staged global -> VGPR -> LDS one step ahead, XOR-swizzled so that fragment reads are conflict-free.
#include <type_traits>
#include <utility>

// developer ablation switches for tools/w8a8_ablate.sh (always 0 in the shipped library): 1 no MFMA, 2 no steady-state W
// loads, 4 no steady-state A loads, 8 no LDS fragment reads, 16 no block barriers in the K loop, 32 no LDS stores
#ifndef QL_W8A8_ABLATE
#define QL_W8A8_ABLATE 0
#endif
#include "launch.h"
#include "ql_common.h"
#include "vmq.h"

namespace ql {

typedef int i32x16 __attribute__((ext_vector_type(16)));

// QL_W8A8_STAMPS (developer build, tools/w8a8_timeline.sh): thread 0 of every block stamps s_memrealtime (100 MHz) at
// fixed points of the GEMM kernel - a measured timeline of one launch instead of inferences from ablations.
#ifdef QL_W8A8_STAMPS
constexpr int kStampBlocks = 4096, kStampPoints = 8;
__device__ unsigned long long ql_w8a8_stamps[kStampBlocks * kStampPoints];
#define QL_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < ql::kStampBlocks) \
        ql::ql_w8a8_stamps[blockIdx.x * ql::kStampPoints + (i)] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define QL_STAMP(i) do { } while (0)
#endif

// =============================================================================================
// activation quantisation
// =============================================================================================
// MODE 0: row-wise, scale from this row's maximum, written to a_scale[m]
// MODE 1: maxima only: a_scale[m] = max_k |A[m,k]| (no quantisation)         - per-tensor pass 1
// MODE 2: quantise with the scale derived from max(tensor_rowmax[0..M))      - per-tensor pass 2 (a_scale untouched)
// One block per row, thread t owns VPT adjacent 16-byte vectors (8 values each) of the row.
template <typename T, int VPT, int MODE>
__global__ __launch_bounds__(256) void act_quant_rows_kernel(const T* __restrict__ A, int8_t* __restrict__ Aq,
                                                             float* __restrict__ a_scale, int K, int64_t lda,
                                                             const float* __restrict__ tensor_rowmax, int M) {
    static_assert(sizeof(T) == 2, "16-bit activations (fp32 takes the two-pass kernel)");
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int nvec = K >> 3;
    const u32x4* src = reinterpret_cast<const u32x4*>(A + (int64_t)m * lda);
    u32x4 r[VPT];
    float v[VPT][8];
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int vi = tid * VPT + u;
        r[u] = src[vi < nvec ? vi : nvec - 1];                    // clamped: every load unconditional
    }
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        unpack8<T>(r[u], v[u]);
        if (tid * VPT + u < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(v[u][e]));
        }
    }
    float s;
    if constexpr (MODE == 2) {
        float tm = 0.f;
        for (int i = tid; i < M; i += 256) tm = fmaxf(tm, tensor_rowmax[i]);
        mx = tm;
    }
    mx = wave_max_dpp(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if constexpr (MODE == 1) {
        if (tid == 0) a_scale[m] = mx;
        return;
    }
    s = fmaxf(mx / 127.0f, 1e-10f);                               // quantizer.py:18 (clamp keeps 0 / 0 out)
    if (MODE == 0 && tid == 0) a_scale[m] = s;
    int8_t* dst = Aq + (int64_t)m * K;
    u32 packed[VPT][2];
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32 w = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float q = rintf(v[u][4 * h + e] / s);             // true division, round half to even: bit-exact vs torch
                q = fminf(fmaxf(q, -127.f), 127.f);
                w |= ((u32)(int)q & 0xFFu) << (8 * e);
            }
            packed[u][h] = w;
        }
    }
    if constexpr (VPT % 2 == 0) {
#pragma unroll
        for (int u = 0; u < VPT; u += 2) {
            const int vi = tid * VPT + u;
            if (vi + 1 < nvec) {
                *reinterpret_cast<u32x4*>(dst + (int64_t)vi * 8) = u32x4{packed[u][0], packed[u][1], packed[u + 1][0], packed[u + 1][1]};
            } else if (vi < nvec) {
                *reinterpret_cast<u32x2*>(dst + (int64_t)vi * 8) = u32x2{packed[u][0], packed[u][1]};
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int vi = tid * VPT + u;
            if (vi < nvec) *reinterpret_cast<u32x2*>(dst + (int64_t)vi * 8) = u32x2{packed[u][0], packed[u][1]};
        }
    }
}

// fp32 activations / rows too long for the register-resident kernel / misaligned rows: two passes over the row
template <typename T, int MODE>
__global__ __launch_bounds__(256) void act_quant_rows_generic_kernel(const T* __restrict__ A, int8_t* __restrict__ Aq,
                                                                     float* __restrict__ a_scale, int K, int64_t lda,
                                                                     const float* __restrict__ tensor_rowmax, int M) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const T* a = A + (int64_t)m * lda;
    float mx = 0.f;
    if constexpr (MODE == 2) {
        for (int i = tid; i < M; i += 256) mx = fmaxf(mx, tensor_rowmax[i]);
    } else {
        for (int k = tid; k < K; k += 256) mx = fmaxf(mx, fabsf(Act<T>::load(a + k)));
    }
    mx = wave_max_dpp(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if constexpr (MODE == 1) {
        if (tid == 0) a_scale[m] = mx;
        return;
    }
    const float s = fmaxf(mx / 127.0f, 1e-10f);
    if (MODE == 0 && tid == 0) a_scale[m] = s;
    int8_t* q = Aq + (int64_t)m * K;
    for (int k = tid; k < K; k += 256) {
        float v = rintf(Act<T>::load(a + k) / s);
        v = fminf(fmaxf(v, -127.f), 127.f);
        q[k] = (int8_t)v;
    }
}

// per-tensor pass 3: every a_scale[m] = the one scale (computed from the row maxima BEFORE any of them is overwritten)
__global__ __launch_bounds__(256) void act_scale_broadcast_kernel(float* __restrict__ a_scale, int M) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float mx = 0.f;
    for (int i = tid; i < M; i += 256) mx = fmaxf(mx, a_scale[i]);
    mx = wave_max_dpp(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();                                              // every read of a_scale precedes every write
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float s = fmaxf(mx / 127.0f, 1e-10f);
    for (int i = tid; i < M; i += 256) a_scale[i] = s;
}

template <typename T, int MODE>
static int launch_act_quant_mode(const void* A, int8_t* Aq, float* a_scale, int64_t M, int64_t K, int64_t lda,
                                 const float* rowmax, hipStream_t st) {
    const bool vec = sizeof(T) == 2 && K % 8 == 0 && (lda % 8 == 0 || M == 1) && ((uintptr_t)A & 15) == 0 &&
                     ((uintptr_t)Aq & 15) == 0 && K <= 256 * 8 * 8;
    if constexpr (sizeof(T) == 2) {
        if (vec) {
            const int64_t nvec = K / 8;
#define QL_AQ(VPT_) act_quant_rows_kernel<T, VPT_, MODE><<<(unsigned)M, 256, 0, st>>>((const T*)A, Aq, a_scale, (int)K, lda, rowmax, (int)M)
            if (nvec <= 256) QL_AQ(1);
            else if (nvec <= 512) QL_AQ(2);
            else if (nvec <= 1024) QL_AQ(4);
            else QL_AQ(8);
#undef QL_AQ
            return finish_launch(QL_K_ACT_QUANT);
        }
    }
    act_quant_rows_generic_kernel<T, MODE><<<(unsigned)M, 256, 0, st>>>((const T*)A, Aq, a_scale, (int)K, lda, rowmax, (int)M);
    return finish_launch(QL_K_ACT_QUANT);
}

template <typename T>
static int launch_act_quant(const void* A, int8_t* Aq, float* a_scale, int64_t M, int64_t K, int64_t lda, bool per_tensor,
                            hipStream_t st) {
    if (!per_tensor) return launch_act_quant_mode<T, 0>(A, Aq, a_scale, M, K, lda, nullptr, st);
    // per-tensor symmetric (ONNX branch 2): row maxima -> quantise every row with the tensor's scale -> broadcast the scale
    int rc = launch_act_quant_mode<T, 1>(A, Aq, a_scale, M, K, lda, nullptr, st);
    if (rc) return rc;
    rc = launch_act_quant_mode<T, 2>(A, Aq, a_scale, M, K, lda, a_scale, st);
    if (rc) return rc;
    act_scale_broadcast_kernel<<<1, 256, 0, st>>>(a_scale, (int)M);
    return finish_launch();
}

int act_quant_rowwise(int dtype, const void* A, int8_t* Aq, float* a_scale, int64_t M, int64_t K, int64_t lda, bool per_tensor,
                      hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F32: return launch_act_quant<float>(A, Aq, a_scale, M, K, lda, per_tensor, st);
    case QL_DTYPE_F16: return launch_act_quant<f16>(A, Aq, a_scale, M, K, lda, per_tensor, st);
    case QL_DTYPE_BF16: return launch_act_quant<__bf16>(A, Aq, a_scale, M, K, lda, per_tensor, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

// =============================================================================================
// W8A8 GEMM on the tile-major weights
//   Wm[ct][kt][h][lane][16 B] (w8_tile_kernel, w8_gemm.hip): lane = 32 kb + j, half h holds bytes k = 64 kt + 32 kb + 16 h
//   + 0..15 of output channel 32 ct + j, zero padded.  One 32-deep MFMA sub-step consumes one half: sub-step (u, h) of a
//   128-byte K step (u = which 64-byte unit, h = which half) uses k = 128 t + 64 u + 32 kb + 16 h + 0..15, so the
//   A fragment of lane (i, kb) is the 16-byte chunk 4 u + 2 kb + h of row i's 128-byte line (any K assignment that
//   is the same for both operands is valid: the contraction is a sum over k).
//   Block = 8 waves: wave w serves column tile (w & 3) of the block's 128 columns and K-parity group (w >> 2).
//   LDS: per group two A-tile buffers of BM x 128 bytes, chunk c of row r at chunk position 8 r + (c ^ ((r >> 1) & 7)).
// =============================================================================================
// NG = K-parity groups per block: 2 (8 waves; the shape for few row tiles, where a CU gets ONE block and the two groups
// are its only latency hiding) or 1 (4 waves, <= 256 registers, no combine step: two independent blocks per CU whose
// prologues / epilogues overlap each other's K loop - the many-row shape; one 8-wave block per CU showed 6.4 us of setup,
// first-tile latency and epilogue around every 19 us loop, profiles/r02_w8a8_timeline.txt).
// COLG (with NG = 2): the two groups are COLUMN groups instead - the block covers 256 columns, both groups walk the same K
// chunks and share ONE A tile in LDS (half the A bytes per flop from L2 / the fabric, no exchange at the end).
// ---- NITER + ADMA form: the wave's vector-memory queue counted by hand --------------------------------------------------------
// Every load of the K pipeline is an inline-asm statement (hipcc sees none of them and inserts no vmcnt waits of its own):
// W fragments global -> VGPR, A pieces global -> LDS by LDS-DMA (no VGPR round trip, no ds_write pass: the ablation builds put
// the ds_write pass at 2.7 of the loop's 8.4 us).  Loads complete in issue order, so "operand X has landed" = "at most N
// younger loads are still outstanding"; the issue order is fixed at compile time, these functions restate it.
//   prologue:            A(0) W(0) [PF PF] A(1) W(1) A(2)                      (A(k): 4 pieces, W(k): 4 S fragments of chunk-iteration k)
//   iteration j, behind the MFMAs of sub-step s:   W(j + 2, s)   then   A(j + 3, s) if s < 4
#ifndef QL_W8A8_PF
#define QL_W8A8_PF 0                                // L2-prefetch instructions per wave behind W(0) (0: none; 2 measured 18.5 vs 17.1 us cold, 16.7 vs 14.0 us cache-hot: a 64-line gather costs the load path more than it saves)
#endif
constexpr int kAdmaPieces = 4, kAdmaPf = QL_W8A8_PF;
constexpr int adma_prologue(int subs) { return 3 * kAdmaPieces + 2 * subs + kAdmaPf; }      // subs: W fragments per chunk iteration (4 S)
constexpr int adma_min(int a, int b) { return a < b ? a : b; }
constexpr int adma_issued_before(int subs, int niter, int j, int s) {      // loads issued before sub-step s of iteration j begins (closed
    // form, no loops: inner loops with a trip count that depends on the unrolled index kept hipcc from unrolling the K loop)
    return adma_prologue(subs) + subs * adma_min(j, niter > 2 ? niter - 2 : 0) + kAdmaPieces * adma_min(j, niter > 3 ? niter - 3 : 0) +
           (j + 2 < niter ? s : 0) + (j + 3 < niter ? adma_min(s, kAdmaPieces) : 0);
}
constexpr int adma_idx_w(int subs, int niter, int k, int s) {              // position of W(k, s) in the issue order
    return k == 0 ? kAdmaPieces + s : k == 1 ? 2 * kAdmaPieces + subs + kAdmaPf + s : adma_issued_before(subs, niter, k - 2, s);
}
constexpr int adma_idx_a_last(int subs, int niter, int k) {                // position of the last piece of A(k)
    return k == 0 ? kAdmaPieces - 1 : k == 1 ? 2 * kAdmaPieces + subs + kAdmaPf - 1 : k == 2 ? adma_prologue(subs) - 1
                  : adma_issued_before(subs, niter, k - 3, kAdmaPieces - 1) + 1;
}
// NITER > 0: the K loop fully unrolled for exactly NITER iterations per K-parity group (K = NITER * KP * BK, no K tail): every
// prefetch condition is a compile-time constant, so hipcc's vmcnt waits are exact counts.  Round 3, read in the ISA: with
// K = 4096 (8 iterations) the generic form below never enters its steady-state loop (it needs U + DEPTH + 2 = 10 iterations)
// and runs ALL of config 3 in the peeled tail, whose conditional loads make the compiler drain the queue (vmcnt(0)) at the
// head of every iteration - load latency, LDS staging and MFMAs in series, 1.2 us per iteration where each alone is 0.3 - 0.5.
// The unrolled form also spreads the next loads / LDS stores over the sub-steps (one W load behind each MFMA pair) instead
// of issuing 10 KB per wave in one burst at the end of the iteration, where all 8 waves queue at the CU's 64 B/clk load path.
// ---- SK = 2 (round 6): a GRID-level K split on top of the two K-parity groups of a block ----------------------------------------
// Config 3 (512 x 4096 x 4096) covers the chip with 64 x 128 tiles = 256 blocks x (64 + 128) rows x 4096 B = 201 MB pulled out of the L2s
// for 25 MB of operands; its load-only floor is 12.7 us from HBM (tools/microbench/l2_share_probe2.hip).  128 x 128 tiles x 2 K slices
// are 256 blocks too, pull 134 MB (floor 9.3 us) and cost one exact int32 hand-off per tile pair: the two blocks of a tile take a
// ticket from a per-tile counter at ENTRY (the atomic's round trip hides under the K loop); the even ticket PUBLISHES its 128 x 128
// partial (64 KB, lane-linear 16-byte pieces), release, flag = ticket + 1; the odd ticket waits for the flag (its partner holds a ticket,
// so it is running: no dependence on an unscheduled block), acquire, adds the pieces and runs the epilogue.  Integer sums: bit-equal
// to the unsplit kernel.  splitk_ws: uint32 tickets[kSplitkMaxTiles] | flags[kSplitkMaxTiles] | partial tiles; zeroed ONCE by the host,
// never reset (tickets only count up: their parity is the role, their value the epoch) - the layout does not depend on the shape, so
// launches of different shapes may take turns on one workspace (a shape-dependent layout let one shape's flags become another's tickets:
// odd first tickets, finishers without a publisher - caught by tools/w8a8_splitk_debug.py).
constexpr int kSplitkMaxTiles = 1024;
template <typename T, int MT, int S, int DEPTH, int NG, bool COLG = false, int NITER = 0, bool ADMA = false, int SK = 1>
__global__ __launch_bounds__(NG * 256, NG == 1 ? 2 : 1) void w8a8_tiled_kernel(const int8_t* __restrict__ Aq, const int8_t* __restrict__ Wm, int M,
                                                         int N, int K, int nbx, int rotate, int super_rows, const float* __restrict__ a_scale,
                                                         const T* __restrict__ S_, const T* __restrict__ bias,
                                                         T* __restrict__ C, int64_t ldc, unsigned* __restrict__ splitk_ws = nullptr) {
    // One loop iteration of a K-parity group covers a CHUNK of BK = 128 S bytes of K (S 128-byte steps = 2 S tile-major
    // units = 4 S MFMA sub-steps): S = 2 halves the barriers per byte; MT = 4 keeps S = 1 (LDS).
    constexpr int BM = 32 * MT;
    constexpr int BK = 128 * S;
    constexpr int CPR = 8 * S;                         // 16-byte chunks per tile row
    constexpr int KP = COLG ? 1 : NG;                  // K-parity groups
    constexpr int BN = COLG ? 128 * NG : 128;          // columns per block
    constexpr int STG = COLG ? NG * 256 : 256;         // threads staging one A tile
    constexpr int NCH = BM * CPR;                      // chunks per tile
    constexpr int ACH = (NCH + STG - 1) / STG;         // ... staged per thread per iteration
    constexpr bool kAllStage = NCH % STG == 0;
    constexpr int BUF = BM * BK;                       // one A-tile buffer
    constexpr int NS = 4 * S;                          // MFMA sub-steps per chunk
    const T* __restrict__ Sc = S_;
    constexpr int NB = ADMA ? 4 : 3;                   // A-tile buffers per K-parity group
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 groups][NB buffers][BUF]; reused by the epilogue

    const int tid = threadIdx.x, lane = tid & 63, tg = COLG ? tid : tid & 255;
    QL_STAMP(0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform on purpose: chunk indices, the K-tail
    const int grp = NG == 1 ? 0 : wave >> 2, wv = wave & 3;       // tests and the buffer bases then live in SGPRs
    const int j = lane & 31, kb = lane >> 5;
    static_assert(SK == 1 || (SK == 2 && NITER > 0 && ADMA && !COLG && NG == 2), "the grid-level K split exists for the unrolled LDS-DMA form");
    int slice = 0, tile_id = 0;
    TileXY tile;
    if constexpr (SK == 1) {
        tile = super_rows ? xcd_tile_super(blockIdx.x, gridDim.x, nbx, super_rows) : xcd_tile(blockIdx.x, gridDim.x, nbx);
    } else {
        // an XCD's contiguous positions = (tile, slice) pairs, slice fastest (partners share an L2), tiles column-major: the XCD owns
        // whole column tiles x every row tile x both slices - the sharing pattern the probe priced
        const unsigned c = blockIdx.x & 7u, i = blockIdx.x >> 3, q8 = gridDim.x >> 3, r8 = gridDim.x & 7u;
        const unsigned p = (c < r8 ? c * (q8 + 1) : r8 * (q8 + 1) + (c - r8) * q8) + i;
        const unsigned nby = (gridDim.x / SK) / (unsigned)nbx;
        tile_id = (int)(p / SK);
        slice = (int)(p % SK);
        tile = TileXY{(int)((unsigned)tile_id / nby), (int)((unsigned)tile_id % nby)};
    }
    __shared__ unsigned s_ticket;
    if constexpr (SK > 1) {
        if (tid == 0) s_ticket = __hip_atomic_fetch_add(splitk_ws + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int m0 = tile.y * BM, n0 = tile.x * BN;
    const int kgrp = COLG ? 0 : grp;                   // K-parity index of this group
    const int ksteps64 = (K + 63) >> 6;                // tile-major units per column tile
    const int nchunks = (K + BK - 1) / BK;
    const int chunk0 = SK == 1 ? 0 : slice * (nchunks / SK);    // first K chunk of this block's slice (SK > 1: nchunks % (SK KP) == 0)
    const int niter = SK > 1 ? nchunks / (SK * KP) : KP == 1 ? nchunks : (nchunks + 1) >> 1;   // both groups run the same number of iterations (barriers!)
    const int ctiles = (N + 31) >> 5;
    const int ct_raw = tile.x * (BN / 32) + (COLG ? grp * 4 : 0) + wv;
    const int8_t* wbase = Wm + (int64_t)(ct_raw < ctiles ? ct_raw : ctiles - 1) * ksteps64 * 2048 + lane * 16;
    char* lds_a = smem + kgrp * (NB * BUF);

    // swizzle: chunk c of row r sits at chunk position CPR r + (c ^ x(r)); 128-byte rows: x = (r >> 1) & 7, 256-byte
    // rows (one row = all 64 banks): x = r & 15 - fragment reads (16 lanes = 16 rows, one chunk index) conflict-free
    auto swz = [](int r) { return S == 1 ? ((r >> 1) & 7) : (r & 15); };
    const int8_t* a_src[ACH];
    int a_dst[ACH];
#pragma unroll
    for (int u = 0; u < ACH; ++u) {
        const int q = tg + u * STG, r = (q / CPR) % BM, c = q % CPR;
        a_src[u] = Aq + (int64_t)((m0 + r < M) ? (m0 + r) : (M - 1)) * K + c * 16;
        a_dst[u] = (r * CPR + (c ^ swz(r))) * 16;
    }
    const int c_mine = tg % CPR;                       // (tg + STG u) % CPR is the same for every u
    const int klast = K - 16;                          // last in-bounds 16-byte chunk start (K % 16 == 0)
    // fragment read offsets: sub-step s = (step, unit, half) reads chunk 8 step + 4 unit + 2 kb + half of row mt * 32 + j;
    // swz(mt * 32 + j) does not depend on mt, so ONE offset per sub-step serves every row tile (+ mt * 32 rows: immediate)
    int a_rd[NS];
#pragma unroll
    for (int sub = 0; sub < NS; ++sub) {
        const int c = 8 * (sub >> 2) + 4 * ((sub >> 1) & 1) + 2 * kb + (sub & 1);
        a_rd[sub] = (j * CPR + (c ^ swz(j))) * 16;
    }

    i32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0;

    struct Stage {
        i32x4 a[ACH];                                  // A chunk of iteration (slot's iteration + 2): see below
        i32x4 w[NS];                                   // W fragments of the slot's iteration
    };
    Stage st[DEPTH];
    // WHAT BOUNDS THE K LOOP (round 2, measured: tools/w8a8_timeline.sh stamps, ablation builds, five loop structures).
    // 2 or 3 LDS buffers, 128 or 256 bytes of K per barrier, fragments 1 sub-step / 2 sub-steps / a whole chunk ahead, ring
    // depth 2 / 3 / 4 / 6, lane-contiguous or 32-byte-strided weight tiles, K rotation: the K loop of 512 x 4096 x 4096 takes
    // 8.2 - 12.6 us in every one of them, i.e. the 256 CUs pull their 196 MB of operands (every W line is wanted by 8 CUs,
    // every A line by 32) through the L2s at 17 - 24 TB/s whatever the issue pattern; MORE loads in flight (ring depth 6)
    // make it slower (12.5 us), fewer barriers do not help, an MFMA-only ablation of the loop runs in 2.5 us.  A 64 x 128
    // tile per CU is the least-traffic way to cover 512 x 4096 on 256 CUs without split-K slabs, so config 3 is bound by
    // operand delivery from L2 (~9 us of loop), not by its 3.4 us of MFMA work.  profiles/r02_w8a8_timeline.txt.
    // K ROTATION (see the header): block (x, y) starts at phase (2 y + x mod 4) of its K walk.  Measured neutral at
    // 512 x 4096 x 4096 (19.2 vs 19.4 us) - kept behind QLINEAR_W8A8_ROTATE=1 for experiments, default off.
    const int phases = niter < 16 ? niter : 16;
    // NITER form: the blocks that share a weight panel (the row tiles of one column tile, all on one XCD) start at different K
    // chunks, so that between them their first iteration requests the WHOLE panel from HBM at once; walking K in lockstep they
    // all wait for the same few lines and the compulsory misses of the launch go out 1/NITER at a time (tools/microbench/
    // l2_share_probe.hip, this pattern with weights from HBM: 15.3 us per launch in lockstep, 11.7 us rotated, 8.8 us L2-hot)
    const int rot = NITER > 0 ? ((rotate > 0 && NITER % rotate == 0) ? (tile.y % rotate) * (NITER / rotate) : 0) : rotate > 0 ? ((2 * tile.y + (tile.x & 3)) % phases) * (niter / phases) : 0;
    auto chunk_of = [&](int i) {                       // K chunk of loop iteration i (i may run past niter: clamped by users)
        int r = i + rot;
        r = r >= niter ? r - niter : r;
        return chunk0 + KP * r + kgrp;
    };
    auto load_w = [&](int i, Stage& sg) {
        int t = chunk_of(i < niter ? i : niter - 1);
        t = t < nchunks ? t : nchunks - 1;
#pragma unroll
        for (int u = 0; u < 2 * S; ++u) {
            int ua = 2 * S * t + u;
            ua = ua < ksteps64 ? ua : ksteps64 - 1;
            const int8_t* w0 = wbase + (int64_t)ua * 2048;
            sg.w[2 * u] = *reinterpret_cast<const i32x4*>(w0);
            sg.w[2 * u + 1] = *reinterpret_cast<const i32x4*>(w0 + 1024);
        }
    };
    auto load_a = [&](int i, i32x4 (&dst)[ACH]) {
        int t = chunk_of(i < niter ? i : niter - 1);
        t = t < nchunks ? t : nchunks - 1;
        const int off = t * BK + c_mine * 16 <= klast ? t * BK : klast - c_mine * 16;
#pragma unroll
        for (int u = 0; u < ACH; ++u) dst[u] = *reinterpret_cast<const i32x4*>(a_src[u] + off);
    };
    auto store_a = [&](int buf, const i32x4 (&src)[ACH]) {
#pragma unroll
        for (int u = 0; u < ACH; ++u)
            if (!(QL_W8A8_ABLATE & 32) && (kAllStage || tg + u * STG < NCH))
                *reinterpret_cast<i32x4*>(lds_a + buf * BUF + a_dst[u]) = src[u];
    };
    auto read_a = [&](int buf, int sub, i32x4 (&fr)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (QL_W8A8_ABLATE & 8) fr[mt] = i32x4{a_rd[sub], buf, mt, sub};
            else fr[mt] = *reinterpret_cast<const i32x4*>(lds_a + buf * BUF + mt * 32 * BK + a_rd[sub]);
        }
    };
    auto mma = [&](const i32x4 (&fa)[MT], const i32x4& w) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (QL_W8A8_ABLATE & 1) acc[mt][0] += fa[mt][0] ^ fa[mt][1] ^ w[0] ^ w[3];
            else acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[mt], w, acc[mt], 0, 0, 0);
        }
    };
    // Fragments run ONE SUB-STEP AHEAD, across the iteration boundary too: the A tile of iteration i + 1 is complete in
    // LDS since the barrier that ended iteration i - 1 (three buffers), so its first fragments are requested before the
    // barrier that ends iteration i and the MFMAs of i + 1 start without an LDS round trip behind the barrier.
    // (Two sub-steps of lookahead were measured too: 20.9 against 18.9 us at 512 x 4096 x 4096 - the registers cost more
    // than the LDS latency they hide.)
    i32x4 fa[2][MT];
    auto mma_chunk_full = [&](int buf, int nbuf, const i32x4 (&w)[NS]) {
#pragma unroll
        for (int sub = 0; sub < NS; ++sub) {
            if (sub + 1 < NS) read_a(buf, sub + 1, fa[(sub + 1) & 1]);
            else read_a(nbuf, 0, fa[0]);
            mma(fa[sub & 1], w[sub]);
            __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);            // next sub-step's DS reads first
            __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);            // then this sub-step's MFMAs
        }
    };
    // last chunk of K (or none left for this group): only `units` of its 2 S tile-major units exist
    auto mma_chunk_partial = [&](int buf, int nbuf, const i32x4 (&w)[NS], int units) {
#pragma unroll
        for (int u = 0; u < 2 * S; ++u) {
            if (u < units) {                                               // wave-uniform
                i32x4 f1[MT];
                if (u > 0) read_a(buf, 2 * u, fa[0]);
                read_a(buf, 2 * u + 1, f1);
                mma(fa[0], w[2 * u]);
                mma(f1, w[2 * u + 1]);
            }
        }
        read_a(nbuf, 0, fa[0]);
    };
    auto mma_chunk = [&](int i, int buf, int nbuf, const i32x4 (&w)[NS]) {
        const int t = chunk_of(i);
        int units = ksteps64 - 2 * S * t;
        units = t >= nchunks ? 0 : units;
        if (units >= 2 * S) mma_chunk_full(buf, nbuf, w);
        else mma_chunk_partial(buf, nbuf, w, units);
    };

    if constexpr (NITER > 0 && ADMA) {
        static_assert(DEPTH == 2 && ACH == kAdmaPieces && NS >= kAdmaPieces && kAllStage, "the issue order restated by adma_*()");
        // per-lane source offsets of the A pieces: LDS-DMA writes lane-linear, so the swizzle goes into the SOURCE address -
        // the thread at chunk position q of the tile (row q / CPR, position cp = q % CPR) fetches source chunk cp ^ swz(row)
        unsigned a_off[ACH];
#pragma unroll
        for (int u = 0; u < ACH; ++u) {
            const int q = tg + u * STG, r = q / CPR, cp = q % CPR;
            a_off[u] = (unsigned)((m0 + r < M) ? (m0 + r) : (M - 1)) * (unsigned)K + (unsigned)((cp ^ swz(r)) * 16);
        }
        const unsigned lds_grp = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)smem + (unsigned)(kgrp * (NB * BUF)) + (unsigned)((tg >> 6) * 1024)));
        const unsigned w_off = (unsigned)lane * 16u;
        const int ct = ct_raw < ctiles ? ct_raw : ctiles - 1;
        const unsigned long long a_base = sgpr64((unsigned long long)(uintptr_t)Aq);
        const unsigned long long w_base = sgpr64((unsigned long long)(uintptr_t)Wm + (unsigned long long)ct * (unsigned long long)ksteps64 * 2048ull);
        auto issue_a = [&](int k, int u) {             // piece u of A(k) -> buffer k % NB
            const int t = chunk_of(k);
            glds16(lds_grp + (unsigned)((k % NB) * BUF + u * STG * 16), a_off[u], sgpr64(a_base + (unsigned long long)(t * BK)));
        };
        i32x4 wr[DEPTH][NS];
        auto issue_w = [&](int k, int sub) {           // fragment `sub` of W(k) -> register slot k % DEPTH
            const int t = chunk_of(k);
            gload16(wr[k % DEPTH][sub], w_off, sgpr64(w_base + (unsigned long long)((2 * S * t + (sub >> 1)) * 2048 + (sub & 1) * 1024)));
        };
        QL_STAMP(7);
#pragma unroll
        for (int u = 0; u < ACH; ++u) issue_a(0, u);
#pragma unroll
        for (int sub = 0; sub < NS; ++sub) issue_w(0, sub);
        // L2 prefetch (cold weights): one dword per 128-byte line, 64 lines = 8 KB per instruction.  The row tiles that share this
        // column tile each touch ANOTHER K chunk of the wave's weight stream (between them: all of it), and the XCD's 256 waves
        // touch 8 KB each of the activation tile rows - every compulsory miss of the launch is in flight behind the first tile's
        // loads instead of going out one pipeline stage at a time (tools/microbench/l2_share_probe.hip: 15.2 -> 12.2 us)
        int pf0 = 0, pf1 = 0;
        if constexpr (kAdmaPf == 2) {
            const int tp = KP * (tile.y % NITER) + kgrp;
            const int pw = (int)(blockIdx.x >> 3) * (NG * 4) + wave;               // wave index within the XCD (speed only)
            const int64_t a_bytes = (int64_t)M * K;
            int64_t a_pf = (int64_t)pw * 8192 + lane * 128;
            a_pf = a_pf < a_bytes ? a_pf : a_bytes - 4;
            asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(pf0) : "v"((unsigned)lane * 128u), "s"(sgpr64(w_base + (unsigned long long)(2 * S * tp) * 2048ull)) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(pf1) : "v"(Aq + a_pf) : "memory");
        }
#pragma unroll
        for (int u = 0; u < ACH; ++u) issue_a(1, u);
#pragma unroll
        for (int sub = 0; sub < NS; ++sub) issue_w(1, sub);
#pragma unroll
        for (int u = 0; u < ACH; ++u) issue_a(2, u);
        QL_STAMP(1);
        vm_wait_imm<adma_prologue(NS) - adma_idx_a_last(NS, NITER, 1) - 1>();     // A(0), A(1) have landed (this wave's pieces)
        __syncthreads();
        QL_STAMP(2);
        read_a(0, 0, fa[0]);
        static_for<NITER>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int buf = i % NB, nbuf = (i + 1) % NB;
            static_for<NS>([&](auto sc) {
                constexpr int sub = decltype(sc)::value;
                if constexpr (sub + 1 < NS) read_a(buf, sub + 1, fa[(sub + 1) & 1]);
                else if constexpr (i + 1 < NITER) read_a(nbuf, 0, fa[0]);
                __builtin_amdgcn_sched_barrier(0);
                vm_wait_imm<adma_issued_before(NS, NITER, i, sub) - adma_idx_w(NS, NITER, i, sub) - 1>(wr[i % DEPTH][sub]);
                if constexpr (i == 1 && sub == 0 && kAdmaPf == 2) asm volatile("" : "+v"(pf0), "+v"(pf1));   // older than W(1): landed; registers free from here
                mma(fa[sub & 1], wr[i % DEPTH][sub]);
                if constexpr (i + 2 < NITER) issue_w(i + 2, sub);
                if constexpr (sub < ACH && i + 3 < NITER) issue_a(i + 3, sub);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (i + 2 < NITER) vm_wait_imm<adma_issued_before(NS, NITER, i + 1, 0) - adma_idx_a_last(NS, NITER, i + 2) - 1>();   // A(i + 2) landed
            if constexpr (i + 1 < NITER) __syncthreads();
        });
        QL_STAMP(3);
    } else {
    // prologue: A tiles of iterations 0 and 1 into LDS, ring slot d <- W of iteration d and A of iteration d + 2.
    // Request order = need order: the first MFMA needs A(0) and W(0) only.
    {
        i32x4 a0[ACH], a1[ACH];
        QL_STAMP(7);                                   // address setup done: the first load goes now
        load_a(0, a0);
        load_w(0, st[0]);
        load_a(1, a1);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (d > 0) load_w(d, st[d]);
            load_a(d + 2, st[d].a);
        }
        QL_STAMP(1);                                   // setup done, every prologue load issued
        store_a(0, a0);
        store_a(1, a1);
    }
    __syncthreads();
    QL_STAMP(2);                                       // first A tiles in LDS: the first MFMA can go
    read_a(0, 0, fa[0]);

    // K loop.  UNIFIED = one loop shape for every iteration (ring depth a multiple of 3: slot = i % DEPTH, LDS buffer =
    // i % 3): every load unconditional - past the end it re-reads the last chunk (cache hits, never used) - and the trip
    // count rounded up to the unroll (an extra iteration does no MFMAs).  The peeled form (main loop + tail with
    // conditional loads) drains the ring in every tail iteration (hipcc's vmcnt bookkeeping turns conservative: vmcnt(0)
    // in the ISA), and with K = 4096 the tail is most of the loop.
    if constexpr (NITER > 0) {
        static_assert(DEPTH == 2 && ACH <= NS / 2, "schedule below: ACH stores, then ACH loads, over the NS sub-steps");
#pragma unroll
        for (int i = 0; i < NITER; ++i) {
            const int d = i % DEPTH, buf = i % 3, nbuf = (i + 1) % 3, wbuf = (i + 2) % 3;
            const int tw = chunk_of(i + DEPTH), ta = chunk_of(i + DEPTH + 2);              // chunks fetched this iteration
#pragma unroll
            for (int sub = 0; sub < NS; ++sub) {
                if (sub + 1 < NS) read_a(buf, sub + 1, fa[(sub + 1) & 1]);
                else if (i + 1 < NITER) read_a(nbuf, 0, fa[0]);
                __builtin_amdgcn_sched_barrier(0);                                 // the next fragments are requested BEFORE these MFMAs
                mma(fa[sub & 1], st[d].w[sub]);
                if (i + DEPTH < NITER && !(QL_W8A8_ABLATE & 2))                    // the fragment just consumed is re-requested
                    st[d].w[sub] = *reinterpret_cast<const i32x4*>(wbase + (int64_t)(2 * S * tw + (sub >> 1)) * 2048 + (sub & 1) * 1024);
                if (sub < ACH && i + 2 < NITER && !(QL_W8A8_ABLATE & 32) && (kAllStage || tg + sub * STG < NCH))
                    *reinterpret_cast<i32x4*>(lds_a + wbuf * BUF + a_dst[sub]) = st[d].a[sub];
                if (sub >= NS - ACH && i + DEPTH + 2 < NITER && !(QL_W8A8_ABLATE & 4))
                    st[d].a[sub - (NS - ACH)] = *reinterpret_cast<const i32x4*>(a_src[sub - (NS - ACH)] + ta * BK);
                // (pinned with sched_barrier: given sched_group_barrier hints hipcc moved the fragment reads BEHIND the MFMAs of the
                // sub-step before and waited lgkmcnt(0) in front of every MFMA - no lookahead left, seen in the ISA)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(QL_W8A8_ABLATE & 16)) __syncthreads();
        }
        QL_STAMP(3);
    } else if constexpr (DEPTH % 3 == 0) {
        for (int it = 0; it < niter; it += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int i = it + d;
                const int buf = d % 3, nbuf = (d + 1) % 3, wbuf = (d + 2) % 3;
                store_a(wbuf, st[d].a);
                const int t = chunk_of(i);
                int units = ksteps64 - 2 * S * t;
                units = (i >= niter || t >= nchunks) ? 0 : units;
                if (units >= 2 * S) mma_chunk_full(buf, nbuf, st[d].w);
                else mma_chunk_partial(buf, nbuf, st[d].w, units);
                load_w(i + DEPTH, st[d]);
                load_a(i + DEPTH + 2, st[d].a);
                if (!(QL_W8A8_ABLATE & 16)) __syncthreads();
            }
        }
        QL_STAMP(3);
    } else {
    // main loop, unrolled so that ring slot and LDS buffers are compile-time constants.  Iteration i: the A chunk of
    // iteration i + 2 (in this slot since DEPTH iterations) goes to buffer (i + 2) % 3; MFMAs on buffer i % 3 with the
    // slot's W fragments; THEN the slot is re-loaded with W of i + DEPTH and A of i + DEPTH + 2 (loading first would need
    // a register copy of the pending tile, for which hipcc drains vmcnt(0) - seen in the ISA); one barrier.
    constexpr int U = DEPTH % 3 == 0 ? DEPTH : 3 * DEPTH;
    int it = 0;
    for (; it + U + DEPTH + 2 <= niter; it += U) {     // every prefetch issued here is one the loop will consume
#pragma unroll
        for (int d = 0; d < U; ++d) {
            const int slot = d % DEPTH, buf = d % 3, nbuf = (d + 1) % 3, wbuf = (d + 2) % 3;
            store_a(wbuf, st[slot].a);
            mma_chunk_full(buf, nbuf, st[slot].w);
            load_w(it + d + DEPTH, st[slot]);
            load_a(it + d + DEPTH + 2, st[slot].a);
            if (!(QL_W8A8_ABLATE & 16)) __syncthreads();
        }
    }
    QL_STAMP(3);                                       // main loop done
    // tail: prefetches only while there is something left to fetch; chunks may be partial / absent
#pragma unroll
    for (int d = 0; d < U + DEPTH + 1; ++d) {
        const int i = it + d;
        if (i < niter) {                               // block-uniform
            const int slot = d % DEPTH, buf = d % 3, nbuf = (d + 1) % 3, wbuf = (d + 2) % 3;   // `it` is a multiple of U
            if (i + 2 < niter) store_a(wbuf, st[slot].a);
            mma_chunk(i, buf, nbuf, st[slot].w);
            if (i + DEPTH < niter) load_w(i + DEPTH, st[slot]);
            if (i + DEPTH + 2 < niter) load_a(i + DEPTH + 2, st[slot].a);
            __syncthreads();
        }
    }

    }
    }   // register-staged forms
    QL_STAMP(4);                                       // K loop done
    // epilogue operands requested NOW: their global round trip overlaps the exchange below instead of sitting in the
    // wave's tail (measured on the ablation builds: the epilogue was 2.1 us of a 19 us kernel)
    const int nw = n0 + (COLG ? grp * 128 : 0) + wv * 32;     // first column of this wave
    const int n = nw + j;
    const float ws = Act<T>::load(Sc + (n < N ? n : N - 1));
    constexpr int OWN = KP == 1 ? MT : (MT == 1 ? 1 : MT / 2);            // row tiles this group finishes
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

    // ---- combine the two K-parity groups through LDS (exact: int32), then the rank-1 scale epilogue ----------
    // row tile mt is finished by group (mt & 1) (MT == 1: group 0): each group hands the OTHER group's tiles over
    // as 16-byte pieces [tile slot][piece q][lane] - consecutive lanes, consecutive 16 bytes: conflict-free
    if constexpr (KP == 2) {
        constexpr int SLOTS = (MT + 1) / 2;                                    // tiles a group receives
        i32x4* xch = reinterpret_cast<i32x4*>(smem);
    #pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int owner = owner_of(mt);
            if (owner != grp) {
                const int slot = mt >> 1;
    #pragma unroll
                for (int q = 0; q < 4; ++q)
                    xch[(((owner * 4 + wv) * SLOTS + slot) * 4 + q) * 64 + lane] =
                        i32x4{acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]};
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
                    const i32x4 o = xch[(((grp * 4 + wv) * SLOTS + slot) * 4 + q) * 64 + lane];
    #pragma unroll
                    for (int e = 0; e < 4; ++e) acc[mt][4 * q + e] += o[e];
                }
            }
        }
        __syncthreads();                                                       // LDS is reused for the output tiles below
    }
    QL_STAMP(5);                                       // K-parity groups combined
    if constexpr (SK == 2) {
        unsigned* flag = splitk_ws + kSplitkMaxTiles;                          // fixed layout: one workspace serves every shape in turn
        constexpr int kPieces = OWN * 4;                                       // 16-byte pieces per lane
        i32x4* mine = reinterpret_cast<i32x4*>(splitk_ws + 2 * kSplitkMaxTiles) + ((size_t)tile_id * (NG * 4) + (grp * 4 + wv)) * (kPieces * 64) + lane;
        const unsigned ticket = s_ticket;                                      // written before the K loop's first barrier
        // The hand-off itself in the guide's "drained sc1" form (MI355X_MICROARCH.md, hand-off price list, R1): 16-byte WRITE-THROUGH stores
        // (sc0 sc1: the bytes leave the XCD's L2), every storing wave drains its own stores (vmcnt(0)), block barrier, ONE relaxed
        // agent-scope flag store; the reader polls with ONE lane and fetches the pieces with sc0 sc1 loads (L1 bypassed) - no release /
        // acquire fence anywhere.  (First version: plain stores + an agent-scope release fence in all 512 threads = `buffer_wbl2 sc1` x 2048
        // per launch: the publishers spent 20.6 us behind their K loop, profiles/r06_w8a8_config3_splitk.txt.)
        if ((ticket & 1u) == 0) {                                              // publisher
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {                                  // compile-time register indices: no `2 o + grp`
                if (!owns(mt)) continue;
                const int o = own_slot(mt);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const i32x4 v = {acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]};
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(mine + (o * 4 + q) * 64), "v"(v) : "memory");
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag + tile_id, ticket + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            QL_STAMP(6);
            return;
        }
        if (tid == 0) {                                                        // finisher: the partner took ticket - 1, so it is running
            unsigned spins = 0;
            while (__hip_atomic_load(flag + tile_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ticket) {
#ifdef QL_SPLITK_DEBUG
                if (++spins > (1u << 18)) { __hip_atomic_store(flag + tile_id, 0xDEAD0000u | (ticket & 0xFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
#else
                if (++spins > (1u << 24)) __builtin_trap();                    // seconds: a lost partner must be loud, not a hang
#endif
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        {
            i32x4 other[OWN * 4];
#pragma unroll
            for (int i = 0; i < OWN * 4; ++i)
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(other[i]) : "v"(mine + i * 64) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (!owns(mt)) continue;
                const int o = own_slot(mt);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[mt][4 * q + e] += other[o * 4 + q][e];
            }
        }
        __syncthreads();                                                       // the epilogue reuses LDS; everyone past the loads
    }

    if (QL_W8A8_ABLATE & 64) {                         // no epilogue: one store keeps the accumulators alive
        int x = 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) x ^= acc[mt][i];
        if (x == 0x12345678) C[0] = (T)1.f;
        return;
    }
    const bool wide = (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    T* lds_wave = reinterpret_cast<T*>(smem) + (grp * 4 + wv) * 1024;     // 2 KB per wave (16 KB <= the A buffers)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (!owns(mt)) continue;
        const int o = own_slot(mt);
        if constexpr (sizeof(T) == 2) {
            if (wide) {
                store_tile_32x32<T>(lds_wave, C, ldc, m0 + mt * 32, nw, M, N, bias, lane,
                                    [&](int i) { return (float)acc[mt][i] * (asc[o][i] * ws); });
                continue;
            }
        }
        if (n < N) {
            const T* bn = bias ? bias + n : nullptr;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
                if (m < M) store_out<T>(C + (int64_t)m * ldc + n, (float)acc[mt][i] * (asc[o][i] * ws), bn);
            }
        }
    }
    QL_STAMP(6);                                       // output tiles stored (this wave)
}

template <typename T, int MT, int S, int DEPTH, int NG, bool COLG = false, int NITER = 0, bool ADMA = false>
static int launch_w8a8_tiled_mt(const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* Sc, const void* bias, void* C,
                                int64_t M, int64_t N, int64_t K, int64_t ldc, hipStream_t st) {
    constexpr int BM = 32 * MT;
    constexpr int BN = COLG ? 128 * NG : 128;
    const int nbx = (int)((N + BN - 1) / BN), nby = (int)((M + BM - 1) / BM);
    constexpr int kTiles = (COLG ? 1 : NG) * (ADMA ? 4 : 3) * BM * 128 * S;     // K-parity groups x three (LDS-DMA form: four) A buffers
    constexpr int kLds = kTiles < NG * 8192 ? NG * 8192 : kTiles;              // epilogue: 2 KB per wave
    static bool attr_set = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_tiled_kernel<T, MT, S, DEPTH, NG, COLG, NITER, ADMA>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kLds) == hipSuccess;
    }();
    (void)attr_set;
    const int rotate = QL_TUNE("QLINEAR_W8A8_ROTATE", 0);       // K-phase rotation of the column tiles: 17.5 vs 17.2 us in the unrolled form (LABNOTES r3)
    const bool no_super = QL_TUNE("QLINEAR_GEMM_SUPER", 1) == 0;   // super-tile raster (ql_common.h): profiles/r02_w8a8_l2_pmc.txt
    const int sy = NG == 1 ? 8 : 4;                    // blocks in flight per XCD: 64 (two per CU) or 32
    const bool super = !no_super && nbx % 8 == 0 && nby % sy == 0 && nby >= 2 * sy;
    w8a8_tiled_kernel<T, MT, S, DEPTH, NG, COLG, NITER, ADMA><<<(unsigned)(nbx * nby), NG * 256, kLds, st>>>(
        Aq, Wm, (int)M, (int)N, (int)K, super ? nbx : xcd_order(nbx, nby, (double)M * K, (double)N * K), rotate, super ? sy : 0,
        a_scale, (const T*)Sc,
        (const T*)bias, (T*)C, ldc);
    return finish_launch(QL_K_W8A8_TILED);
}

#ifdef QL_DEV_EXPERIMENTS     // recorded experiment (measured slower: include/qlinear_hip_dev.h), developer library only
// 128 x 128 tiles x 2 grid-level K slices (SK = 2 above): M % 128 == 0, N % 128 == 0, K % 512 == 0 (two slices x two parity groups x
// NITER chunks of 128 bytes).  ws: w8a8_splitk_workspace_bytes(), zeroed once by its owner.
size_t w8a8_splitk_workspace_bytes(int64_t M, int64_t N) {
    const size_t tiles = (size_t)((M + 127) / 128) * (size_t)((N + 127) / 128);
    return (size_t)(2 * kSplitkMaxTiles) * 4 + tiles * (size_t)(128 * 128 * 4);
}
bool w8a8_splitk_serves(int64_t M, int64_t N, int64_t K) {
    return K == 4096 && M % 128 == 0 && N % 128 == 0 && M >= 128 && (M / 128) * (N / 128) <= kSplitkMaxTiles && (M / 128) * (N / 128) * 2 <= 2 * (int64_t)cu_count() &&
           (M / 128) * (N / 128) * 2 >= (int64_t)cu_count() / 2 && M * K < ((int64_t)1 << 31);
}
template <typename T>
static int launch_w8a8_tiled_sk2(const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* Sc, const void* bias, void* C,
                                 int64_t M, int64_t N, int64_t K, int64_t ldc, void* ws, hipStream_t st) {
    constexpr int MT = 4, S = 1, NITER = 8;
    const int nbx = (int)(N / 128), nby = (int)(M / 128);
    constexpr int kLds = 2 * 4 * (32 * MT) * 128 * S;                          // two K-parity groups x four LDS-DMA buffers
    static bool attr_set = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_tiled_kernel<T, MT, S, 2, 2, false, NITER, true, 2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kLds) == hipSuccess;
    }();
    (void)attr_set;
    w8a8_tiled_kernel<T, MT, S, 2, 2, false, NITER, true, 2><<<(unsigned)(nbx * nby * 2), 512, kLds, st>>>(
        Aq, Wm, (int)M, (int)N, (int)K, nbx, 0, 0, a_scale, (const T*)Sc, (const T*)bias, (T*)C, ldc, (unsigned*)ws);
    return finish_launch(QL_K_W8A8_TILED);
}
int w8a8_gemm_tiled_splitk(int dtype, const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* S, const void* bias, void* C,
                           int64_t M, int64_t N, int64_t K, int64_t ldc, void* ws, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F32: return launch_w8a8_tiled_sk2<float>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, ws, st);
    case QL_DTYPE_F16: return launch_w8a8_tiled_sk2<f16>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, ws, st);
    case QL_DTYPE_BF16: return launch_w8a8_tiled_sk2<__bf16>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, ws, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

#endif

template <typename T>
static int launch_w8a8_tiled(const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* S, const void* bias, void* C,
                             int64_t M, int64_t N, int64_t K, int64_t ldc, hipStream_t st) {
    const int forced_mt = QL_TUNE("QLINEAR_W8A8_MT", 0), forced_ng = QL_TUNE("QLINEAR_W8A8_NG", 0);   // tuning sweeps (developer build)
    const int64_t nb = (N + 127) / 128;
    // tallest row tile that still gives every CU a block
    int mt = 1;
    for (int t = 4; t > 1; t >>= 1)
        if (M > 16 * t && nb * ((M + 32 * t - 1) / (32 * t)) >= 256) { mt = t; break; }
    if (mt == 1 && M > 32) mt = 2;
    if (forced_mt == 1 || forced_mt == 2 || forced_mt == 4) mt = forced_mt;
    // two independent 4-wave blocks per CU once there are at least two 128-row tiles per CU
    int ng = (mt == 4 && nb * ((M + 127) / 128) >= 512) ? 1 : 2;
    if (forced_ng == 1 || forced_ng == 2) ng = forced_ng;
#ifdef QL_DEV_TUNING     // column-group variant (two 128-column groups per block): measured behind the K-parity groups, LABNOTES r2
    if (mt == 4 && QL_TUNE("QLINEAR_W8A8_COLG", 0) == 1) return launch_w8a8_tiled_mt<T, 4, 1, 2, 2, true>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
#endif
    if (mt == 4 && ng == 1) return launch_w8a8_tiled_mt<T, 4, 1, 2, 1>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
    if (mt == 4) return launch_w8a8_tiled_mt<T, 4, 1, 2, 2>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
    // config 3's shape: 8 iterations of 256 bytes per K-parity group fully unrolled, A by LDS-DMA, hand-counted VM queue
    // (18.7 -> 17.1 us, profiles/r03_w8a8_kernel_stats.csv; developer build: QLINEAR_W8A8_UNROLL=0 / QLINEAR_W8A8_ADMA=0 for the A/B)
    const bool unroll = QL_TUNE("QLINEAR_W8A8_UNROLL", 1) != 0, adma = QL_TUNE("QLINEAR_W8A8_ADMA", 1) != 0;
    if (mt == 2 && K == 4096 && unroll && adma && M * K < (int64_t)1 << 31)
        return launch_w8a8_tiled_mt<T, 2, 2, 2, 2, false, 8, true>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
#ifdef QL_DEV_TUNING
    if (mt == 2 && K == 4096 && unroll)                // the register-staged unrolled form
        return launch_w8a8_tiled_mt<T, 2, 2, 2, 2, false, 8>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
#endif
    if (mt == 2) return launch_w8a8_tiled_mt<T, 2, 2, 2, 2>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
    return launch_w8a8_tiled_mt<T, 1, 2, 3, 2>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
}

int w8a8_gemm_tiled(int dtype, const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* S, const void* bias, void* C,
                    int64_t M, int64_t N, int64_t K, int64_t ldc, hipStream_t st) {
    if (w8a8_gemm256_supported(dtype, M, N, K, Aq)) {  // prefill row counts: 256 x 256 tiles (w8a8_gemm256.hip)
        // a last round of blocks filled to 40 % or less (qkv_proj at 8192 rows: 576 blocks = 2.25 rounds) is peeled off: whole
        // rounds on the 256-tile kernel, the remaining rows as a second launch on the 128-row tiles below (the weight-only GEMM's
        // rule, w4_gemm.hip: w4_gemm256_rows)
        const int64_t nbx = (N + 255) / 256, nby = (M + 255) / 256, blocks = nbx * nby, cus = cu_count();
        const int64_t full = blocks / cus, tail = blocks - full * cus, m_main = (full * cus / nbx) * 256;
        const size_t esz = dtype == QL_DTYPE_F32 ? 4 : 2;
        if (!(dispatch_flags() & QL_D_NOPEEL) && full >= 1 && tail > 0 && tail * 10 <= cus * 4 && m_main >= 256 && m_main < M) {
            const int rc = w8a8_gemm256(dtype, Aq, a_scale, Wm, S, bias, C, m_main, N, K, ldc, st);
            if (rc != 0) return rc;
            return w8a8_gemm_tiled(dtype, Aq + m_main * K, a_scale + m_main, Wm, S, bias, (char*)C + (size_t)(m_main * ldc) * esz, M - m_main, N, K,
                                   ldc, st);
        }
        return w8a8_gemm256(dtype, Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
    }
    switch (dtype) {
    case QL_DTYPE_F32: return launch_w8a8_tiled<float>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
    case QL_DTYPE_F16: return launch_w8a8_tiled<f16>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
    case QL_DTYPE_BF16: return launch_w8a8_tiled<__bf16>(Aq, a_scale, Wm, S, bias, C, M, N, K, ldc, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

}  // namespace ql

#ifdef QL_W8A8_STAMPS
extern "C" int qlinear_w8a8_stamps_read(unsigned long long* out, int blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ql::ql_w8a8_stamps), sizeof(unsigned long long) * ql::kStampPoints * blocks);
}
#endif
