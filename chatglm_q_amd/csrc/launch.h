// Internal launcher interface between abi.hip and the kernel translation units.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include "tune.h"
#include <stddef.h>
#include <stdint.h>

namespace ql {

// Every kernel launch funnels through this: bumps the launch counter, notes WHICH kernel family ran (QL_K_*,
// include/qlinear_hip.h: qlinear_last_dispatch) and turns the launch status into the ABI's return convention
// (0 OK, >0 hipError_t).
int finish_launch(int kernel_family = 1 /* QL_K_OTHER */);
int cu_count();   // of the current device (cached per device, abi.hip)

// activation prologue / gate epilogue of the one-row fused GEMVs (w4_packed.hip, w8_kernels.hip)
enum { PRO_NONE = 0, PRO_SILU = 1, PRO_ADDNORM = 2, PRO_NORM = 3 };   // PRO_NORM: ADDNORM without delta / hout (internal)
struct Prologue {
    const void* delta;      // PRO_ADDNORM: residual contribution to add first (nullable)
    const void* ln_weight;  // PRO_ADDNORM
    void* hout;             // PRO_ADDNORM: updated residual stream
    float eps;
    int gate_epilogue;      // 1: columns come in quads (h0, h1, gate0, gate1); C gets N / 2 columns
                            //    out[2t + i] = round(round(silu(y_i)) * y_{i+2}), y = rounded sum (+ bias)   model.py:200-201
};

// Derived int4g32 layout = two copies of the weights in one buffer (qlinear_w4g32_packed_bytes):
//   part 1 "column-major" (the GEMVs): Wt[n][g] 16-byte units, then Sp[n/4][g][n%4] scales (w4_packed.hip)
//   part 2 "tile-major"  (the MFMA kernels): Wm[ct][kt][lane], ct = n / 32, kt = 64-deep K step, lane = 32 kb + j
//          holding the unit of column 32 ct + j, group 2 kt + kb - exactly what lane (j, kb) of a wave feeds its MFMA
//          sub-steps from, so a wave's load instruction is 1 KB contiguous and consecutive steps are consecutive
//          (the per-column layout gave those kernels 32 bytes per 128-byte line per step, scattered over 32 DRAM
//          rows: a timing probe with this addressing ran w_in at 8 rows in 21 instead of 27 us); then Sm likewise.
//          Columns past N and the missing half of an odd last step hold q = 0 with scale 0.
struct W4Layout {
    int64_t G, Npad, ctiles, ksteps;
    size_t off_sp, off_wm, off_sm, bytes;
};
inline W4Layout w4_layout(int64_t N, int64_t K, size_t esize) {
    W4Layout L;
    L.G = K / 32;
    L.Npad = (N + 3) & ~(int64_t)3;
    L.ctiles = (N + 31) / 32;
    L.ksteps = (L.G + 1) / 2;
    L.off_sp = (size_t)(L.Npad * L.G) * 16;
    L.off_wm = (L.off_sp + (size_t)(L.Npad * L.G) * esize + 15) & ~(size_t)15;
    L.off_sm = L.off_wm + (size_t)(L.ctiles * L.ksteps) * 1024;
    L.bytes = L.off_sm + (size_t)(L.ctiles * L.ksteps) * 64 * esize;
    return L;
}
// The two parts may also live in two allocations (qlinear_w4g32_repack_gemv / qlinear_w4g32_tile): part 1 is then a
// buffer of off_wm bytes, part 2 ("tiled") one of bytes - off_wm, its scales at off_sm - off_wm.  The MFMA launchers
// take the address of part 2, the GEMV launchers the address of part 1.

constexpr int64_t kCanonMChunk = 64;   // rows per pass of the canonical split-K path (bounds the workspace)

// w4_kernels.hip
int w4_generic(int dtype, const void* A, const uint8_t* Wq, const void* S, const void* bias, void* C, int64_t M,
               int64_t N, int64_t K, int64_t group, int64_t lda, int64_t ldc, hipStream_t st);
size_t w4_canon_workspace_bytes(int64_t M, int64_t N, int64_t K);
int w4_canon(int dtype, const void* A, const uint8_t* Wq, const void* S, const void* bias, void* C, void* ws,
             int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc, hipStream_t st);
int w4_repack(int dtype, const uint8_t* Wq, const void* S, void* packed, int64_t N, int64_t K, hipStream_t st);   // both parts, one buffer
int w4_repack_gemv(int dtype, const uint8_t* Wq, const void* S, void* gemv, int64_t N, int64_t K, hipStream_t st);     // part 1 only
int w4_unpack_gemv(int dtype, const void* gemv, uint8_t* Wq, void* S, int64_t N, int64_t K, hipStream_t st);           // part 1 -> canonical buffers
int w4_tile(int dtype, const void* gemv, void* tiled, int64_t N, int64_t K, hipStream_t st);                          // part 2 from part 1
int w4_tiled(int dtype, const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
             int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st);   // any row count on part 2 (fp16 / bf16)
int w4_packed(int dtype, const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N,
              int64_t K, int64_t lda, int64_t ldc, bool strict, void* ws, size_t ws_bytes, hipStream_t st);

int w4_packed_residual(int dtype, bool strict, const void* A, const void* packed, const void* bias, const void* resid, void* C, int64_t N,
                       int64_t K, hipStream_t st);
int w4_packed_fused(int dtype, int kind, bool gate_epilogue, bool strict, const void* A, const void* packed, const void* bias, void* C,
                    int64_t N, int64_t K, const void* delta, const void* ln_weight, void* hout, float eps, hipStream_t st);

// w4_gemm.hip (M > 4, fp16 / bf16, MFMA)
int w4_packed_gemm(int dtype, const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N,
                   int64_t K, int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st);   // tiled: part 2
size_t w4_packed_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
bool w4_rows_use_gemm(int64_t M, int64_t N, int64_t K);
size_t w4_packed_workspace_bytes(int64_t M, int64_t N, int64_t K);   // of whichever kernel w4_packed() picks
// w4_fewrow.hip (rows <= 32, fp16 / bf16): independent K-slice waves, MFMA
bool w4_fewrow_supported(int64_t M, int64_t N, int64_t K);
size_t w4_fewrow_workspace_bytes(int64_t M, int64_t N, int64_t K);
int w4_fewrow(int dtype, const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
              int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st, bool gate = false);   // tiled: part 2

// w4_packed.hip: the two MLP projections of a one-row decode step in one launch (experiment); ws: w4_mlp_pair_workspace_bytes() zeroed once
size_t w4_mlp_pair_workspace_bytes();
int w4_mlp_pair(int dtype, const void* x, const void* ln_weight, float eps, const void* packed_a, const void* bias_a, int64_t Na,
                int64_t Ka, const void* packed_b, const void* bias_b, int64_t Nb, int64_t Kb, const void* resid, void* mid,
                void* out, void* ws, hipStream_t st);

// w4_engine.hip: the MLP of a one-row decode step as ONE persistent launch (LDS-DMA loader wave + consumer waves per CU, granule
// hand-off of the row between the projections); bit-equal to w4_packed_fused(PRO_NORM, gate) + w4_packed_residual
int w4_gemv_ksplit(int64_t quads, int64_t G);                 // w4_packed.hip: K slices per quad of the one-row GEMV for a shape
bool w4_mlp_engine_supported(int64_t Na, int64_t Ka, int64_t Nb, int64_t Kb);
size_t w4_mlp_engine_workspace_bytes(int64_t N_in);
int w4_mlp_engine(int dtype, bool strict, const void* x, const void* ln_weight, float eps, const void* packed_a, const void* bias_a,
                  int64_t Na, int64_t Ka, const void* packed_b, const void* bias_b, int64_t Nb, int64_t Kb, void* out, void* ws,
                  void* trace, hipStream_t st);

// w4_rows4.hip (1..4 rows, fp16 / bf16, exact-dequant arithmetic): 4x4x4 MFMA on part 1 of the derived layout; ks = K slices per quad
bool w4_rows4_supported(int dtype, int64_t M, int64_t N, int64_t K, int64_t lda);
bool w4_rows4_serves(int dtype, int64_t M, int64_t N, int64_t K, int64_t lda, bool strict);   // w4_packed.hip: the routing rule
// w4_rows16.hip (3..16 rows, fp16 / bf16, reference rounding): 16x16x32 MFMA on part 1, K split over the waves of a workgroup, one launch
bool w4_rows16_serves(int dtype, int64_t M, int64_t N, int64_t K, int64_t lda);
int w4_rows16(int dtype, const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
              int64_t ldc, hipStream_t st, bool gate = false);
int w4_rows4(int dtype, int ks, const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
             int64_t lda, int64_t ldc, hipStream_t st, bool gate = false, const void* delta = nullptr, const void* ln_weight = nullptr,
             void* hout = nullptr, float eps = 0.f);
int w4_rows4_fused(int dtype, bool gate, const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                   const void* delta, const void* ln_weight, void* hout, float eps, hipStream_t st);   // w4_packed.hip: picks the K split
int w4_rows4_gated(int dtype, const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                   int64_t lda, int64_t ldc, hipStream_t st);                                   // w4_packed.hip: picks the K split

// wq_gemm_f32.hip (round 5): fp32 activations, 128+ rows, on v_mfma_f32_32x32x2_f32 from the canonical buffers (int4g32 / int8 per channel)
bool wq_gemm_f32_serves(int64_t M, int64_t N, int64_t K);
int w4_gemm_f32(const void* A, const uint8_t* Wq, const void* S, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                int64_t ldc, hipStream_t st);
int w8_gemm_f32(const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t ldw,
                int64_t lda, int64_t ldc, hipStream_t st);

// w4_tgemm.hip (backward: grad_A = grad_out . dequant(W)^T on the canonical layout; fp16 / bf16, MFMA)
int w4_tgemm(int dtype, const void* A, const uint8_t* Wq, const void* S, void* C, int64_t M, int64_t Nout, int64_t Kc,
             int64_t lda, int64_t ldw, int64_t lds, int64_t ldc, hipStream_t st);

// w8_kernels.hip
int w8_generic(int dtype, const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t M,
               int64_t N, int64_t K, int64_t ldw_k, int64_t ldw_n, int64_t lda, int64_t ldc, hipStream_t st);
// one decode row, fp16, K % 16 == 0: add + RMSNorm prologue, optional SiLU * gate epilogue (rows of W in (h, h, gate, gate) quads)
int w8_gemv_residual(int dtype, const void* A, const int8_t* W, const void* S, const void* bias, const void* resid, void* C,
                     int64_t N, int64_t K, int64_t ldw, hipStream_t st);
int w8_gemv_fused(int dtype, bool gate_epilogue, const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t N,
                  int64_t K, int64_t ldw, const void* delta, const void* ln_weight, void* hout, float eps, hipStream_t st);
int w8_gemv(int dtype, const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t M,
            int64_t N, int64_t K, int64_t ldw, int64_t lda, int64_t ldc, bool strict, hipStream_t st);
// MFMA GEMM launch plan (both weight formats).  mt: tile height in units of 32 rows - the tallest tile that
// still leaves at least two blocks per CU.  ksplit: when even 32-row tiles give fewer than one block per CU
// (few rows: the launch is bound by the weight stream and needs memory-level parallelism, not MFMA rate), K is
// split over blockIdx.z into fp32 slabs in the workspace and a second launch sums them.  `per` = K steps
// (64 k each) per slab.  No or too small a workspace degrades to ksplit = 1.
// Developer build: QLINEAR_GEMM_MT / QLINEAR_GEMM_KSPLIT override (tune.h).
struct GemmPlan {
    int mt, ksplit, per;
};
inline GemmPlan gemm_plan(int64_t M, int64_t N, int64_t ksteps, size_t ws_bytes) {
    const int forced_mt = QL_TUNE("QLINEAR_GEMM_MT", 0), forced_ks = QL_TUNE("QLINEAR_GEMM_KSPLIT", 0);
    const int64_t nb = (N + 127) / 128;
    int mt = 1;
    if (forced_mt == 1 || forced_mt == 2 || forced_mt == 4 || forced_mt == 8) mt = forced_mt;
    else
        for (int t = 4; t > 1; t >>= 1)
            if (M > 16 * t && nb * ((M + 32 * t - 1) / (32 * t)) >= 512) { mt = t; break; }
    const int64_t blocks = nb * ((M + 32 * mt - 1) / (32 * mt));
    int64_t ks = 1;
    if (forced_ks > 0) ks = forced_ks;
    else if (blocks < 256) ks = (256 + blocks - 1) / blocks;   // measured optimum: about one block per CU
    if (ks > 16) ks = 16;
    if (ks > ksteps / 4) ks = ksteps / 4;                       // at least 4 K steps per slab
    while (ks > 1 && (size_t)(ks * M * N) * sizeof(float) > ws_bytes) --ks;
    if (ks < 1) ks = 1;
    const int64_t per = (ksteps + ks - 1) / ks;
    ks = (ksteps + per - 1) / per;                               // every slab non-empty
    return {mt, (int)ks, (int)per};
}
inline size_t gemm_workspace_bytes(int64_t M, int64_t N, int64_t ksteps) {
    const GemmPlan p = gemm_plan(M, N, ksteps, (size_t)-1);
    return p.ksplit > 1 ? (size_t)(p.ksplit * M * N) * sizeof(float) : 0;
}

// w4_gemm256.hip: many rows (prefill) on 256 x 256 tiles, weights dequantised once per block into LDS, A by LDS-DMA
bool w4_gemm256_supported(int64_t M, int64_t N, int64_t K, int64_t lda, const void* A, size_t esize);
bool w4_gemm256_can_run(int64_t M, int64_t N, int64_t K, int64_t lda, const void* A, size_t esize);
int64_t w4_gemm256_rows(int64_t M, int64_t N, int64_t K, int64_t lda, const void* A, size_t esize);   // w4_gemm.hip: 0 / peel / M
bool w4_gemm256_tail_in_kernel(int64_t blocks, int64_t K);   // the persistent launch runs its left-over tiles as half tiles: no peel
int w4_gemm256(int dtype, const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
               int64_t ldc, hipStream_t st);
int w4_gemm256_residual(int dtype, const void* A, const void* tiled, const void* bias, const void* resid, void* C, int64_t M, int64_t N,
                        int64_t K, int64_t lda, int64_t ldc, int64_t ldr, hipStream_t st);   // C = round(round(y) + resid)
int w8_gemm256_gated(int dtype, const void* A, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                     int64_t lda, int64_t ldc, hipStream_t st);
int w8_gemm256_residual(int dtype, const void* A, const int8_t* Wm, const void* S, const void* bias, const void* resid, void* C, int64_t M,
                        int64_t N, int64_t K, int64_t lda, int64_t ldc, int64_t ldr, hipStream_t st);
int w4_gemm256_gated(int dtype, const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                     int64_t ldc, hipStream_t st);   // gate-interleaved copy, SiLU * gate epilogue: C (M, N / 2)
int w8_gemm256(int dtype, const void* A, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
               int64_t lda, int64_t ldc, hipStream_t st);   // int8 per-channel weights (tile-major copy) through the same kernel
// w4_dense256.hip: dequantise the call's weights once into a 16-bit fragment-major image, then a dense 16-bit ring GEMM on it
size_t dense256_image_bytes(int64_t N, int64_t K);
bool dense256_can_run(int64_t M, int64_t N, int64_t K, int64_t lda, const void* A);
int dense256_expand(int dtype, bool w8, const void* tiled, const void* S, void* image, int64_t N, int64_t K, hipStream_t st);
int dense256(int dtype, bool gate, const void* A, const void* image, const void* bias, const void* resid, void* C, int64_t M, int64_t N,
             int64_t K, int64_t lda, int64_t ldc, int64_t ldr, hipStream_t st);
// w8a8_gemm256.hip: int8 activations x tile-major int8 weights, many rows, 256 x 256 tiles, both operands by LDS-DMA
bool w8a8_gemm256_can_run(int dtype, int64_t M, int64_t N, int64_t K, const void* Aq);
bool w8a8_gemm256_supported(int dtype, int64_t M, int64_t N, int64_t K, const void* Aq);
int w8a8_gemm256(int dtype, const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M,
                 int64_t N, int64_t K, int64_t ldc, hipStream_t st);
int w8a8_gemm256_gated(int dtype, const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M,
                       int64_t N, int64_t K, int64_t ldc, hipStream_t st);   // gate-interleaved copy, SiLU * gate epilogue: C (M, N / 2)
// w8_gemm.hip (M > 4, fp16 / bf16, MFMA)
int w8_gemm(int dtype, const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t M, int64_t N,
            int64_t K, int64_t ldw, int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st);
size_t w8_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
// tile-major derived copy of int8 weights (16-byte aligned rows, K % 16 == 0) and the MFMA kernels on it
size_t w8_tiled_bytes(int64_t N, int64_t K);
int w8_tile(const int8_t* W, int8_t* Wm, int64_t N, int64_t K, int64_t ldw, hipStream_t st);
size_t w8_tiled_workspace_bytes(int64_t M, int64_t N, int64_t K);
int w8_fwd_tiled(int dtype, const void* A, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M, int64_t N,
                 int64_t K, int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st);   // few-row kernel or tiled GEMM
int w8_gemm_tiled(int dtype, const void* A, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M, int64_t N,
                  int64_t K, int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st);
int w8_gemm_scale_k(int dtype, const void* A, const int8_t* W, const void* S, void* C, int64_t M, int64_t N, int64_t K,
                    int64_t ldw, int64_t lda, int64_t ldc, hipStream_t st);
// w8a8.hip
int act_quant_rowwise(int dtype, const void* A, int8_t* Aq, float* a_scale, int64_t M, int64_t K, int64_t lda, bool per_tensor,
                      hipStream_t st);
int w8a8_gemm_tiled(int dtype, const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* S, const void* bias, void* C,
                    int64_t M, int64_t N, int64_t K, int64_t ldc, hipStream_t st);
// 128 x 128 tiles x 2 grid-level K slices (config 3's shape class); ws zeroed once by its owner
size_t w8a8_splitk_workspace_bytes(int64_t M, int64_t N);
bool w8a8_splitk_serves(int64_t M, int64_t N, int64_t K);
int w8a8_gemm_tiled_splitk(int dtype, const int8_t* Aq, const float* a_scale, const int8_t* Wm, const void* S, const void* bias, void* C,
                           int64_t M, int64_t N, int64_t K, int64_t ldc, void* ws, hipStream_t st);
int w8a8_gemm(int dtype, const int8_t* Aq, const float* a_scale, const int8_t* W, const void* S, const void* bias,
              void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st);
size_t w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K);

// w4a8.hip: int4g32 weights x int8 activations (i8 MFMA, one group per MFMA, fp32 fold); fp16 / bf16
size_t w4a8_packed_bytes(int64_t N, int64_t K, int dtype);
int w4a8_pack(int dtype, const uint8_t* Wq, const void* S, void* out, int64_t N, int64_t K, hipStream_t st);
int w4a8_gemm(int dtype, const int8_t* Aq, const float* a_scale, const void* packed, const void* bias, void* C, int64_t M,
              int64_t N, int64_t K, int64_t ldc, hipStream_t st);

// embed_kernels.hip
int qembedding_w4(int dtype, const int64_t* ids, const uint8_t* Wq, const void* S, void* out, int64_t count,
                  int64_t V, int64_t D, int64_t group, hipStream_t st);
int qembedding_w8(int dtype, const int64_t* ids, const int8_t* W, const void* S, void* out, int64_t count,
                  int64_t V, int64_t D, hipStream_t st);

// decode_ops.hip
int rmsnorm(int dtype, const void* X, const void* Delta, const void* W, void* Hout, void* Out, int64_t rows, int64_t dim,
            int64_t ldx, int64_t ldo, float eps, hipStream_t st);
int rope_kv_write(int dtype, const void* QKV, const void* table, const int64_t* pos, const int64_t* widx, void* Qout,
                  void* Kc, void* Vc, int64_t B, int64_t S, int64_t H, int64_t G, int64_t D, int64_t capacity,
                  int64_t ldqkv, hipStream_t st);
int decode_attention(int dtype, const void* Q, const void* Kc, const void* Vc, const float* mask, void* Out, int64_t B,
                     int64_t H, int64_t G, int64_t D, int64_t capacity, hipStream_t st);
// Bytes that block g of the NEXT launch on the stream will read first: [base[r] + g * block_bytes[r], + block_bytes[r])
// for r = 0, 1 (block_bytes[r] == 0: no such region), g < blocks.  The grouped attention kernel is a chain of round
// trips on B * G workgroups; its launch carries extra workgroups that pull these bytes into L2 / the memory-side cache
// meanwhile, placed on the XCD of the workgroup that will read them (workgroups go to XCDs round robin).
struct Prefetch {
    const char* base[2];
    int64_t block_bytes[2];
    int blocks;
};
int decode_attention_rope(int dtype, const void* QKV, const void* table, const int64_t* pos, const int64_t* widx, void* Kc,
                          void* Vc, const float* mask, void* Out, int64_t B, int64_t H, int64_t G, int64_t D,
                          int64_t capacity, int64_t ldqkv, float* split_ws, const Prefetch& pf, hipStream_t st);   // split_ws: nullable
// how the one-row GEMVs walk their weights (for Prefetch): bytes per workgroup of the weight and the scale stream
void w4_gemv_blocks(int64_t N, int64_t K, int64_t* w_block_bytes, int64_t* s_block_bytes, int64_t* s_offset, int64_t* blocks);
void w8_gemv_blocks(int64_t N, int64_t K, int64_t ldw, int64_t* w_block_bytes, int64_t* blocks);
size_t decode_attention_split_bytes(int64_t B, int64_t H, int64_t D, int64_t capacity);
int silu_mul(int dtype, const void* In, void* Out, int64_t rows, int64_t hidden, int64_t ldin, int64_t ldo, hipStream_t st);
// quantising producers (decode_ops.hip): the same rows, also (or only: Out == nullptr) as int8 + one fp32 scale per row
int rmsnorm_quant(int dtype, const void* X, const void* Delta, const void* W, void* Hout, void* Out, int8_t* Aq, float* a_scale,
                  int64_t rows, int64_t dim, int64_t ldx, int64_t ldo, float eps, hipStream_t st);
int silu_mul_quant(int dtype, const void* In, void* Out, int8_t* Aq, float* a_scale, int64_t rows, int64_t hidden, int64_t ldin,
                   int64_t ldo, hipStream_t st);
int masked_softmax(int dtype, const void* Sc, const float* mask, void* P, int64_t rows, int64_t Tn, int64_t mask_rows, int64_t lds,
                   int64_t ldm, int64_t ldp, hipStream_t st);
int greedy_advance(int dtype, const void* logits, int64_t B, int64_t N, int64_t ldl, int64_t* tok, int64_t* write_index,
                   int64_t* pos, float* mask, int64_t capacity, hipStream_t st);

// sampler.hip: top-k / top-p sampling + the step's bookkeeping in one launch (chatglm_q/decoder.py:12-27)
int top_p_sample(int dtype, const void* logits, int64_t B, int64_t N, int64_t ldl, int64_t top_k, float top_p, float temperature,
                 const float* dparams, uint64_t* rng_state, int64_t* tok, int64_t* write_index, int64_t* pos, float* mask,
                 int64_t capacity, float* probs_out, int64_t* index_out, float* u_out, int64_t out_ld, hipStream_t st);

// prefill_attention.hip: many-position attention in one launch (D = 128, H = 16 G, fp16 / bf16)
int prefill_attention(int dtype, const void* Q, const void* Kc, const void* Vc, const float* mask, const uint8_t* flags, void* Out,
                      int64_t B, int64_t S, int64_t Tkv, int64_t H, int64_t G, int64_t cap, int64_t ldm, hipStream_t st);
void prefill_attention_tiles(int64_t* q_block, int64_t* k_tile);

}  // namespace ql
