/* qlinear_hip.h - C ABI of libqlinear_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the quantized-linear forward path of K024/chatglm-q.  Each entry point
 * replaces one host->device launch site of the reference (paths relative to the reference
 * checkout):
 *
 *   qlinear_w4g32_fwd / _fwd_packed  <- dynamic_quant_matmul_s4      chatglm_q/int4/triton_ops.py:90-139
 *                                       (kernel _dynamic_quant_matmul_s4_kernel, :18-87)
 *   qlinear_w8_fwd                   <- dynamic_quant_matmul         chatglm_q/int8/triton_ops.py:87-127
 *                                       (kernel _dynamic_quant_matmul_kernel, :13-84)
 *   qlinear_act_quant_i8_rowwise,
 *   qlinear_w8a8_fwd                 <- the int8-activation semantic that exists only in
 *                                       DynamicQuantizeMatMul.symbolic chatglm_q/int8/qlinear.py:56-70
 *                                       and quantize_int8           chatglm_q/int8/quantizer.py:11-19
 *   qlinear_qembedding_*             <- QEmbedding.forward           chatglm_q/int4/qlinear.py:122-131,
 *                                                                    chatglm_q/int8/qlinear.py:118-120
 *
 * Conventions
 *   - plain C types only; all pointers are DEVICE pointers unless stated; the library never
 *     allocates, frees or retains them; the caller owns outputs and workspaces.
 *   - every launch goes to the hipStream_t passed in (as void*); no hidden synchronisation,
 *     no use of the default stream, no mutable global state apart from an idempotent
 *     per-process device-property cache.  Thread-safe.
 *   - return value: 0 = OK; < 0 = argument error (QL_ERR_*); > 0 = hipError_t of the launch.
 *     Nothing is thrown across the ABI, nothing aborts.
 *   - dtype: activation / scale / bias / output element type (QL_DTYPE_*); scales, bias and the
 *     output always have the activation dtype, as the reference asserts
 *     (chatglm_q/int4/triton_ops.py:108).
 *   - rounding: every dequantised weight is rounded to the activation dtype before it is
 *     multiplied, products accumulate in fp32, the sum is rounded once to the activation dtype,
 *     and the optional bias is added after that rounding (a second rounding) - the reference's
 *     exact sequence (chatglm_q/int4/triton_ops.py:72-80, chatglm_q/int4/qlinear.py:92-93).
 */
#ifndef QLINEAR_HIP_H
#define QLINEAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): qlinear_w4g32_fwd_packed_residual gained `flags` in front of `stream` (round 3); qlinear_last_dispatch /
 * qlinear_dispatch_reset / qlinear_dispatch_reload / qlinear_gemm256_serves / qlinear_tiled_dispatch added; the experimental entry points (one-launch MLP pair / persistent MLP
 * engine / W4A8) moved to include/qlinear_hip_dev.h and libqlinear_hip_dev.so. */
/* 3 (round 6): qlinear_top_p_sample added; round 5's additions (qlinear_w4g32_unpack_gemv, qlinear_dispatch_flags,
 * qlinear_w4g32_packed_dispatch, QL_K code 19 = ROWS16, qlinear_gated_serves weight_bits 88 = int8 x int8) had gone out under 2. */
#define QLINEAR_ABI_VERSION 3

/* activation dtypes */
#define QL_DTYPE_F32 0
#define QL_DTYPE_F16 1
#define QL_DTYPE_BF16 2

/* status codes */
#define QL_OK 0
#define QL_ERR_NULL_POINTER (-1)
#define QL_ERR_BAD_SHAPE (-2)      /* M/N/K <= 0, K odd, K % group != 0, ld* too small */
#define QL_ERR_BAD_DTYPE (-3)
#define QL_ERR_BAD_GROUP (-4)      /* group size unsupported by this entry point */
#define QL_ERR_MISALIGNED (-5)     /* pointer / leading dimension breaks the alignment contract */
#define QL_ERR_WORKSPACE (-6)      /* workspace missing or too small (see qlinear_workspace_bytes) */
#define QL_ERR_UNSUPPORTED (-7)

/* flags (bit set) accepted by the forward entry points that take a `flags` argument */
#define QL_FLAG_STRICT_ROUNDING 1  /* round every dequantised weight to the activation dtype before the
                                      multiply, bit-for-bit the reference's sequence.  Default (0) feeds
                                      the exactly dequantised weight to the dot product instead, which is
                                      closer to real arithmetic and differs from the reference only by that
                                      per-weight rounding (~2e-4 relative in fp16; within the 1e-3 bar). */

#define QL_FLAG_ACT_PER_TENSOR 2    /* activation quantisation: ONE scale max|A| / 127 for the whole (M, K) tensor - the per-tensor
                                      symmetric variant of DynamicQuantizeMatMul.symbolic's second branch
                                      (chatglm_q/int8/qlinear.py:64-70: ReduceMax(Abs(A)) / 127, QuantizeLinear with zero
                                      point 0) - instead of one scale per row (quantize_int8, chatglm_q/int8/quantizer.py:
                                      11-19).  a_scale still has M entries (all equal), so the GEMM epilogue is unchanged. */

/* operations, for qlinear_workspace_bytes */
#define QL_OP_W4G32_FWD 1
#define QL_OP_W4G32_FWD_PACKED 2
#define QL_OP_W8_FWD 3
#define QL_OP_W8A8_FWD 4
#define QL_OP_W8_FWD_TILED 5
#define QL_OP_W8A8_LINEAR_TILED 6  /* REQUIRED by qlinear_w8a8_linear_tiled: the int8 activations and their scales */
#define QL_OP_W4A8_LINEAR 7        /* REQUIRED by qlinear_w4a8_linear: the same */

int qlinear_abi_version(void);
const char* qlinear_status_string(int status);

/* Number of kernel launches this process has issued through the library (monotonic, relaxed).
 * Lets a caller prove that a result came from the HIP path and not from any fallback. */
uint64_t qlinear_launch_count(void);

/* Debug query: WHICH kernel families the calling thread's library calls launched since qlinear_dispatch_reset(): one byte per
 * launch, newest in the low byte, up to 8 launches (older ones fall off).  tests/test_dispatch_sweep_gpu.py asserts with it that
 * the kernel really changes where the dispatch tables say it does.  Thread-local, no synchronisation, no effect on results. */
#define QL_K_OTHER 1            /* repacks, norms, attention, embeddings, quantisers, ... */
#define QL_K_W4_GEMV 2          /* w4_packed.hip: 1..2 rows (4 per pass), part 1 */
#define QL_K_W4_ROWS4 3         /* w4_rows4.hip: 2..4 rows on v_mfma_f32_4x4x4, part 1 */
#define QL_K_W4_FEWROW 4        /* w4_fewrow.hip: 3..32 rows not served by QL_K_W4_ROWS16, part 2 */
#define QL_K_W4_GEMM128 5       /* w4_gemm.hip: 32..128-row tiles, part 2 (+ QL_K_SPLITK_REDUCE for few rows) */
#define QL_K_W4_GEMM256 6       /* w4_gemm256.hip: 256 x 256 tiles, part 2 */
#define QL_K_W4_CANON 7         /* w4_kernels.hip: canonical layout, split-K */
#define QL_K_W4_GENERIC 8       /* w4_kernels.hip: any group size / alignment */
#define QL_K_SPLITK_REDUCE 9
#define QL_K_W8_GEMV 10         /* w8_kernels.hip: up to 4 rows per pass on the (N, K) buffer */
#define QL_K_W8_FEWROW 11       /* w8_gemm.hip: 3..32 rows, tile-major copy */
#define QL_K_W8_GEMM128 12      /* w8_gemm.hip: 32..128-row tiles (row-major or tile-major weights) */
#define QL_K_W8_GEMM256 13      /* w4_gemm256.hip<W8>: 256 x 256 tiles, tile-major copy */
#define QL_K_W8_GENERIC 14
#define QL_K_W8A8_ROWMAJOR 15   /* w8_kernels.hip: round 1's i8 x i8 kernel on the (N, K) buffer */
#define QL_K_W8A8_TILED 16      /* w8a8.hip: 64 / 128-row tiles, tile-major copy */
#define QL_K_W8A8_GEMM256 17    /* w8a8_gemm256.hip */
#define QL_K_ACT_QUANT 18
#define QL_K_W4_ROWS16 19       /* w4_rows16.hip: 3..16 rows on v_mfma_f32_16x16x32, part 1, one launch */
uint64_t qlinear_last_dispatch(void);
void qlinear_dispatch_reset(void);
void qlinear_dispatch_reload(void);   /* parse QLINEAR_DISPATCH again (it is read once, at the first dispatch decision) */
/* The parsed QLINEAR_DISPATCH switches as the library holds them (bit 0 no256, 1 nopeel, 2 nofewrow, 3 norows4, 4 nogroupattn,
 * 5 nohalf, 6 nof32mfma, 7 norows16): a host that sizes a workspace or picks an entry point by one of them asks here instead of parsing the variable again. */
unsigned qlinear_dispatch_flags(void);

/* Environment variables read by THIS library (chatglm_q_amd/csrc/tune.h): exactly one,
 *   QLINEAR_DISPATCH = comma list of kernel families the dispatch must not use: no256, nopeel, nofewrow, norows4, nogroupattn,
 *   nohalf (the half-tile last round inside the int4g32 256-tile GEMM's launch: whole tiles only, the older peel where it applies),
 *   nof32mfma (the fp32 matrix-instruction kernel that serves fp32 activations from 128 rows on: the VALU kernels instead),
 *   norows16 (the one-launch 3..16-row kernel on part 1: the few-row kernel on part 2 instead)
 * (every family has a slower fallback computing the same function; for A/B measurements and triage).  The host package reads
 * QLINEAR_LIB_PATH (another build of this library) and QLINEAR_STRICT (0 / 1 / auto: the per-weight rounding policy, see
 * QL_FLAG_STRICT_ROUNDING) - chatglm_q_amd/_lib.py.  Every other tuning value is a compile-time constant with its measurement
 * cited at its use; the developer build (make dev, -DQL_DEV_TUNING) turns them back into environment variables. */

/* Bytes of scratch the op wants for this shape (0 = none).  The caller allocates it (e.g. from
 * torch's caching allocator so stream semantics hold) and passes it to the op.  It is REQUIRED by
 * QL_OP_W4G32_FWD (QL_ERR_WORKSPACE otherwise) and OPTIONAL for QL_OP_W4G32_FWD_PACKED / QL_OP_W8_FWD[_TILED] / QL_OP_W8A8_FWD,
 * where it lets a few-row (5 <= M < ~256) MFMA GEMM split K over workgroups into fp32 slabs that a
 * second launch sums; without it (NULL / too small / not 16-byte aligned) those shapes run unsplit -
 * same rounding sequence, several times slower. */
size_t qlinear_workspace_bytes(int op, int64_t M, int64_t N, int64_t K, int64_t group);

/* ---- int4 group-quantised weights, canonical (reference) layout ---------------------------
 * A   (M, K) activations, row stride lda elements, unit inner stride
 * Wq  (K/2, N) uint8, contiguous; byte [k/2, n] = row 2*(k/2) low nibble, row 2*(k/2)+1 high
 *     nibble; stored nibble = q + 8 (chatglm_q/int4/quantizer.py:24-28)
 * S   (K/group, N) scales, contiguous, activation dtype
 * bias (N) or NULL;  C (M, N) row stride ldc
 * Any group size dividing K is accepted; group == 32 with N % 8 == 0 takes the fast kernels. */
int qlinear_w4g32_fwd(const void* A, const uint8_t* Wq, const void* S, const void* bias, void* C,
                      int64_t M, int64_t N, int64_t K, int64_t group, int64_t lda, int64_t ldc,
                      int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the int4g32 product w.r.t. the activations (the weights are frozen integers):
 *   dA (M, K) = Gout (M, N) . dequant(Wq, S)^T      canonical layout, group 32, fp16 / bf16, N % 16 == 0
 * Same arithmetic as the reference's transposed kernel (every weight rounded to the activation dtype, fp32
 * accumulation, one output rounding).  Replaces dynamic_quant_matmul_transposed_s4
 * (chatglm_q/int4/triton_ops.py:142-264) behind DynamicQuantizeMatMul.backward (chatglm_q/int4/qlinear.py:53-64);
 * unlike the reference's kernel it does not need power-of-two sizes (triton_ops.py:246). */
int qlinear_w4g32_bwd_input(const void* Gout, const uint8_t* Wq, const void* S, void* dA, int64_t M, int64_t N, int64_t K,
                            int64_t group, int64_t ldg, int64_t ldda, int dtype, void* stream);

/* ---- int4 g32, derived layout ---------------------------------------------------------------
 * A lazily built, non-persistent re-arrangement of the SAME weights, two copies in one buffer:
 *   part 1, column-major (rows <= 2, GEMV): the 32 nibbles of one (column, group) in one 16-byte unit,
 *           units K-contiguous per column, scales regrouped per 4 columns - one wave owns whole output
 *           columns and no cross-workgroup reduction is needed;
 *   part 2, tile-major (rows >= 3, MFMA kernels): the same units ordered [column / 32][64-deep K step][lane],
 *           i.e. in the order the lanes of a wave feed their MFMA fragments (1 KB contiguous per load).
 * Built once per weight by qlinear_w4g32_repack from the canonical buffers, which remain the source of truth
 * (state_dict).  group must be 32.  Exact offsets: chatglm_q_amd/csrc/launch.h (W4Layout). */
size_t qlinear_w4g32_packed_bytes(int64_t N, int64_t K, int64_t group, int dtype);
int qlinear_w4g32_repack(const uint8_t* Wq, const void* S, void* packed, int64_t N, int64_t K,
                         int64_t group, int dtype, void* stream);
int qlinear_w4g32_fwd_packed(const void* A, const void* packed, const void* bias, void* C,
                             int64_t M, int64_t N, int64_t K, int64_t group, int64_t lda,
                             int64_t ldc, int dtype, int flags, void* workspace,
                             size_t workspace_bytes, void* stream);

/* The two parts as two allocations, so that a weight which never sees three or more rows (decode-only sessions, fp32)
 * never pays for part 2: part 1 ("gemv", qlinear_w4g32_gemv_bytes) is what qlinear_w4g32_fwd_packed reads for M <= 2
 * and every one-row fused / residual entry point reads; part 2 ("tiled", qlinear_w4g32_tiled_bytes; 0 for fp32, which
 * has no MFMA path) is built FROM part 1 by qlinear_w4g32_tile and serves any row count through
 * qlinear_w4g32_fwd_tiled (the few-row and MFMA GEMM kernels; workspace: qlinear_workspace_bytes(QL_OP_W4G32_FWD_PACKED)).
 * gemv_bytes + tiled_bytes == packed_bytes, and a buffer of qlinear_w4g32_repack is exactly gemv followed by tiled.
 * The host module (chatglm_q_amd/int4/qlinear.py) builds part 2 on the first forward with >= 3 rows. */
/* 1 when qlinear_w4g32_fwd_packed serves this call from part 2 (a caller that keeps the parts apart then calls
 * qlinear_w4g32_fwd_tiled), 0 when part 1 does: one or two rows (GEMV), 2..4 rows in the default arithmetic on the 4x4x4
 * matrix instruction while the staged rows stay within 64 KB (w4_rows4.hip), 3..16 rows of the narrow layer shapes on the
 * 16x16x32 matrix instruction in one launch (w4_rows16.hip: the rule is at rows16_cfg there), fp32 at any row count. */
int qlinear_w4g32_rows_on_tiled(int64_t M, int64_t N, int64_t K, int dtype, int flags);
/* The QL_K_* family that serves such a call from part 1 (QL_K_W4_GEMV / QL_K_W4_ROWS4 / QL_K_W4_ROWS16), 0 when part 2 serves it. */
int qlinear_w4g32_packed_dispatch(int64_t M, int64_t N, int64_t K, int dtype, int flags);
/* 1 when qlinear_w4g32_fwd_tiled_gated (weight_bits 4) / qlinear_w8_fwd_tiled_gated (weight_bits 8) / qlinear_w8a8_fwd_tiled_gated
 * (weight_bits 88: int8 activations x int8 weights, the ring kernel's own rule) serves M rows of a first MLP
 * projection (N = 2 * hidden outputs) with SiLU * gate in its epilogue, 0 when it would return QL_ERR_UNSUPPORTED for the shape:
 * lets a host ask BEFORE it builds the gate-interleaved tile-major copy (~120 MB per ChatGLM2-6B layer). */
int qlinear_gated_serves(int64_t M, int64_t N, int64_t K, int dtype, int weight_bits);
size_t qlinear_w4g32_gemv_bytes(int64_t N, int64_t K, int64_t group, int dtype);
size_t qlinear_w4g32_tiled_bytes(int64_t N, int64_t K, int64_t group, int dtype);
int qlinear_w4g32_repack_gemv(const uint8_t* Wq, const void* S, void* gemv, int64_t N, int64_t K, int64_t group, int dtype,
                              void* stream);
/* The inverse of qlinear_w4g32_repack_gemv (round 5): part 1 -> the canonical buffers Wq (K / 2, N) uint8 and S (K / 32, N), byte for
 * byte.  A host that keeps only derived layouts resident (chatglm_q_amd: DynamicQuantizeLinear.drop_canonical) serves state_dict()
 * and checkpoints - the reference's buffer contract, chatglm_q/loader.py:90-104 - through it. */
int qlinear_w4g32_unpack_gemv(const void* gemv, uint8_t* Wq, void* S, int64_t N, int64_t K, int64_t group, int dtype, void* stream);
int qlinear_w4g32_tile(const void* gemv, void* tiled, int64_t N, int64_t K, int64_t group, int dtype, void* stream);
int qlinear_w4g32_fwd_tiled(const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                            int64_t group, int64_t lda, int64_t ldc, int dtype, void* workspace, size_t workspace_bytes,
                            void* stream);
/* The many-row kernel of qlinear_w4g32_fwd_tiled alone (round 3, w4_gemm256.hip: 256 x 256 output tiles, the weights of a K tile
 * dequantised once per block into LDS, A by LDS-DMA; same arithmetic: every weight (n - 8) * s rounded to the activation dtype,
 * chatglm_q/int4/triton_ops.py:72-73, fp32 accumulation).  qlinear_w4g32_fwd_tiled picks it by itself when the grid fills the
 * chip in whole rounds (prefill row counts); this entry runs it for any M.  QL_ERR_UNSUPPORTED unless K % 64 == 0, K >= 128,
 * 16-byte aligned rows of A and M * lda * 2 < 2^31.  fp16 / bf16. */
int qlinear_w4g32_fwd_tiled256(const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                               int64_t group, int64_t lda, int64_t ldc, int dtype, void* stream);
/* 1 when the dispatch of qlinear_w4g32_fwd_tiled / qlinear_w8_fwd_tiled (and the _gated / _residual entry points, which exist on
 * that kernel only) takes the 256 x 256-tile kernel for M contiguous, 16-byte aligned rows of K 16-bit activations: K % 64 == 0,
 * K >= 128 and a grid that fills the chip in whole rounds (at least one block per CU, last round >= 70 % full).  Lets a caller
 * decide BEFORE it builds a gate-interleaved tile-major copy whether the gated entry point will serve the row count. */
int qlinear_gemm256_serves(int64_t M, int64_t N, int64_t K);
/* Host-only: what qlinear_w4g32_fwd_tiled (weight_bits = 4) / qlinear_w8_fwd_tiled (8) launches for M contiguous 16-bit rows:
 * returns the QL_K_* family of the first launch; *rows_first (nullable) = the rows it serves - fewer than M only for
 * QL_K_W4_GEMM256, when the rows of a thinly filled last round go to QL_K_W4_GEMM128 as a second launch ("peel").  The dispatch
 * table as a function, for tests and tools; 0 for bad arguments. */
int qlinear_tiled_dispatch(int weight_bits, int64_t M, int64_t N, int64_t K, int64_t* rows_first);

/* One-row (decode) forward on the derived layout with an activation PROLOGUE fused into the staging of the
 * activation row, so that the small op in front of the QLinear call costs no launch of its own (SURVEY.md 8f
 * N1).  fp16 / bf16, group 32; exact-dequant arithmetic unless QL_FUSED_STRICT is OR-ed into `prologue` (then every
 * dequantised weight is rounded to the activation dtype first, as QL_FLAG_STRICT_ROUNDING does for qlinear_w4g32_fwd_packed:
 * what the host module asks for with bf16 activations).  Rounding sequence as the model graph's:
 *   QL_PRO_SILU     A is (h | gate), 2K values: row = round(round(silu(h)) * gate)            chatglm_q/model.py:200-201
 *   QL_PRO_ADDNORM  hnew = round(A + delta) (delta nullable), written to hout (nullable);
 *                   row = round(round(hnew * rsqrt(mean(hnew^2) + eps)) * ln_weight)          chatglm_q/model.py:62-73,243-245
 * and optionally the gated-activation EPILOGUE of the MLP's first projection (OR the flag into `prologue`):
 *   QL_EPI_SILU_GATE  the N packed columns come in quads (h_2t, h_2t+1, gate_2t, gate_2t+1) - i.e. the module's
 *                   (K, 2H) weight was column-permuted before qlinear_w4g32_repack (bias likewise) - and C
 *                   receives N / 2 values: C[2t + i] = round(round(silu(y_i)) * y_{i+2}), y = round(sum) (+ bias,
 *                   rounded): the (.., 2H) intermediate of chatglm_q/model.py:200-201 never exists. */
#define QL_PRO_SILU 1
#define QL_PRO_ADDNORM 2
#define QL_EPI_SILU_GATE 0x100
#define QL_FUSED_STRICT 0x200
int qlinear_w4g32_fwd_packed_fused(int prologue, const void* A, const void* packed, const void* bias, void* C,
                                   int64_t N, int64_t K, const void* delta, const void* ln_weight, void* hout,
                                   float eps, int dtype, void* stream);

/* ---- int8 per-output-channel weights -------------------------------------------------------
 * W  int8, logical (K, N) with element strides (ldw_k, ldw_n) - exactly what the reference
 *    wrapper receives: the module passes weight.t() of its (N, K) row-major buffer, i.e.
 *    (ldw_k, ldw_n) = (1, K) (chatglm_q/int8/qlinear.py:90); the reference test passes a
 *    contiguous (K, N), i.e. (N, 1) (tests/test_triton_ops.py:11).
 * S  (N) scales, activation dtype. */
int qlinear_w8_fwd(const void* A, const int8_t* W, const void* S, const void* bias, void* C,
                   int64_t M, int64_t N, int64_t K, int64_t ldw_k, int64_t ldw_n, int64_t lda,
                   int64_t ldc, int dtype, int flags, void* workspace, size_t workspace_bytes,
                   void* stream);

/* ---- int8 weights, derived tile-major copy (rows >= 3: MFMA kernels) ------------------------
 * tiled[N / 32][64-deep K step][half h][lane][16 bytes]: lane = 32 kb + j, half h holds bytes k = 64 kt + 32 kb + 16 h .. + 15
 * of output channel 32 (n / 32) + j, zero padded - the order in which the lanes of a wave feed their MFMA fragments: every
 * wave load instruction reads 1 KB contiguous (from the (N, K) buffer it is 16 bytes from each of 32 rows).  A lazily
 * built, non-persistent copy like the int4 one; the (N, K) buffer stays the source of truth.  fp16 / bf16,
 * K % 16 == 0.  qlinear_w8_fwd_tiled: few-row kernel (independent K-slice waves) up to 32 rows, tiled GEMM above;
 * workspace as for qlinear_w8_fwd (QL_OP_W8_FWD_TILED, optional). */
size_t qlinear_w8_tiled_bytes(int64_t N, int64_t K);
int qlinear_w8_tile(const int8_t* W, void* tiled, int64_t N, int64_t K, int64_t ldw_n, void* stream);
int qlinear_w8_fwd_tiled(const void* A, const void* tiled, const void* S, const void* bias, void* C, int64_t M, int64_t N,
                         int64_t K, int64_t lda, int64_t ldc, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* The many-row kernel of qlinear_w8_fwd_tiled alone (round 3: the 256 x 256-tile kernel of w4_gemm256.hip fed from the tile-major
 * int8 copy; every weight b * s rounded to the activation dtype, chatglm_q/int8/triton_ops.py:62-73, fp32 accumulation).  Picked
 * automatically at prefill row counts; this entry runs it for any M.  QL_ERR_UNSUPPORTED unless K % 64 == 0, K >= 128, 16-byte
 * aligned rows of A and M * lda * 2 < 2^31. */
int qlinear_w8_fwd_tiled256(const void* A, const void* tiled, const void* S, const void* bias, void* C, int64_t M, int64_t N,
                            int64_t K, int64_t lda, int64_t ldc, int dtype, void* stream);
/* The int4 kernel's prefill epilogues for int8 weight-only (round 3): `tiled` = qlinear_w8_tile of the weights (for _gated: of the
 * gate-interleaved row order (h_2t, h_2t+1, gate_2t, gate_2t+1), S and bias permuted alike; C is (M, N / 2), see
 * qlinear_w4g32_fwd_tiled_gated); _residual: C = round(round(y) + residual), see qlinear_w4g32_fwd_tiled_residual.  Bit-equal to
 * qlinear_w8_fwd_tiled followed by qlinear_silu_mul / an elementwise add.  QL_ERR_UNSUPPORTED unless the 256 x 256-tile kernel serves
 * the row count. */
int qlinear_w8_fwd_tiled_gated(const void* A, const void* tiled, const void* S, const void* bias, void* C, int64_t M, int64_t N,
                               int64_t K, int64_t lda, int64_t ldc, int dtype, void* stream);
int qlinear_w8_fwd_tiled_residual(const void* A, const void* tiled, const void* S, const void* bias, const void* residual, void* C,
                                  int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc, int64_t ldr, int dtype, void* stream);

/* Few rows (batched decode: 2..32) through a first MLP projection with the SiLU * gate EPILOGUE: `packed` holds the
 * gate-interleaved column order (h_2t, h_2t+1, gate_2t, gate_2t+1), C gets N / 2 columns,
 * C[m, 2t+i] = round(round(silu(y_i)) * y_{i+2}), y = rounded sum (+ bias, rounded) - chatglm_q/model.py:200-201.
 * QL_ERR_UNSUPPORTED when the shape is not served by the few-row kernel without K slabs (N % 32 != 0, narrow
 * matrices, other row counts): run qlinear_w4g32_fwd_packed + qlinear_silu_mul instead.  fp16 / bf16. */
int qlinear_w4g32_fwd_packed_gated(const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                                   int64_t lda, int64_t ldc, int dtype, void* stream);
/* Row counts for which qlinear_w4g32_rows_on_tiled is 0 (2..4 rows while the staged rows stay within 64 KB) are served from part 1
 * by the 4x4x4-MFMA kernel - `packed` may then be a part-1-only buffer (qlinear_w4g32_repack_gemv of the permuted weights).
 * The same on part 2 alone (qlinear_w4g32_tile of the gate-interleaved part 1): */
int qlinear_w4g32_fwd_tiled_gated(const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                                  int64_t lda, int64_t ldc, int dtype, void* stream);
/* Prefill row counts with the residual add of the block in the GEMM's epilogue (round 3): C = round(round(A . dequant(W) (+ bias)) +
 * residual) - `hidden = hidden + sublayer(...)`, chatglm_q/model.py:243,245, with the sublayer output rounded first as the reference
 * materialises it; bit-equal to qlinear_w4g32_fwd_tiled + an elementwise add.  residual (M, N) with row stride ldr; must not overlap C.
 * QL_ERR_UNSUPPORTED unless the 256 x 256-tile kernel serves the row count (callers then add the residual themselves). */
int qlinear_w4g32_fwd_tiled_residual(const void* A, const void* tiled, const void* bias, const void* residual, void* C, int64_t M,
                                     int64_t N, int64_t K, int64_t lda, int64_t ldc, int64_t ldr, int dtype, void* stream);

/* The same prologue (QL_PRO_ADDNORM, optionally | QL_EPI_SILU_GATE) for 2..4 contiguous rows (batched decode), on the 4x4x4-MFMA
 * kernel: A, delta (nullable), hout (nullable; written when delta is given) are (M, K), C is (M, N) or (M, N / 2) with the gate
 * epilogue; bit-equal to qlinear_add_rmsnorm (qlinear_rmsnorm without delta) followed by the projection (and qlinear_silu_mul).
 * QL_ERR_UNSUPPORTED outside the row counts / shapes for which qlinear_w4g32_rows_on_tiled is 0, or K > 8192. */
int qlinear_w4g32_fwd_rows_fused(int prologue, const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N,
                                 int64_t K, const void* delta, const void* ln_weight, void* hout, float eps, int dtype, void* stream);

/* One-row forward on the derived layout whose output is added to the residual stream in the EPILOGUE:
 * C[n] = round(y[n] + residual[n]), y = round(sum) (+ bias, rounded) - chatglm_q/model.py:243,245
 * (hidden = hidden + attention(...), hidden = hidden + ffn(...)).  The next projection's QL_PRO_ADDNORM prologue then
 * runs without delta / hout (one operand less to stage in every workgroup).  C may alias residual.  fp16 / bf16.
 * flags: 0 or QL_FLAG_STRICT_ROUNDING. */
int qlinear_w4g32_fwd_packed_residual(const void* A, const void* packed, const void* bias, const void* residual, void* C,
                                      int64_t N, int64_t K, int dtype, int flags, void* stream);

/* The int8 twin (fp16 activations, K-contiguous weight rows of stride ldw, per-channel scales S). */
int qlinear_w8_fwd_residual(const void* A, const int8_t* W, const void* S, const void* bias, const void* residual, void* C,
                            int64_t N, int64_t K, int64_t ldw, int dtype, void* stream);

/* One-row (decode) int8 forward with the add + RMSNorm PROLOGUE (QL_PRO_ADDNORM) and optionally the SiLU * gate
 * EPILOGUE (| QL_EPI_SILU_GATE: the N rows of W - and S, bias - come in (h_2t, h_2t+1, gate_2t, gate_2t+1) quads,
 * C receives N / 2 values): the int8 twin of qlinear_w4g32_fwd_packed_fused, same rounding sequence.
 * W (N, K) int8 row-major with row stride ldw_n (the module's buffer), fp16, K % 16 == 0, K <= 16384. */
int qlinear_w8_fwd_fused(int prologue, const void* A, const int8_t* W, const void* S, const void* bias, void* C, int64_t N,
                         int64_t K, int64_t ldw_n, const void* delta, const void* ln_weight, void* hout, float eps, int dtype,
                         void* stream);

/* Backward of the int8 product w.r.t. the activations:
 *   dA (M, K) = Gout (M, N) . (Wkn * S[None, :])^T     Wkn = the logical (K, N) matrix, CONTIGUOUS (row stride N),
 *                                                      i.e. module.weight.t().contiguous(); S (N); fp16 / bf16
 * every weight rounded to the activation dtype, fp32 accumulation, one output rounding - the arithmetic of
 * dynamic_quant_matmul_transposed (chatglm_q/int8/triton_ops.py:130-245) behind DynamicQuantizeMatMul.backward
 * (chatglm_q/int8/qlinear.py:41-52). */
int qlinear_w8_bwd_input(const void* Gout, const int8_t* Wkn, const void* S, void* dA, int64_t M, int64_t N, int64_t K,
                         int64_t ldg, int64_t ldda, int dtype, void* stream);

/* ---- int8 activations x int8 weights (true i8 x i8 -> i32 MFMA contraction) ---------------
 * Row-wise symmetric activation quantisation in fp32 arithmetic:
 *   a_scale[m] = max(max_k |A[m,k]| / 127, 1e-10);  Aq = clamp(rint(A / a_scale), -127, 127)
 * (quantize_int8, chatglm_q/int8/quantizer.py:11-19).  Aq is (M, K) contiguous. */
int qlinear_act_quant_i8_rowwise(const void* A, int8_t* Aq, float* a_scale, int64_t M, int64_t K,
                                 int64_t lda, int dtype, void* stream);
/* The same with `flags`: 0 = row-wise (identical to qlinear_act_quant_i8_rowwise), QL_FLAG_ACT_PER_TENSOR = one scale
 * for the whole tensor, a_scale[m] = max(max_{m,k} |A[m,k]| / 127, 1e-10) for every m (the clamp keeps the 0 / 0 of an
 * all-zero tensor out - the NaN the reference's comment at qlinear.py:65 mentions); rounding = round half to even,
 * saturated to [-127, 127] (ONNX QuantizeLinear; |x / scale| <= 127 by construction, so -128 never occurs). */
int qlinear_act_quant_i8(const void* A, int8_t* Aq, float* a_scale, int64_t M, int64_t K, int64_t lda, int dtype,
                         int flags, void* stream);
/* C[m,n] = round(acc_i32[m,n] * (a_scale[m] * w_scale[n])) (+ bias); W is (N, K) row-major. */
int qlinear_w8a8_fwd(const int8_t* Aq, const float* a_scale, const int8_t* W, const void* S, const void* bias,
                     void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, int dtype, void* workspace,
                     size_t workspace_bytes, void* stream);

/* The same contraction with the weights in the tile-major derived copy of qlinear_w8_tile (a lane's 16 bytes are its
 * MFMA operand: W bypasses LDS; two K-parity wave groups per block).  K % 16 == 0.  No workspace. */
int qlinear_w8a8_fwd_tiled(const int8_t* Aq, const float* a_scale, const void* tiled, const void* S, const void* bias,
                           void* C, int64_t M, int64_t N, int64_t K, int64_t ldc, int dtype, void* stream);
/* The many-row kernel of qlinear_w8a8_fwd_tiled alone (round 3, w8a8_gemm256.hip: 256 x 256 output tiles, both operands global ->
 * LDS by LDS-DMA - the tile-major weights land as the MFMA fragments they are -, v_mfma_i32_32x32x32_i8; integer stage exact, same
 * epilogue).  qlinear_w8a8_fwd_tiled picks it by itself at prefill row counts; this entry runs it for any M.  QL_ERR_UNSUPPORTED
 * unless fp16 / bf16, K % 128 == 0, K >= 256 and M * K < 2^31. */
int qlinear_w8a8_fwd_tiled256(const int8_t* Aq, const float* a_scale, const void* tiled, const void* S, const void* bias, void* C,
                              int64_t M, int64_t N, int64_t K, int64_t ldc, int dtype, void* stream);
/* int8 ACTIVATIONS x the gate-interleaved tile-major copy of a first MLP projection (round 4): the many-row ring kernel with SiLU * gate
 * (chatglm_q/model.py:200-201) in its epilogue, C (M, N / 2) with 8-byte aligned rows; bit-equal to qlinear_w8a8_fwd_tiled followed by
 * qlinear_silu_mul.  QL_ERR_UNSUPPORTED when the 256 x 256-tile kernel does not serve the row count (qlinear_gated_serves(.., 88)). */
int qlinear_w8a8_fwd_tiled_gated(const int8_t* Aq, const float* a_scale, const void* tiled, const void* S, const void* bias, void* C,
                                 int64_t M, int64_t N, int64_t K, int64_t ldc, int dtype, void* stream);
/* Both steps of the int8-activation linear in ONE call (two launches): quantise A (M, K) row-wise (or per tensor:
 * flags & QL_FLAG_ACT_PER_TENSOR) into the workspace, then the tile-major GEMM.  workspace: qlinear_workspace_bytes(
 * QL_OP_W8A8_LINEAR_TILED, M, N, K, 0) bytes, 16-byte aligned; it holds Aq (M * K bytes) followed by a_scale (M floats)
 * and may be inspected afterwards. */
int qlinear_w8a8_linear_tiled(const void* A, const void* tiled, const void* S, const void* bias, void* C, int64_t M,
                              int64_t N, int64_t K, int64_t lda, int64_t ldc, int dtype, int flags, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- quantised embedding gathers ("next" row N3) -------------------------------------------
 * ids: (count) int64 token ids.  int4: Wq (V/2, D) packs along the vocabulary axis, S (V/group, D).
 * int8: W (V, D), S (D).  out: (count, D) activation dtype. */
int qlinear_qembedding_w4(const int64_t* ids, const uint8_t* Wq, const void* S, void* out,
                          int64_t count, int64_t V, int64_t D, int64_t group, int dtype, void* stream);
int qlinear_qembedding_w8(const int64_t* ids, const int8_t* W, const void* S, void* out,
                          int64_t count, int64_t V, int64_t D, int dtype, void* stream);

/* ---- fused neighbours of the QLinear calls in one decode step ("next" row N1) ----------------------
 * Same rounding sequence as the model graph (chatglm_q/model.py): every intermediate the reference
 * materialises in the activation dtype is rounded to it.
 * rmsnorm:           out[r,:] = round(round(x[r,:] * rsqrt(mean(x^2) + eps)) * w)            model.py:62-73
 * add_rmsnorm:       h = round(x + delta) written to Hout (the block's residual add, model.py:243,245), then
 *                    out = rmsnorm(h): one launch for both
 * rope_kv_write:     qkv (B*S, (H+2G)*D) -> q (B*S, H*D) rotated; k (rotated), v written to the caches
 *                    (B, capacity, G, D) at row widx[s]; table (max_pos, D/2, 2) = (cos, sin); pos (B*S)
 *                                                                                              model.py:139-155
 * decode_attention:  one query position: q (B, H*D), mask (B, capacity) additive fp32 -> out (B, H*D)
 *                                                                                              model.py:157-175
 * decode_attention_rope: rope_kv_write (S = 1) and decode_attention in ONE launch: qkv (B, (H+2G)*D) in, the
 *                    cache row widx[0] written by one block per (b, group), out (B, H*D).  The reference appends the
 *                    step's key / value at the END of its cache (model.py:148-151); here the cache is preallocated, so
 *                    rows behind widx[0] hold nothing of the sequence: the caller masks them (as it must for the
 *                    per-head kernels too) and the 16-heads-per-group kernel does not compute on them whatever the
 *                    mask says; with widx[0] outside [0, capacity) nothing is written and the mask alone decides
 * silu_mul:          in (rows, 2*hidden) -> out[r,i] = round(round(silu(in[r,i])) * in[r,hidden+i])   model.py:200-201
 * Bounds of the rotary entry points (no table length crosses the ABI): `table` must hold capacity + 1 positions; pos
 * values are clamped to [0, capacity] - positions are 1-based counts (model.py:307-308), a valid one is at most its
 * cache row + 1 - and a widx outside [0, capacity) writes no cache row. */
int qlinear_rmsnorm(const void* X, const void* W, void* Out, int64_t rows, int64_t dim, int64_t ldx, int64_t ldo,
                    float eps, int dtype, void* stream);
int qlinear_add_rmsnorm(const void* X, const void* Delta, const void* W, void* Hout, void* Out, int64_t rows,
                        int64_t dim, int64_t ld, float eps, int dtype, void* stream);
int qlinear_rope_kv_write(const void* QKV, const void* table, const int64_t* pos, const int64_t* widx, void* Qout,
                          void* Kcache, void* Vcache, int64_t B, int64_t S, int64_t H, int64_t G, int64_t D,
                          int64_t capacity, int64_t ldqkv, int dtype, void* stream);
int qlinear_decode_attention(const void* Q, const void* Kcache, const void* Vcache, const float* mask, void* Out,
                             int64_t B, int64_t H, int64_t G, int64_t D, int64_t capacity, int dtype, void* stream);
/* split_workspace (nullable, qlinear_decode_attention_split_bytes): long contexts - 256-position windows over
 * blockIdx.y leave (max, exp-sum, unnormalised output) there and a second launch combines them; the probabilities are
 * then not rounded to the activation dtype before P.V (same function within fp tolerance).  Without it one block per
 * (sequence, head) walks the whole cache with the reference's rounding points (capacity <= ~15 k).
 * H == 16 G, D == 128, fp16 / bf16, ldqkv % 8 == 0: one block per (sequence, group, window) on the matrix cores
 * (a single launch up to capacity 256, split_workspace needed above; QLINEAR_DISPATCH=nogroupattn turns it off). */
size_t qlinear_decode_attention_split_bytes(int64_t B, int64_t H, int64_t D, int64_t capacity);
int qlinear_decode_attention_rope(const void* QKV, const void* table, const int64_t* pos, const int64_t* widx, void* Kcache,
                                  void* Vcache, const float* mask, void* Out, int64_t B, int64_t H, int64_t G, int64_t D,
                                  int64_t capacity, int64_t ldqkv, int dtype, void* split_workspace,
                                  size_t split_workspace_bytes, void* stream);
/* The same launch, carrying extra workgroups that read the weights of the NEXT one-row linear on the stream (the
 * attention output projection) into L2 / the memory-side cache while the attention itself - a chain of round trips on
 * B * G workgroups - leaves the chip and HBM idle.  next_weights: QL_NEXT_W4G32_PACKED = buffer of
 * qlinear_w4g32_repack, QL_NEXT_W8_ROWS = contiguous (N, K) int8 rows; NULL: no prefetch.  Only the group kernel
 * (H == 16 G, D == 128, fp16 / bf16) prefetches; results never depend on it. */
#define QL_NEXT_W4G32_PACKED 1
#define QL_NEXT_W8_ROWS 2
int qlinear_decode_attention_rope_prefetch(const void* QKV, const void* table, const int64_t* pos, const int64_t* widx,
                                           void* Kcache, void* Vcache, const float* mask, void* Out, int64_t B, int64_t H,
                                           int64_t G, int64_t D, int64_t capacity, int64_t ldqkv, int dtype,
                                           void* split_workspace, size_t split_workspace_bytes, const void* next_weights,
                                           int next_kind, int64_t next_N, int64_t next_K, void* stream);
/* masked_softmax: P[r, :] = round(softmax_fp32(scores[r, :] + mask[r % mask_rows, :])) - the add / fp32 softmax / cast
 * between the two GEMMs of the many-position attention (chatglm_q/model.py:166-170) in one pass; mask nullable. */
int qlinear_masked_softmax(const void* scores, const float* mask, void* P, int64_t rows, int64_t T, int64_t mask_rows,
                           int64_t lds, int64_t ldm, int64_t ldp, int dtype, void* stream);
/* prefill_attention (round 3): the many-position attention between qkv_proj and o_proj of a prefill chunk in ONE launch
 * (chatglm_q/model.py:157-175: q / sqrt(d), q k^T, + mask, fp32 softmax, cast, p v) instead of two batched GEMMs around
 * qlinear_masked_softmax - the (heads x S x T) score matrix is never written.  Q (B, S, H, D) rotated queries (qlinear_rope_kv_write),
 * Kcache / Vcache (B, capacity, G, D) of which rows [0, T) are attended, mask (B, S, T) additive fp32 with row stride ldm
 * (nullable = no mask), Out (B, S, H * D).  Applied exactly as the reference applies it: round(score) + mask in fp32, so a row
 * whose keys are all blocked comes out as the uniform average.  tile_flags (nullable; uint8 (B, ceil(S / q_block), ceil(T / k_tile)),
 * block sizes from qlinear_prefill_attention_tiles) lets the kernel skip key tiles and mask loads:
 *   0 = every mask entry of the tile <= -1e9 AND every query row of the block has an entry >= -1e6 somewhere (the tile's
 *       probabilities are exactly 0 in fp32 for any fp16-range score): skipped;  2 = every entry == 0: no mask loads;  1 = otherwise.
 * The caller derives them from the SAME mask (fused_ops.attention_tile_flags); wrong flags give wrong results, not faults.
 * QL_ERR_UNSUPPORTED unless D == 128, H == 16 G and fp16 / bf16 (ChatGLM2's geometry): callers keep the GEMM route then. */
int qlinear_prefill_attention_tiles(int64_t* q_block, int64_t* k_tile);
int qlinear_prefill_attention(const void* Q, const void* Kcache, const void* Vcache, const float* mask, const uint8_t* tile_flags,
                              void* Out, int64_t B, int64_t S, int64_t T, int64_t H, int64_t G, int64_t D, int64_t capacity,
                              int64_t ldm, int dtype, void* stream);
int qlinear_silu_mul(const void* In, void* Out, int64_t rows, int64_t hidden, int64_t ldin, int64_t ldo, int dtype,
                     void* stream);
/* Quantising producers (round 3): the row a norm / activation kernel holds in registers is ALSO emitted as int8 + one fp32 scale
 * per row - bit for bit what qlinear_act_quant_i8 (row-wise: quantize_int8, chatglm_q/int8/quantizer.py:11-19) makes of the
 * rounded 16-bit output row - so that the int8-activation GEMM behind it (qlinear_w8a8_fwd_tiled / qlinear_w8a8_fwd) needs no
 * quantiser launch ("pre-quantized activations", north_star).  Aq is (rows, dim) contiguous, a_scale (rows).  Out is nullable
 * (nobody else reads the 16-bit row); Delta / Hout as in qlinear_add_rmsnorm (both null: plain qlinear_rmsnorm).  fp16 / bf16.
 * Callers: chatglm_q/model.py:213,244 (RMSNorm in front of qkv_proj / w_in), :200-201 (SiLU * gate in front of w_out). */
int qlinear_rmsnorm_quant_i8(const void* X, const void* Delta, const void* W, void* Hout, void* Out, int8_t* Aq, float* a_scale,
                             int64_t rows, int64_t dim, int64_t ld, float eps, int dtype, void* stream);
int qlinear_silu_mul_quant_i8(const void* In, void* Out, int8_t* Aq, float* a_scale, int64_t rows, int64_t hidden, int64_t ldin,
                              int64_t ldo, int dtype, void* stream);
/* greedy decode bookkeeping in one launch (chatglm_q/decoder.py:85,97 with temperature -> 0): tok[b] = argmax of logits
 * row b (lowest index on ties), pos[b] += 1, write_index[0] += 1, mask[b][new write_index] = 0 for every row. */
int qlinear_greedy_advance(const void* logits, int64_t B, int64_t N, int64_t ldl, int64_t* tok, int64_t* write_index,
                           int64_t* pos, float* mask, int64_t capacity, int dtype, void* stream);
/* The reference's decoding mode in one launch behind lm_head (chatglm_q/decoder.py:12-27 `top_p_sampling`, called per token at :85):
 *   probs = softmax(float(logits) / temperature); sort descending, keep top_k (ties: lowest index first); zero every entry whose
 *   PRECEDING cumulative mass exceeds top_p; renormalise; one multinomial draw; tok[b] = its token index.
 * logits (B, N) with row stride ldl, one workgroup per row, any N, top_k <= 1024 (QL_ERR_UNSUPPORTED above, unless N <= 1024).
 * dev_params (nullable, device float[3] = top_k, top_p, temperature) overrides the three scalars at run time, so that ONE captured
 * graph serves every sampler setting.  rng_state (device uint64[1 + B], nullable = seed 0 / counter 0): [0] the seed, [1 + b] row
 * b's draw counter, advanced by one per launch - the draw is u = (Philox4x32-10(counter = (ctr, b, 0), key = seed).x >> 8) / 2^24
 * and the token is the first sorted entry whose inclusive cumulative kept mass exceeds u * (kept mass): replays of a graph draw
 * fresh numbers, equal (seed, counter) reproduce the token.  top_k = 1 is qlinear_greedy_advance's argmax bit for bit.
 * write_index / pos / mask (all nullable together): the decode step's bookkeeping exactly as qlinear_greedy_advance does it.
 * probs_out (B, out_ld) / index_out (B, out_ld) (nullable together): the filtered, renormalised distribution in sorted order and its
 * token indices (the first min(top_k, N) entries of a row are written) - what the parity tests compare with the reference;
 * u_out (B, nullable): the uniform number each row drew. */
int qlinear_top_p_sample(const void* logits, int64_t B, int64_t N, int64_t ldl, int64_t top_k, float top_p, float temperature,
                         const float* dev_params, uint64_t* rng_state, int64_t* tok, int64_t* write_index, int64_t* pos, float* mask,
                         int64_t capacity, float* probs_out, int64_t* index_out, float* u_out, int64_t out_ld, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QLINEAR_HIP_H */
