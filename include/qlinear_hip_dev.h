/* qlinear_hip_dev.h - entry points that exist in libqlinear_hip_dev.so ONLY (make -C chatglm_q_amd/csrc dev).
 *
 * Recorded experiments: each was built to the conventions of qlinear_hip.h, measured on MI355X, lost to the product path, and is
 * kept (with its parity test, marker `dev`) so that the measurement can be repeated - not shipped.  LABNOTES.md has the numbers:
 *   qlinear_w4g32_mlp_pair    round 2: both MLP projections of a one-row decode step in one launch, chained grids   27.5 vs 23.0 us
 *   qlinear_w4g32_mlp_engine  round 3: the same as ONE persistent launch on an LDS-DMA ring                         32.0 vs 23.0 us
 *   qlinear_w4a8_*            round 2 / 3: int4g32 weights x int8 activations on the i8 matrix cores: the per-group fp32 fold
 *                             caps it at half the i8 rate; behind the f16-MFMA weight-only GEMM at every shape
 * The developer library also contains everything qlinear_hip.h declares, compiled with -DQL_DEV_TUNING: every tuning constant
 * of the launchers is an environment variable there (chatglm_q_amd/csrc/tune.h).
 */
#ifndef QLINEAR_HIP_DEV_H
#define QLINEAR_HIP_DEV_H

#include "qlinear_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* EXPERIMENT (one launch instead of two): the MLP of a one-row decode step,
 *   Out = round(w_out(round(silu(h) * gate)) + residual),  (h | gate) = w_in(rmsnorm(X) * ln_weight),
 * i.e. qlinear_w4g32_fwd_packed_fused(QL_PRO_ADDNORM | QL_EPI_SILU_GATE) followed by qlinear_w4g32_fwd_packed_residual, bit
 * for bit, with the second projection's workgroups in the SAME launch behind the first one's: they request their weights
 * at once and wait (bounded) for the (1, N_in / 2) row `mid` through arrival counters in `workspace`
 * (qlinear_w4g32_mlp_pair_workspace_bytes() bytes, 64-byte aligned, zeroed ONCE by the caller; the kernel resets it).
 * packed_in: gate-interleaved part 1 of the (K, N_in) projection; packed_out: part 1 of the (N_in / 2, N_out) projection.
 * QL_ERR_UNSUPPORTED for shapes other than the ChatGLM2-6B layer's kernel configuration: use the two calls. */
size_t qlinear_w4g32_mlp_pair_workspace_bytes(void);
int qlinear_w4g32_mlp_pair(const void* X, const void* ln_weight, float eps, const void* packed_in, const void* bias_in, int64_t N_in,
                           const void* packed_out, const void* bias_out, int64_t N_out, int64_t K, const void* residual, void* mid,
                           void* Out, void* workspace, int dtype, void* stream);

/* The MLP of a one-row decode step as ONE PERSISTENT launch (w4_engine.hip):
 *   Out = round(w_out(round(silu(h) * gate)) + X),  (h | gate) = w_in(rmsnorm(X) * ln_weight)       chatglm_q/model.py:199-201,244-245
 * bit for bit what qlinear_w4g32_fwd_packed_fused(QL_PRO_ADDNORM | QL_EPI_SILU_GATE) followed by
 * qlinear_w4g32_fwd_packed_residual(residual = X) compute (flags: 0 or QL_FLAG_STRICT_ROUNDING, as there).  One workgroup per CU:
 * a loader wave streams both projections' packed weights through an LDS ring with LDS-DMA and never waits for an activation,
 * seven consumer waves compute the two-launch kernels' per-wave sums out of LDS, the (1, N_in / 2) row between the projections
 * travels as 8-byte {data, tag} granules in `workspace`.
 * workspace: qlinear_w4g32_mlp_engine_workspace_bytes(N_in) bytes, 64-byte aligned, zeroed ONCE by the caller and then owned by
 *   launches of ONE stream (the launch epoch lives in it; word 2 is an error code: non-zero = a bounded wait gave up and the
 *   results of that launch are garbage).  X and Out must not alias.  fp16 / bf16, group 32.
 * packed_in: gate-interleaved part 1 of the (K, N_in) projection; packed_out: part 1 of the (N_in / 2, N_out) one; N_out == K.
 * QL_ERR_UNSUPPORTED when qlinear_w4g32_mlp_engine_supported is 0 (K slice of a task longer than 128 groups, LDS): two launches. */
size_t qlinear_w4g32_mlp_engine_workspace_bytes(int64_t N_in);
int qlinear_w4g32_mlp_engine_supported(int64_t N_in, int64_t K, int64_t N_out);
int qlinear_w4g32_mlp_engine(const void* X, const void* ln_weight, float eps, const void* packed_in, const void* bias_in, int64_t N_in,
                             const void* packed_out, const void* bias_out, int64_t N_out, int64_t K, void* Out, void* workspace, int dtype,
                             int flags, void* stream);

/* ---- int4 g32 weights x int8-quantised activations (W4A8: SURVEY.md 8d config 5, BASELINE configs[4]) ------------
 * C[m,n] = round(a_scale[m] * sum_g s[g,n] * (sum_{k in group g} Aq[m,k] * (nibble[k,n] - 8))) (+ bias):
 * the activation side of the int8 path (row-wise / per-tensor symmetric quantisation, chatglm_q/int8/quantizer.py:11-19,
 * chatglm_q/int8/qlinear.py:60-70) with the int4 weight decode (chatglm_q/int4/triton_ops.py:71-73).  One 32-deep
 * v_mfma_i32_32x32x32_i8 = one group; its exact int32 result is scaled into an fp32 accumulator.  The integer stage is
 * exact; the result differs from the weight-only (W4A16) path by the activation quantisation error (~1e-2 relative) -
 * an opt-in accuracy / throughput trade, not a parity claim against the Triton reference.
 * packed_a8: a third derived layout of the canonical buffers (qlinear_w4a8_pack; group 32, K % 32 == 0, fp16 / bf16),
 * nibbles ordered so that two bit operations per dword give the MFMA's int8 operand (chatglm_q_amd/csrc/w4a8.hip). */
size_t qlinear_w4a8_packed_bytes(int64_t N, int64_t K, int64_t group, int dtype);
int qlinear_w4a8_pack(const uint8_t* Wq, const void* S, void* packed_a8, int64_t N, int64_t K, int64_t group, int dtype,
                      void* stream);
int qlinear_w4a8_fwd(const int8_t* Aq, const float* a_scale, const void* packed_a8, const void* bias, void* C, int64_t M,
                     int64_t N, int64_t K, int64_t ldc, int dtype, void* stream);
/* quantise A (flags: 0 row-wise, QL_FLAG_ACT_PER_TENSOR) into the workspace, then qlinear_w4a8_fwd: one call, two launches;
 * workspace as for qlinear_w8a8_linear_tiled (QL_OP_W4A8_LINEAR). */
int qlinear_w4a8_linear(const void* A, const void* packed_a8, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                        int64_t lda, int64_t ldc, int dtype, int flags, void* workspace, size_t workspace_bytes, void* stream);

/* round 4 experiment (w4_dense256.hip): many-row weight-only GEMM as two launches - the call's weights dequantised ONCE into a 16-bit
 * fragment-major image of qlinear_dev_dense256_image_bytes(N, K) bytes (from part 2 of the int4 layout, weight_bits 4, or the tile-major
 * int8 copy + its scales S, weight_bits 8; the reference's per-weight rounding), then a dense ring GEMM on that image; gate: SiLU * gate
 * epilogue on a gate-interleaved copy (C has N / 2 columns), residual nullable. */
size_t qlinear_dev_dense256_image_bytes(int64_t N, int64_t K);
int qlinear_dev_dense256_expand(const void* tiled, const void* S, void* image, int64_t N, int64_t K, int dtype, int weight_bits, void* stream);
int qlinear_dev_dense256_fwd(const void* A, const void* image, const void* bias, const void* residual, void* C, int64_t M, int64_t N, int64_t K,
                             int64_t lda, int64_t ldc, int64_t ldr, int dtype, int gate, void* stream);


/* round 6 experiment, measured SLOWER and therefore here (profiles/r06_w8a8_config3_splitk.txt: 21.5 us against 17.0 for the GEMM of
 * config 3 - the load-only floor fell from 12.7 to 9.3 us as the probe promised, the K loop from 10.4 to 9.5 us, and the exact int32 hand-off
 * between the two workgroups of a tile costs 2.2 us on the publishing and 6.4 us on the finishing side).
 * int8 activations x tile-major int8 weights on 128 x 128 tiles with a GRID-level K split by 2 (csrc/w8a8.hip, SK = 2): the shape
 * class of BASELINE config 3 (512 x 4096 -> 4096: too few rows to cover the chip with tall tiles), where the 64 x 128-tile kernel of
 * qlinear_w8a8_fwd_tiled pulls 201 MB of operands out of the L2s for 25 MB of data.  Twice the tile, half the K range per workgroup,
 * one exact int32 hand-off per tile pair through `workspace`: results are bit-equal to qlinear_w8a8_fwd_tiled (integer sums, the same
 * epilogue).  qlinear_dev_w8a8_splitk_workspace_bytes: 0 when the shape is not served (then call qlinear_w8a8_fwd_tiled), else the size of
 * `workspace` - which the caller zeroes ONCE after allocating it and then only hands back: the kernel keeps per-tile ticket counters in
 * it that count up across calls (their parity is a workgroup's role, their value the epoch of the hand-off flag).  One workspace per
 * stream of launches (two launches must not share it concurrently).  QL_ERR_UNSUPPORTED for other shapes, QL_ERR_WORKSPACE when small. */
size_t qlinear_dev_w8a8_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K);
int qlinear_dev_w8a8_fwd_tiled_splitk(const int8_t* Aq, const float* a_scale, const void* tiled, const void* S, const void* bias, void* C,
                                  int64_t M, int64_t N, int64_t K, int64_t ldc, int dtype, void* workspace, size_t workspace_bytes,
                                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QLINEAR_HIP_DEV_H */
