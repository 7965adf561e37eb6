// int4 nibble -> activation-dtype dequantisation primitives (gfx950).
//
// Semantics restated from the reference (chatglm_q/int4/triton_ops.py:71-73,
// chatglm_q/int4/qlinear.py:29-32): value = (nibble - 8) * scale, the product ROUNDED to the
// activation dtype before it meets the activation; accumulate in fp32.
#pragma once
#include "ql_common.h"

namespace ql {

// ---- fp16: exponent-splice trick -------------------------------------------------------------
// 0x6400 is fp16 1024.0 with ulp 1, so (0x6400 | n) == 1024 + n exactly for n in [0,15], and
// (0x6400 | (n << 4)) == 1024 + 16 n.  One v_and_or_b32 therefore converts TWO nibbles (one per
// 16-bit half of the word) and one packed op removes the offset:
//   low-nibble form  : (1024 + n) - 1032            = n - 8   (exact)
//   high-nibble form : (1024 + 16 n) * 1/16 - 72    = n - 8   (exact, single v_pk_fma_f16)
// A 32-bit word holds 8 nibbles p0..p7 (p0 = bits 3:0).  quad_from_word() returns the four exact
// half2 values  E0 = (p0, p4), E1 = (p1, p5), E2 = (p2, p6), E3 = (p3, p7), each already minus 8.
struct NibblePairs {
    h2 e0, e1, e2, e3;
};

__device__ __forceinline__ NibblePairs nibble_pairs_f16(u32 w) {
    const u32 kMagic = 0x64006400u;
    const h2 k1032 = {(f16)1032.0f, (f16)1032.0f};
    const h2 kInv16 = {(f16)0.0625f, (f16)0.0625f};
    const h2 kM72 = {(f16)-72.0f, (f16)-72.0f};
    const u32 w8 = w >> 8;
    NibblePairs r;
    r.e0 = as_h2((w & 0x000F000Fu) | kMagic) - k1032;
    r.e1 = as_h2((w & 0x00F000F0u) | kMagic) * kInv16 + kM72;   // contracts to v_pk_fma_f16 (exact)
    r.e2 = as_h2((w8 & 0x000F000Fu) | kMagic) - k1032;
    r.e3 = as_h2((w8 & 0x00F000F0u) | kMagic) * kInv16 + kM72;
    return r;
}

// ---- generic (fp32 / bf16): per-nibble -------------------------------------------------------
// (n - 8) * s has at most 4 + 24 significant bits of product, so fma(n, s, -8 s) (with -8 s exact)
// is the correctly rounded fp32 product; for bf16 the product is exact in fp32 and is then rounded
// once to bf16 (v_cvt_pk_bf16_f32).
template <typename T>
__device__ __forceinline__ float dequant_nibble(u32 w, int p, float s, float m8s) {
    const float n = (float)((w >> (4 * p)) & 0xFu);
    return Act<T>::round(__builtin_fmaf(n, s, m8s));
}

}  // namespace ql
