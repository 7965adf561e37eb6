// MFMA fragment traits shared by the int4g32 GEMM kernels (w4_gemm.hip, w4_skinny.hip): the 32x32x16 MFMA of the
// activation dtype and the in-register dequantisation of one 32-bit word (8 nibbles) into a B fragment with the
// reference's rounding (every weight (n - 8) * s rounded to the activation dtype, chatglm_q/int4/triton_ops.py:72-73).
#pragma once
#include "ql_common.h"
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
    // dword i (0..3) of the same fragment alone: lets a kernel spread the dequant of the next fragment over the MFMAs of
    // the current one, instruction by instruction
    static __device__ __forceinline__ u32 dequant_part(u32 w, int i, u32 k_mask_lo, u32 k_mask_hi, u32 k_magic, h2 s2) {
        const h2 k1032 = {(f16)1032.0f, (f16)1032.0f};
        const h2 kInv16 = {(f16)0.0625f, (f16)0.0625f};
        const h2 kM72 = {(f16)-72.0f, (f16)-72.0f};
        const u32 v = i >= 2 ? w >> 8 : w;
        if (i & 1) return as_u32((as_h2((v & k_mask_hi) | k_magic) * kInv16 + kM72) * s2);
        return as_u32((as_h2((v & k_mask_lo) | k_magic) - k1032) * s2);
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
    static __device__ __forceinline__ u32 dequant_part(u32 w, int i, u32 k_mask_lo, u32, u32 k_magic, float s) {
        const float m136s = -136.0f * s;
        const u32 x = ((w >> (4 * i)) & k_mask_lo) | k_magic;
        const float lo = __builtin_fmaf(u32_as_f32(x << 16), s, m136s);
        const float hi = __builtin_fmaf(u32_as_f32(x & 0xFFFF0000u), s, m136s);
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        const bf2 p = {(__bf16)lo, (__bf16)hi};
        return __builtin_bit_cast(u32, p);
    }
    static __device__ __forceinline__ float scale_pair(const __bf16* p, bool valid) { return valid ? (float)*p : 0.f; }
    static constexpr u32 kMagic = 0x43004300u;
};

}  // namespace ql
