// Exponent-splice traits of the int4g32 one-row kernels (w4_packed.hip, w4_engine.hip): how a word of eight nibbles becomes
// v_dot2c operands, and where a 16-byte activation chunk lives in the LDS image of a row.
#pragma once
#include "ql_common.h"

namespace ql {

// LDS image of the activation rows: 16-byte chunk cc of row m (k = 8 cc .. 8 cc + 7) lives at chunk position
// m * K/8 + 4 g + (j ^ ((g >> 2) & 3)) with g = cc >> 2 (its group), j = cc & 3.  A lane reads the 4 chunks of ITS group
// (64-byte lane stride); the XOR spreads each 16-lane ds_read_b128 service group over all 16 four-bank slots.
__device__ __forceinline__ int a_chunk_pos(int g, int j) { return 4 * g + (j ^ ((g >> 2) & 3)); }

template <typename T> struct Splice;
template <> struct Splice<f16> {
    // 0x6400 | n = 1024 + n for a nibble at mantissa bits 0..3, 0x6400 | (n << 4) = 1024 + 16 n for one at bits 4..7:
    // nibble pairs 0 and 1 of a word (bits 0..3 / 16..19 and 4..7 / 20..23) are spliced where they lie, pairs 2 and
    // 3 after ONE shift by 8 (one shift per word instead of three to bring every pair to the same bits).  The even
    // pairs feed the `e` chain (unit weight), the odd pairs the `o` chain (16 x): sum (n - 8) a =
    //   e + o / 16 - (1032 sum_e a + 72 sum_o a)
    static constexpr u32 kMagic = 0x64006400u, kMask = 0x000F000Fu, kMaskOdd = 0x00F000F0u, kOnes = 0x3C003C00u;
    static constexpr bool kSplitChains = true;
    static __device__ __forceinline__ float combine(float e, float o) { return __builtin_fmaf(o, 0.0625f, e); }
    static __device__ __forceinline__ float offset(float ae, float ao) { return __builtin_fmaf(1032.0f, ae, 72.0f * ao); }
    static __device__ __forceinline__ float dot(u32 x, u32 a, float acc) {
        return __builtin_amdgcn_fdot2(as_h2(x), as_h2(a), acc, false);
    }
    static __device__ __forceinline__ float lo(u32 s) { return (float)as_h2(s).x; }
    static __device__ __forceinline__ float hi(u32 s) { return (float)as_h2(s).y; }
};
template <> struct Splice<__bf16> {
    // 0x4300 | n = 128 + n (7 mantissa bits: only the low nibble position splices); every pair is shifted to bits 0..3
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    static constexpr u32 kMagic = 0x43004300u, kMask = 0x000F000Fu, kMaskOdd = 0x000F000Fu, kOnes = 0x3F803F80u;
    static constexpr bool kSplitChains = false;
    static __device__ __forceinline__ float combine(float e, float o) { return e + o; }
    static __device__ __forceinline__ float offset(float ae, float ao) { return 136.0f * (ae + ao); }
    static __device__ __forceinline__ float dot(u32 x, u32 a, float acc) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x), __builtin_bit_cast(bf2, a), acc, false);
    }
    static __device__ __forceinline__ float lo(u32 s) { return u32_as_f32(s << 16); }
    static __device__ __forceinline__ float hi(u32 s) { return u32_as_f32(s & 0xFFFF0000u); }
};


}  // namespace ql
