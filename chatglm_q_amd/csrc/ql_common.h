// Shared device helpers for the gfx950 quantized-linear kernels.
// CDNA4 only: wave64, v_pk_*_f16, v_dot2c_f32_f16, v_cvt_pk_bf16_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qlinear_hip.h"

namespace ql {

typedef uint32_t u32;
typedef _Float16 f16;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

__device__ __forceinline__ h2 as_h2(u32 u) { return __builtin_bit_cast(h2, u); }
__device__ __forceinline__ u32 as_u32(h2 h) { return __builtin_bit_cast(u32, h); }
__device__ __forceinline__ float u32_as_f32(u32 u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ u32 f32_as_u32(float f) { return __builtin_bit_cast(u32, f); }

// ---------------------------------------------------------------------------------------------
// Activation dtype traits.  "round()" rounds an fp32 value to the activation dtype and returns it
// as fp32: the reference rounds every dequantised weight to the activation dtype before the dot
// (chatglm_q/int4/triton_ops.py:72-73) and the output once after it (:80).
// ---------------------------------------------------------------------------------------------
template <typename T> struct Act;

template <> struct Act<float> {
    static constexpr int code = QL_DTYPE_F32;
    typedef float storage;
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ float round(float x) { return x; }
    static __device__ __forceinline__ void store(float* p, float x) { *p = x; }
};

template <> struct Act<f16> {
    static constexpr int code = QL_DTYPE_F16;
    typedef f16 storage;
    static __device__ __forceinline__ float load(const f16* p) { return (float)*p; }
    static __device__ __forceinline__ float round(float x) { return (float)(f16)x; }
    static __device__ __forceinline__ void store(f16* p, float x) { *p = (f16)x; }
};

template <> struct Act<__bf16> {
    static constexpr int code = QL_DTYPE_BF16;
    typedef __bf16 storage;
    static __device__ __forceinline__ float load(const __bf16* p) { return (float)*p; }
    static __device__ __forceinline__ float round(float x) { return (float)(__bf16)x; }
    static __device__ __forceinline__ void store(__bf16* p, float x) { *p = (__bf16)x; }
};

// Output epilogue shared by every kernel: one rounding of the fp32 accumulator to the activation
// dtype, then the bias add as a second rounded operation - exactly the reference's
// "acc.to(C.dtype)" (chatglm_q/int4/triton_ops.py:80) followed by the in-place "out += bias"
// (chatglm_q/int4/qlinear.py:92-93).
template <typename T>
__device__ __forceinline__ void store_out(T* c, float acc, const T* bias_n) {
    float y = Act<T>::round(acc);
    if (bias_n) y = y + Act<T>::load(bias_n);
    Act<T>::store(c, y);
}

// ---------------------------------------------------------------------------------------------
// Wave-level helpers (wave64).
// ---------------------------------------------------------------------------------------------
// Sum over the 64 lanes of a wave, result in every lane.  Four DPP steps (full-rate VALU modifiers, no
// LDS round trip as ds_bpermute-based shuffles have) leave each 16-lane row's sum in all of its lanes;
// the four row sums are then read through SGPRs.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]  : lane ^ 1
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]  : lane ^ 2
    v += dpp_move<0x141>(v);   // row_half_mirror      : i <-> 7 - i  (joins the two quads of each 8)
    v += dpp_move<0x140>(v);   // row_mirror           : i <-> 15 - i (joins the two halves of the row)
    const int iv = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48)));
}

// 8 activation-dtype values held in one 16-byte register quad <-> fp32 (16-bit dtypes only)
template <typename T>
__device__ __forceinline__ void unpack8(u32x4 r, float (&v)[8]) {
    static_assert(sizeof(T) == 2, "16-bit activation dtypes");
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        T lo, hi;
        const uint16_t l16 = (uint16_t)(r[e] & 0xFFFFu), h16 = (uint16_t)(r[e] >> 16);
        __builtin_memcpy(&lo, &l16, 2);
        __builtin_memcpy(&hi, &h16, 2);
        v[2 * e] = (float)lo;
        v[2 * e + 1] = (float)hi;
    }
}
template <typename T>
__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {   // rounds each value to T
    static_assert(sizeof(T) == 2, "16-bit activation dtypes");
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const T lo = (T)v[2 * e], hi = (T)v[2 * e + 1];
        uint16_t l16, h16;
        __builtin_memcpy(&l16, &lo, 2);
        __builtin_memcpy(&h16, &hi, 2);
        r[e] = (u32)l16 | ((u32)h16 << 16);
    }
    return r;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

}  // namespace ql
