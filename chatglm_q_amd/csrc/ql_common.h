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
__device__ __forceinline__ float i32_as_f32(int i) { return __builtin_bit_cast(float, i); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
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

// two activations <-> one 32-bit word (16-bit dtypes)
template <typename T>
__device__ __forceinline__ u32 pack2(float a, float b) {
    const T x = (T)a, y = (T)b;
    uint16_t lo, hi;
    __builtin_memcpy(&lo, &x, 2);
    __builtin_memcpy(&hi, &y, 2);
    return (u32)lo | ((u32)hi << 16);
}
template <typename T>
__device__ __forceinline__ void unpack2(u32 w, float& a, float& b) {
    const uint16_t lo = (uint16_t)(w & 0xFFFFu), hi = (uint16_t)(w >> 16);
    T x, y;
    __builtin_memcpy(&x, &lo, 2);
    __builtin_memcpy(&y, &hi, 2);
    a = (float)x;
    b = (float)y;
}

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

// Epilogue of the 32x32 MFMA accumulator tiles (16-bit dtypes): the C/D layout gives a lane ONE column and 16 rows,
// i.e. 2-byte global stores (measured: the epilogue was 40 % of the 512 x 4096 x 4096 int8 GEMM and 10 % of the
// int4 GEMM at M = 8192).  The wave writes its rounded tile into a private 2 KB LDS region [32 rows][32 cols] and
// reads it back as 16-byte row chunks: 2 stores per lane instead of 16.  `val(i)` = final fp32 value of accumulator
// element i (row (i & 3) + 8 (i >> 2) + 4 kb, column j) BEFORE the output rounding; bias (nullable) is added as a
// second rounded operation (store_out's sequence).  Requires 16-byte aligned C rows (n0 % 8 == 0, ldc % 8 == 0).
#ifndef QL_STORE_TILE_WAIT
#define QL_STORE_TILE_WAIT 0
#endif
// Which element of a 32 x 32 output block accumulator value i (0..15) of a lane is.  LAYOUT 0: one v_mfma_*_32x32 tile (row (i & 3) +
// 8 (i >> 2) + 4 (lane >> 5), column lane & 31).  LAYOUT 1 (round 4): 2 x 2 v_mfma_*_16x16 tiles, i = r + 4 dn + 8 dm: tile (dm, dn)
// register r = row 16 dm + 4 (lane >> 4) + r, column 16 dn + (lane & 15).
template <int LAYOUT>
struct Tile32 {
    static __device__ __forceinline__ int row(int i, int lane) {
        return LAYOUT == 0 ? (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5) : 16 * (i >> 3) + 4 * (lane >> 4) + (i & 3);
    }
    static __device__ __forceinline__ int col(int i, int lane) { return LAYOUT == 0 ? (lane & 31) : 16 * ((i >> 2) & 1) + (lane & 15); }
};
template <typename T, int LAYOUT = 0, bool NT = false, typename F>
__device__ __forceinline__ void store_tile_32x32(T* lds_wave, T* __restrict__ C, int64_t ldc, int m_base, int n0, int M, int N,
                                                 const T* __restrict__ bias, int lane, F val) {
    static_assert(sizeof(T) == 2, "16-bit outputs");
    typedef Tile32<LAYOUT> TM;
    const int c0 = TM::col(0, lane), c1 = TM::col(4, lane);      // the lane's column(s): one (LAYOUT 0) or two
    const float b0 = (bias && n0 + c0 < N) ? Act<T>::load(bias + n0 + c0) : 0.f;
    const float b1 = (LAYOUT != 0 && bias && n0 + c1 < N) ? Act<T>::load(bias + n0 + c1) : b0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float y = Act<T>::round(val(i));
        if (bias) y = y + ((LAYOUT != 0 && ((i >> 2) & 1)) ? b1 : b0);
        Act<T>::store(lds_wave + TM::row(i, lane) * 32 + TM::col(i, lane), y);
    }
    // One wave's LDS instructions execute in issue order, so the reads below see the writes above without a wait (round 3: the
    // explicit lgkmcnt(0) here cost one LDS round trip per 32 x 32 tile, 8 tiles per wave in the 256 x 256-tile GEMMs); the wave
    // barrier only keeps the compiler from moving the reads up.
    __builtin_amdgcn_wave_barrier();
#if QL_STORE_TILE_WAIT
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = lane + 64 * h, row = q >> 2, c8 = (q & 3) * 8;
        const int m = m_base + row, n = n0 + c8;
        if (m >= M || n >= N) continue;
        if (n + 8 <= N) {
            const u32x4 chunk = *reinterpret_cast<const u32x4*>(lds_wave + row * 32 + c8);
            if constexpr (NT) __builtin_nontemporal_store(chunk, reinterpret_cast<u32x4*>(C + (int64_t)m * ldc + n));   // many-row GEMMs: read by the next launch at the earliest
            else *reinterpret_cast<u32x4*>(C + (int64_t)m * ldc + n) = chunk;
        } else {                                 // ragged last chunk of the matrix: element by element from LDS
            for (int e = 0; e < N - n; ++e) C[(int64_t)m * ldc + n + e] = lds_wave[row * 32 + c8 + e];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// XCD-aware tile order for 1-D grids.  Workgroups are dealt to the 8 XCDs round-robin by linear id, each XCD
// with its own L2.  Block `id` of `total` becomes position P in a tile order such that XCD c works on one
// contiguous range of P (bijective for any total): tiles that share an operand panel share an L2.
// nbx > 0: P runs row-major (x fastest; an XCD owns whole row tiles, the A panel is fetched once);
// nbx < 0: P runs column-major over |nbx| columns and `nby` rows (an XCD owns whole column tiles: few-row
// shapes, where the weights are the large operand).
struct TileXY {
    int x, y;
};
__host__ __device__ __forceinline__ TileXY xcd_tile(unsigned id, unsigned total, int nbx) {
    const unsigned c = id & 7u, i = id >> 3, q = total >> 3, r = total & 7u;
    const unsigned p = (c < r ? c * (q + 1) : r * (q + 1) + (c - r) * q) + i;
    if (nbx > 0) return {(int)(p % (unsigned)nbx), (int)(p / (unsigned)nbx)};
    const unsigned nby = total / (unsigned)(-nbx);
    return {(int)(p / nby), (int)(p % nby)};
}
// Grouped order for many-row GEMMs: an XCD's consecutive positions walk GROUPS of `sy` tile rows column by column (sy tiles down, then
// the next column), so that the blocks it runs AT THE SAME TIME (32 CUs x 1 block) are 32 / sy columns x sy rows and share 32 / sy + sy
// operand panels instead of 32 + 1.  Measured reason (profiles/r02_w8a8_l2_pmc.txt): with whole row tiles per XCD the 8192 x 4096 x
// 4096 int8 GEMM hit its L2s only 50 % of the time and pulled 1.09 GB over the fabric - the L2s hold 4 MB, so what counts is which
// tiles are in flight together, not which ones an XCD owns.  Any nbx / nby (round 4: the last group may be shorter; before, the
// order needed nbx % 8 == 0 and the widest layer shapes - 107 and 18 column tiles - fell back to whole rows: 33 panels per 32 tiles).
__host__ __device__ __forceinline__ TileXY xcd_tile_super(unsigned id, unsigned total, int nbx, int sy) {
    const unsigned c = id & 7u, i = id >> 3, q = total >> 3, r = total & 7u;
    const unsigned p = (c < r ? c * (q + 1) : r * (q + 1) + (c - r) * q) + i;
    const unsigned nby = total / (unsigned)nbx, per = (unsigned)sy * (unsigned)nbx, g = p / per, w = p - g * per;
    const unsigned left = nby - g * (unsigned)sy, rows = left < (unsigned)sy ? left : (unsigned)sy;
    return {(int)(w / rows), (int)(g * (unsigned)sy + w % rows)};
}
// which order moves fewer bytes into the L2s: a_bytes / w_bytes are the whole operands
inline int xcd_order(int nbx, int nby, double a_bytes, double w_bytes) {
    const double row_major = a_bytes + w_bytes * (nby < 8 ? nby : 8), col_major = a_bytes * (nbx < 8 ? nbx : 8) + w_bytes;
    return row_major <= col_major ? nbx : -nbx;
}

// Split-K epilogue shared by the canonical-layout kernel and the MFMA GEMMs: sum the fp32 slabs
// partial[s][m][n], round once, add bias, store.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, const T* __restrict__ bias,
                                                            T* __restrict__ C, int M, int N, int64_t ldc, int ksplit) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    float v = 0.f;
    for (int s = 0; s < ksplit; ++s) v += partial[((int64_t)s * M + m) * N + n];
    store_out<T>(C + (int64_t)m * ldc + n, v, bias ? bias + n : nullptr);
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

// sum over aligned groups of W adjacent lanes (W = 4, 8, 16), result in every lane of the group: DPP only
template <int W>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(W == 4 || W == 8 || W == 16, "group of 4, 8 or 16 lanes");
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    if constexpr (W >= 8) v += dpp_move<0x141>(v);
    if constexpr (W >= 16) v += dpp_move<0x140>(v);
    return v;
}
// maximum over the 64 lanes, result in every lane (DPP + readlane, like wave_sum)
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    const int iv = __builtin_bit_cast(int, v);
    return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))),
                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48))));
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

// The same epilogue with the residual stream added (chatglm_q/model.py:243,245: hidden = hidden + sublayer(...)): every 8-column
// row chunk of the rounded tile meets the same chunk of `resid` (row stride ldr) and leaves as round(y + resid) - the sublayer's
// output is rounded to T first, as the reference materialises it.  16-byte aligned rows of C and resid (checked by the ABI).
template <typename T, int LAYOUT = 0, bool NT = false, typename F>
__device__ __forceinline__ void store_tile_32x32_resid(T* lds_wave, T* __restrict__ C, int64_t ldc, const T* __restrict__ resid,
                                                       int64_t ldr, int m_base, int n0, int M, int N, const T* __restrict__ bias,
                                                       int lane, F val) {
    static_assert(sizeof(T) == 2, "16-bit outputs");
    typedef Tile32<LAYOUT> TM;
    const int c0 = TM::col(0, lane), c1 = TM::col(4, lane);      // the lane's column(s): one (LAYOUT 0) or two
    const float b0 = (bias && n0 + c0 < N) ? Act<T>::load(bias + n0 + c0) : 0.f;
    const float b1 = (LAYOUT != 0 && bias && n0 + c1 < N) ? Act<T>::load(bias + n0 + c1) : b0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float y = Act<T>::round(val(i));
        if (bias) y = y + ((LAYOUT != 0 && ((i >> 2) & 1)) ? b1 : b0);
        Act<T>::store(lds_wave + TM::row(i, lane) * 32 + TM::col(i, lane), y);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = lane + 64 * h, row = q >> 2, c8 = (q & 3) * 8;
        const int m = m_base + row, n = n0 + c8;
        if (m >= M || n >= N) continue;
        if (n + 8 <= N) {
            float y[8], r[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(lds_wave + row * 32 + c8), y);
            unpack8<T>(*reinterpret_cast<const u32x4*>(resid + (int64_t)m * ldr + n), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = y[e] + r[e];
            if constexpr (NT) __builtin_nontemporal_store(pack8<T>(y), reinterpret_cast<u32x4*>(C + (int64_t)m * ldc + n));
            else *reinterpret_cast<u32x4*>(C + (int64_t)m * ldc + n) = pack8<T>(y);
        } else {                                 // ragged last chunk of the matrix: element by element
            for (int e = 0; e < N - n; ++e)
                Act<T>::store(C + (int64_t)m * ldc + n + e, (float)lds_wave[row * 32 + c8 + e] + Act<T>::load(resid + (int64_t)m * ldr + n + e));
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// The same epilogue with SiLU * gate (chatglm_q/model.py:200-201) on GATE-INTERLEAVED columns: column quad t of the weight copy is
// (h[2t], h[2t+1], gate[2t], gate[2t+1]), so an 8-column row chunk of the tile holds two complete quads and becomes 4 outputs -
// out[2t + i] = round(round(silu(y_i)) * y_{i+2}), y = rounded sum (+ bias as a second rounded operation) - stored as ONE 8-byte
// chunk of C, which has N / 2 columns.  Requires N % 8 == 0, ldc % 4 == 0 and 8-byte aligned C rows.
template <typename T, int LAYOUT = 0, bool NT = false, typename F>
__device__ __forceinline__ void store_tile_32x32_gated(T* lds_wave, T* __restrict__ C, int64_t ldc, int m_base, int n0, int M, int N,
                                                       const T* __restrict__ bias, int lane, F val) {
    static_assert(sizeof(T) == 2, "16-bit outputs");
    typedef Tile32<LAYOUT> TM;
    const int c0 = TM::col(0, lane), c1 = TM::col(4, lane);      // the lane's column(s): one (LAYOUT 0) or two
    const float b0 = (bias && n0 + c0 < N) ? Act<T>::load(bias + n0 + c0) : 0.f;
    const float b1 = (LAYOUT != 0 && bias && n0 + c1 < N) ? Act<T>::load(bias + n0 + c1) : b0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float y = Act<T>::round(val(i));
        if (bias) y = y + ((LAYOUT != 0 && ((i >> 2) & 1)) ? b1 : b0);
        Act<T>::store(lds_wave + TM::row(i, lane) * 32 + TM::col(i, lane), y);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = lane + 64 * h, row = q >> 2, c8 = (q & 3) * 8;
        const int m = m_base + row, n = n0 + c8;
        if (m >= M || n >= N) continue;
        float y[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(lds_wave + row * 32 + c8), y);
        float o[4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float hv = y[4 * t + i], gv = y[4 * t + 2 + i];
                o[2 * t + i] = Act<T>::round(hv / (1.0f + __expf(-hv))) * gv;
            }
        const u32x2 packed = {pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])};
        if constexpr (NT) __builtin_nontemporal_store(packed, reinterpret_cast<u32x2*>(C + (int64_t)m * ldc + (n >> 1)));
        else *reinterpret_cast<u32x2*>(C + (int64_t)m * ldc + (n >> 1)) = packed;
    }
    __builtin_amdgcn_wave_barrier();
}


__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

}  // namespace ql
