// Many-position (prefill) attention in ONE launch on the matrix cores, gfx950 - the caller-side op between qkv_proj and
// o_proj of a prefill chunk (chatglm_q/model.py:157-175: q / sqrt(d), q k^T, + mask, fp32
// softmax, cast, p v).  The reference materialises the (heads x S x T) score matrix three times (GEMM out, softmax in / out,
// GEMM in); for a chunk of 1024 positions against 2048 keys that is 134 MB per sequence and layer.  Here a workgroup keeps a
// block of query rows in registers and walks the keys 64 at a time with a running maximum and exp-sum per row.
//
// ChatGLM2's geometry only (D = 128, 16 query heads per key / value group, fp16 / bf16) - the same condition as
// decode_attention_mfma_kernel, whose operand idioms this kernel reuses:
//   S^T (keys x heads)  = K (keys x d) Q^T        v_mfma_f32_16x16x32: A = key rows from LDS (pitch 272 B), B = a query
//                                                 position's 16 heads, held in registers for the whole kernel
//   O^T (d x heads)    += V^T (d x keys) P^T      v_mfma_f32_16x16x32: B = exp(S^T - max) exactly as the first product's C
//                                                 layout leaves it (slot 8 q + i of a 32-key step = key 16 (i / 4) + 4 q + i % 4
//                                                 of the step), A = two ds_read_b64_tr_b16 of row-major value rows (pitch 288 B)
// A wave owns R = 2 query positions x the group's 16 heads (every key / value fragment read from LDS feeds two MFMAs); a
// workgroup of NWV waves owns 2 NWV consecutive positions of one (sequence, group) and stages each 64-key tile of K and V
// once (global -> registers -> LDS, double buffered, one barrier per tile).  Rows of a block differ by at most 2 NWV - 1
// positions, so a causal mask cuts whole key tiles for the whole block.
//
// Mask: the reference's additive fp32 mask (B, S, T), applied exactly as the reference does - round(score) + mask in fp32 -
// so a row whose keys are ALL blocked comes out as the uniform average, like the reference's.  `tile_flags` (optional,
// built by the host from the same mask, fused_ops.attention_tile_flags) says per (sequence, query block, key tile):
//   0 = skip: every entry <= -1e9 and every row of the block has an entry >= -1e6 somewhere, so the tile's probabilities
//       are exactly 0 in fp32 whatever the scores are (|score| <= 65504);  2 = every entry is 0: no mask loads;  1 = load it.
// Rounding points (T = activation dtype): q / sqrt(d), the scores, the output - as the reference; P is rounded to T as
// exp(s - running max) and normalised after the second product (the reference rounds exp(s - max) / sum: the same
// relative error, not the same bits); the exp-sum uses the unrounded values.
#include "launch.h"
#include "ql_common.h"
#include <type_traits>

namespace ql {

typedef short pf_s16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct PfMma;
template <> struct PfMma<f16> {
    typedef _Float16 v8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
    }
};
template <> struct PfMma<__bf16> {
    typedef __bf16 v8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
    }
};

constexpr int kPfKeys = 64;                                   // keys per tile
constexpr int kPfKP = 272;                                    // bytes per key row in LDS (b128 fragment reads conflict-free)
constexpr int kPfVP = 288;                                    // bytes per value row in LDS (tr reads conflict-free)
constexpr int kPfR = 2;                                       // query positions per wave
constexpr int kPfLds = 2 * kPfKeys * (kPfKP + kPfVP);         // two buffers of a K and a V tile: 71 680 bytes

template <typename T, int NWV>
__global__ __launch_bounds__(NWV * 64) void prefill_attention_kernel(const T* __restrict__ Q, const T* __restrict__ Kc,
                                                                     const T* __restrict__ Vc, const float* __restrict__ mask,
                                                                     const uint8_t* __restrict__ flags, T* __restrict__ Out,
                                                                     int S, int Tkv, int H, int G, int cap, int nqb, int nkt,
                                                                     int64_t ldm, float sqrt_d) {
    static_assert(sizeof(T) == 2, "16-bit dtypes");
    constexpr int D = 128, HP = 16, R = kPfR, QB = R * NWV, NTH = NWV * 64, KP = kPfKP, VP = kPfVP;
    constexpr int CH = kPfKeys * 16 / NTH;                    // 16-byte chunks of a K (and of a V) tile staged per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char pf_smem[];
    unsigned char* const kimg = pf_smem;                      // [2][64][KP]
    unsigned char* const vimg = pf_smem + 2 * kPfKeys * KP;   // [2][64][VP]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, q = lane >> 4;

    // Workgroups go to the 8 XCDs round robin; the blocks of one (sequence, group) share its keys and values, so they are
    // put on ONE XCD (its L2 then holds that 1 MB instead of every group's), longest blocks (latest positions) first.
    const int total = (int)gridDim.x;
    int v = (int)blockIdx.x;
    if ((total & 7) == 0) v = (v & 7) * (total >> 3) + (v >> 3);
    const int bg = v / nqb, qblk = nqb - 1 - (v - bg * nqb);
    const int b = bg / G, g = bg - b * G;
    const int pos0 = qblk * QB + wv * R;

    const float rcp_d = 1.0f / sqrt_d;
    // the wave's query rows: B operand of the first product, lane (li, q) = head li, d = 32 j + 8 q .. + 7
    u32x4 qf[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int pos = pos0 + r < S ? pos0 + r : S - 1;
        const T* qrow = Q + (((int64_t)b * S + pos) * H + g * HP + li) * D;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(qrow + 32 * j + 8 * q), x);
            // x / sqrt(d) as one Newton-corrected multiply (the IEEE division sequence is 12 instructions x 64 values per lane): the
            // corrected quotient is the correctly rounded one except when it lands within an fp32 ulp of a tie, and the result is
            // rounded to T right after
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float q0 = x[e] * rcp_d;
                x[e] = __builtin_fmaf(__builtin_fmaf(-q0, sqrt_d, x[e]), rcp_d, q0);
            }
            qf[r][j] = pack8<T>(x);
        }
    }
    const uint8_t* frow = flags ? flags + ((int64_t)b * nqb + qblk) * nkt : nullptr;
    auto next_tile = [&](int kt) {
        ++kt;
        while (frow && kt < nkt && frow[kt] == 0) ++kt;
        return kt;
    };

    const int64_t pitch = (int64_t)G * D;
    const T* kbase = Kc + ((int64_t)b * cap * G + g) * D;
    const T* vbase = Vc + ((int64_t)b * cap * G + g) * D;
    u32x4 kst[CH], vst[CH];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = tid + NTH * i, row = c >> 4, col = c & 15;
            const int t = kt * kPfKeys + row < Tkv ? kt * kPfKeys + row : Tkv - 1;
            kst[i] = *reinterpret_cast<const u32x4*>(kbase + t * pitch + 8 * col);
            vst[i] = *reinterpret_cast<const u32x4*>(vbase + t * pitch + 8 * col);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = tid + NTH * i, row = c >> 4, col = c & 15;
            *reinterpret_cast<u32x4*>(kimg + (buf * kPfKeys + row) * KP + 16 * col) = kst[i];
            *reinterpret_cast<u32x4*>(vimg + (buf * kPfKeys + row) * VP + 16 * col) = vst[i];
        }
    };

    float m[R], l[R];                                         // running maximum (same in the 4 lanes of a head), this lane's exp-sum
    f32x4 o[R][8];                                            // O^T: lane (li, q) = head li, d = 16 dt + 4 q + e
#pragma unroll
    for (int r = 0; r < R; ++r) {
        m[r] = -INFINITY;
        l[r] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) o[r][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    auto tile_body = [&](auto general_tag, const int t0, const int fl, const int buf) {
        constexpr bool GEN = decltype(general_tag)::value;
        const unsigned char* kb = kimg + buf * kPfKeys * KP;
        const unsigned char* vb = vimg + buf * kPfKeys * VP;
        float mv[GEN ? R : 1][4][4];
        if constexpr (GEN) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int pos = pos0 + r < S ? pos0 + r : S - 1;
                // unconditional loads (a load under a per-element condition becomes a branch per element): tiles that only need
                // the range check read in-bounds bytes of the key cache and select 0
                const float* mrow = fl == 1 ? mask + ((int64_t)b * S + pos) * ldm : reinterpret_cast<const float*>(kbase);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = t0 + 16 * pt + 4 * q + e;
                        const float lv = mrow[key < Tkv ? key : Tkv - 1];
                        mv[r][pt][e] = fl == 1 ? lv : 0.f;
                    }
            }
        }

        // S^T = K Q^T: lane (li, q) ends with the scores of head li against keys t0 + 16 pt + 4 q + e.  d chunk j outermost:
        // consecutive MFMAs go to 8 different accumulators (a dependent 16 x 16 x 32 pair costs the pipe's full latency)
        f32x4 s[R][4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int r = 0; r < R; ++r) s[r][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(kb + (16 * pt + li) * KP + 64 * j + 16 * q);
#pragma unroll
                for (int r = 0; r < R; ++r) s[r][pt] = PfMma<T>::mma(kf, qf[r][j], s[r][pt]);
            }

        u32x2 pf[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float mx = -INFINITY;
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = Act<T>::round(s[r][pt][e]);
                    if constexpr (GEN) {
                        x += mv[r][pt][e];
                        if (t0 + 16 * pt + 4 * q + e >= Tkv) x = -INFINITY;
                    }
                    s[r][pt][e] = x;
                    mx = fmaxf(mx, x);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(m[r], mx);                 // finite: key t0 is in range and its entry is finite
            const float alpha = __expf(m[r] - mn);            // first tile: exp(-inf) = 0
            m[r] = mn;
            float ls = 0.f;
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                float ev[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ev[e] = __expf(s[r][pt][e] - mn);
                    ls += ev[e];
                }
                pf[r][pt] = u32x2{pack2<T>(ev[0], ev[1]), pack2<T>(ev[2], ev[3])};
            }
            l[r] = __builtin_fmaf(l[r], alpha, ls);
            if (!__all(alpha == 1.0f)) {                      // wave-uniform: the maximum settles after the first tiles
#pragma unroll
                for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[r][dt][e] *= alpha;
            }
        }

        // O^T += V^T P^T, 32 keys per MFMA step
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                const unsigned char* p0 = vb + (32 * c + 4 * q + (li >> 2)) * VP + 2 * (16 * dt + 4 * (li & 3));
                const pf_s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (pf_s16x4 __attribute__((address_space(3)))*)(const_cast<unsigned char*>(p0)));
                const pf_s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (pf_s16x4 __attribute__((address_space(3)))*)(const_cast<unsigned char*>(p0 + 16 * VP)));
                const u32x2 w0 = __builtin_bit_cast(u32x2, a0), w1 = __builtin_bit_cast(u32x2, a1);
                const u32x4 a = u32x4{w0[0], w0[1], w1[0], w1[1]};
#pragma unroll
                for (int r = 0; r < R; ++r)
                    o[r][dt] = PfMma<T>::mma(a, u32x4{pf[r][2 * c][0], pf[r][2 * c][1], pf[r][2 * c + 1][0], pf[r][2 * c + 1][1]}, o[r][dt]);
            }
        }
    };

    int kt = next_tile(-1);
    if (kt < nkt) {
        load_tile(kt);
        store_tile(0);
    }
    __syncthreads();
    int buf = 0;
    while (kt < nkt) {
        const int nk = next_tile(kt);
        if (nk < nkt) load_tile(nk);                          // in flight under this tile's arithmetic

        const int t0 = kt * kPfKeys;
        const int fl = !mask ? 2 : frow ? (int)frow[kt] : 1;   // no mask at all: every tile is 'all zero'
        const bool ragged = t0 + kPfKeys > Tkv;              // the last tile of a T that is not a multiple of 64
        // two copies of the tile body: the general one (mask loads, range checks) runs on the few tiles that need it - left to one
        // body the compiler if-converts both into selects on every score of every tile
        if (fl == 1 || ragged) tile_body(std::true_type{}, t0, fl, buf);
        else tile_body(std::false_type{}, t0, fl, buf);

        if (nk < nkt) store_tile(buf ^ 1);                    // nobody reads that buffer: its readers passed the last barrier
        __syncthreads();
        kt = nk;
        buf ^= 1;
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
        float lt = l[r] + __shfl_xor(l[r], 16);
        lt += __shfl_xor(lt, 32);
        const float inv = 1.0f / lt;
        if (pos0 + r < S) {
            T* dst = Out + (((int64_t)b * S + pos0 + r) * H + g * HP + li) * D + 4 * q;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt)
                *reinterpret_cast<u32x2*>(dst + 16 * dt) =
                    u32x2{pack2<T>(o[r][dt][0] * inv, o[r][dt][1] * inv), pack2<T>(o[r][dt][2] * inv, o[r][dt][3] * inv)};
        }
    }
}

void prefill_attention_tiles(int64_t* q_block, int64_t* k_tile) {
    *q_block = kPfR * 8;
    *k_tile = kPfKeys;
}

template <typename T>
static int launch_prefill_attention(const void* Q, const void* Kc, const void* Vc, const float* mask, const uint8_t* flags, void* Out,
                                    int64_t B, int64_t S, int64_t Tkv, int64_t H, int64_t G, int64_t cap, int64_t ldm, hipStream_t st) {
    constexpr int NWV = 8, QB = kPfR * NWV;
    const int nqb = (int)((S + QB - 1) / QB), nkt = (int)((Tkv + kPfKeys - 1) / kPfKeys);
    static bool attr_set = false;                             // > 64 KB of dynamic LDS needs the opt-in, once per process
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_attention_kernel<T, NWV>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kPfLds) != hipSuccess)
            return (int)hipGetLastError();
        attr_set = true;
    }
    prefill_attention_kernel<T, NWV><<<(unsigned)(nqb * B * G), NWV * 64, kPfLds, st>>>(
        (const T*)Q, (const T*)Kc, (const T*)Vc, mask, flags, (T*)Out, (int)S, (int)Tkv, (int)H, (int)G, (int)cap, nqb, nkt, ldm,
        sqrtf(128.0f));
    return finish_launch();
}

int prefill_attention(int dtype, const void* Q, const void* Kc, const void* Vc, const float* mask, const uint8_t* flags, void* Out,
                      int64_t B, int64_t S, int64_t Tkv, int64_t H, int64_t G, int64_t cap, int64_t ldm, hipStream_t st) {
    if (dtype == QL_DTYPE_F16) return launch_prefill_attention<f16>(Q, Kc, Vc, mask, flags, Out, B, S, Tkv, H, G, cap, ldm, st);
    if (dtype == QL_DTYPE_BF16) return launch_prefill_attention<__bf16>(Q, Kc, Vc, mask, flags, Out, B, S, Tkv, H, G, cap, ldm, st);
    return QL_ERR_UNSUPPORTED;
}

}  // namespace ql
