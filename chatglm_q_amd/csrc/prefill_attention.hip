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
// once (global -> registers -> LDS; two K buffers, three V buffers).  Rows of a block differ by at most 2 NWV - 1 positions, so a
// causal mask cuts whole key tiles for the whole block.  Time is cut into slots of one barrier each and the two waves of a SIMD
// (w, w + 4) run one slot apart: second product of the previous tile + first product of this one in one slot, softmax in the
// next - one wave's MFMAs beside the other's VALU work (the slot loop at the end of the kernel has the buffer accounting).
//
// Mask: the reference's additive fp32 mask (B, S, T), applied exactly as the reference does - round(score) + mask in fp32 -
// so a row whose keys are ALL blocked comes out as the uniform average, like the reference's.  `tile_flags` (optional,
// built by the host from the same mask, fused_ops.attention_tile_flags) says per (sequence, query block, key tile):
//   0 = skip: every entry <= -1e9 and every row of the block has an entry >= -1e6 somewhere, so the tile's probabilities
//       are exactly 0 in fp32 whatever the scores are (|score| <= 65504);  2 = every entry is 0: no mask loads;  1 = load it.
// Rounding points (T = activation dtype): q / sqrt(d), the scores, the output - as the reference; P is rounded to T as
// exp(s - running max) and normalised after the second product (the reference rounds exp(s - max) / sum: the same
// relative error, not the same bits); the exp-sum is accumulated by the matrix pipe from those rounded values (an all-ones A
// operand), i.e. the normaliser is the sum of exactly what the second product multiplies.
#include "launch.h"
#include "ql_common.h"
#include <type_traits>

// developer switches (tools/ab/build_pf_variant.sh; always 0 / 8 in the library): QL_PF_ABLATE bits strip parts of the tile body to
// attribute time (results wrong): 1 no softmax arithmetic, 2 no second product, 4 no first product, 8 no K / V staging in the loop,
// 16 no barrier in the loop.  QL_PF_WAVES: waves per workgroup (8 or 4).
#ifndef QL_PF_ABLATE
#define QL_PF_ABLATE 0
#endif
#ifndef QL_PF_WAVES
#define QL_PF_WAVES 8
#endif

// QL_PF_STAMPS (developer build): workgroup 0's waves stamp s_memtime (shader-clock ticks) at the start of every slot, after its
// arithmetic, after its staging stores and behind its barrier - tools/pf_timeline.py prints where a slot's time goes
#ifdef QL_PF_STAMPS
__device__ unsigned long long ql_pf_stamps[8 * 160 * 4];
#define QL_PF_STAMP(k) do { if (blockIdx.x == 0 && lane == 0 && p < 160) ql_pf_stamps[(wv * 160 + p) * 4 + (k)] = (unsigned long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define QL_PF_STAMP(k) do { } while (0)
#endif

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
// bytes per key row in LDS.  Round 6 (profiles/r06_prefill_attention_pmc.txt): the padded pitch of 272 bytes was NOT conflict-free - ds_read_b128 is
// serviced in four groups of 16 NON-contiguous lanes ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS), which mix two k-quarters q of complementary
// row sets, and a pitch shifts a row by the same 16-byte slot a q step does: one 2-way conflict per group, 8 instead of 4 cycles per fragment read
// (SQ_LDS_BANK_CONFLICT 8.65 M cycles per launch, all of them the first product's: an ablation without it counts 0).  Now: 256-byte rows, the
// 16-byte chunk c of row r stored at chunk c ^ (r & 15) - XOR keeps every group's 16 lanes on 16 different slots for every j and q.
constexpr int kPfKP = 256;
constexpr int kPfVP = 288;                                    // bytes per value row in LDS (tr reads conflict-free)
constexpr int kPfR = 2;                                       // query positions per wave
constexpr int kPfMaxTiles = 2048;                             // key tiles per row of flags kept in LDS (T <= 131 072)
constexpr int kPfLds = kPfKeys * (2 * kPfKP + 3 * kPfVP) + kPfMaxTiles;   // two K tiles, three V tiles, the flags: 90 112 bytes

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
    unsigned char* const vimg = pf_smem + 2 * kPfKeys * KP;   // [3][64][VP]
    uint8_t* const fimg = pf_smem + kPfKeys * (2 * KP + 3 * VP);   // [kPfMaxTiles] this block's tile flags
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
    // The block's row of tile flags is copied into LDS once: read from global inside the loop, every look-up is a vector load whose
    // wait (vmcnt(0)) also drains the K / V staging loads in flight - the load latency of every tile would be exposed
    const bool have_flags = flags != nullptr;
    if (have_flags) {
        const uint8_t* frow_g = flags + ((int64_t)b * nqb + qblk) * nkt;
        for (int i = tid; i < nkt; i += NTH) fimg[i] = frow_g[i];
    }
    __syncthreads();
    const uint8_t* frow = fimg;
    auto next_tile = [&](int kt) {
        ++kt;
        while (have_flags && kt < nkt && frow[kt] == 0) ++kt;
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
    auto store_tile = [&](int kbuf, int vbuf) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = tid + NTH * i, row = c >> 4, col = c & 15;
            *reinterpret_cast<u32x4*>(kimg + (kbuf * kPfKeys + row) * KP + 16 * (col ^ (row & 15))) = kst[i];
            *reinterpret_cast<u32x4*>(vimg + (vbuf * kPfKeys + row) * VP + 16 * col) = vst[i];
        }
    };

    float m[R];                                               // running maximum (the same in the 4 lanes of a head)
    f32x4 lacc[R];                                            // exp-sum of head li (four copies), accumulated by the matrix pipe
    const u32 one2 = pack2<T>(1.0f, 1.0f);
    const u32x4 ones = u32x4{one2, one2, one2, one2};
    f32x4 o[R][8];                                            // O^T: lane (li, q) = head li, d = 16 dt + 4 q + e
    f32x4 s[R][4];                                            // S^T of the tile between the first product and its softmax
    u32x4 pb[R][2];                                           // exp(S^T - max) as the B operands of the second product's two 32-key steps
#pragma unroll
    for (int r = 0; r < R; ++r) {
        m[r] = -INFINITY;
        lacc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) o[r][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // S^T = K Q^T: lane (li, q) ends with the scores of head li against keys 16 pt + 4 q + e of the tile.  d chunk j outermost:
    // consecutive MFMAs go to 8 different accumulators (a dependent 16 x 16 x 32 pair costs the pipe's full latency)
    auto first_product = [&](const int kbuf) {
        const unsigned char* kb = kimg + kbuf * kPfKeys * KP;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int r = 0; r < R; ++r) s[r][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < ((QL_PF_ABLATE & 4) ? 0 : 4); ++j)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(kb + (16 * pt + li) * KP + 16 * ((4 * j + q) ^ li));
#pragma unroll
                for (int r = 0; r < R; ++r) s[r][pt] = PfMma<T>::mma(kf, qf[r][j], s[r][pt]);
            }
    };

    // mask, running maximum, exponentials (rounded to T into pb), exp-sum, rescale of O.  Two copies: the general one (mask loads,
    // range checks) runs on the few tiles that need it - left to one body the compiler if-converts both into selects on every score
    auto softmax = [&](auto general_tag, const int t0, const int fl) {
        constexpr bool GEN = decltype(general_tag)::value;
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
        if constexpr ((QL_PF_ABLATE & 1) != 0) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    pb[r][c] = u32x4{pack2<T>(s[r][2 * c][0], s[r][2 * c][1]), pack2<T>(s[r][2 * c][2], s[r][2 * c][3]),
                                     pack2<T>(s[r][2 * c + 1][0], s[r][2 * c + 1][1]), pack2<T>(s[r][2 * c + 1][2], s[r][2 * c + 1][3])};
        }
        typedef T T2 __attribute__((ext_vector_type(2)));
        typedef float F2 __attribute__((ext_vector_type(2)));
        constexpr float kLog2e = 1.4426950408889634f;
#pragma unroll
        for (int r = 0; r < ((QL_PF_ABLATE & 1) ? 0 : R); ++r) {
            // scores rounded to T two at a time (v_cvt_pk_*); they stay packed: the arithmetic below reads the halves in place
            T2 xr[4][2];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                xr[pt][0] = __builtin_convertvector(F2{s[r][pt][0], s[r][pt][1]}, T2);
                xr[pt][1] = __builtin_convertvector(F2{s[r][pt][2], s[r][pt][3]}, T2);
            }
            float mx;
            if constexpr (GEN) {
                mx = -INFINITY;
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = (float)xr[pt][e >> 1][e & 1] + mv[r][pt][e];
                        if (t0 + 16 * pt + 4 * q + e >= Tkv) x = -INFINITY;
                        s[r][pt][e] = x;
                        mx = fmaxf(mx, x);
                    }
            } else if constexpr (Act<T>::code == QL_DTYPE_F16) {
                T2 pm = xr[0][0];                             // the rounded scores are exact fp16 values: packed maximum
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int h = 0; h < 2; ++h) pm = __builtin_elementwise_max(pm, xr[pt][h]);
                mx = fmaxf((float)pm[0], (float)pm[1]);
            } else {
                mx = -INFINITY;
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) mx = fmaxf(mx, (float)xr[pt][e >> 1][e & 1]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(m[r], mx);                 // finite: key t0 is in range and its entry is finite
            const float alpha = __expf(m[r] - mn);            // first tile: exp(-inf) = 0
            m[r] = mn;
            const float mnl = mn * kLog2e;
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                float ev[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (GEN)                        // masked values reach -1e10: the difference first, exactly
                        ev[e] = __builtin_amdgcn_exp2f((s[r][pt][e] - mn) * kLog2e);
                    else                                      // |x|, |mn| <= 65504: one fma, error <= 6e-8 |mn| in the exponent
                        ev[e] = __builtin_amdgcn_exp2f(__builtin_fmaf((float)xr[pt][e >> 1][e & 1], kLog2e, -mnl));
                }
                pb[r][pt >> 1][2 * (pt & 1)] = __builtin_bit_cast(u32, __builtin_convertvector(F2{ev[0], ev[1]}, T2));
                pb[r][pt >> 1][2 * (pt & 1) + 1] = __builtin_bit_cast(u32, __builtin_convertvector(F2{ev[2], ev[3]}, T2));
            }
            if (!__all(alpha == 1.0f)) {                      // wave-uniform: the maximum settles after the first tiles
#pragma unroll
                for (int e = 0; e < 4; ++e) lacc[r][e] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[r][dt][e] *= alpha;
            }
        }
    };

    // O^T += V^T P^T, 32 keys per MFMA step
    auto second_product = [&](const int vbuf) {
        const unsigned char* vb = vimg + vbuf * kPfKeys * VP;
#pragma unroll
        for (int c = 0; c < ((QL_PF_ABLATE & 2) ? 0 : 2); ++c) {
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                const unsigned char* p0 = vb + (32 * c + 4 * q + (li >> 2)) * VP + 2 * (16 * dt + 4 * (li & 3));
                const pf_s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (pf_s16x4 __attribute__((address_space(3)))*)(const_cast<unsigned char*>(p0)));
                const pf_s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (pf_s16x4 __attribute__((address_space(3)))*)(const_cast<unsigned char*>(p0 + 16 * VP)));
                const u32x4 a = __builtin_bit_cast(u32x4, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int r = 0; r < R; ++r) o[r][dt] = PfMma<T>::mma(a, pb[r][c], o[r][dt]);
            }
            // the exp-sum on the matrix pipe too: an all-ones A operand gives every lane of head li the sum over the step's 32 keys
            // (of the ROUNDED probabilities - what the product above multiplies), with no cross-lane reduction and no VALU adds
#pragma unroll
            for (int r = 0; r < R; ++r) lacc[r] = PfMma<T>::mma(ones, pb[r][c], lacc[r]);
        }
        if constexpr ((QL_PF_ABLATE & 2) != 0) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) o[r][c][0] += __builtin_bit_cast(float, pb[r][c][0] ^ pb[r][c][1] ^ pb[r][c][2] ^ pb[r][c][3]);
        }
    };

    // Ping-pong over time slots (one block barrier per slot): a tile is two slots of work for a wave - X = the second product of
    // the PREVIOUS tile + the first product of this one (matrix pipe), Y = this tile's softmax (VALU) - and the waves w and w + 4,
    // which share a SIMD, run one slot apart: while one is in X the other is in Y.  Group A (waves 0..3) runs X(i) in slot 2i and
    // Y(i) in slot 2i + 1, group B one slot later.  Tile i + 1 is requested at the start of slot 2i and stored at the end of slot
    // 2i + 1: its K buffer (two of them) was last read in slot 2i - 1 (B's X(i - 1)), its V buffer (three) in slot 2i - 1 (B's
    // second product of tile i - 2); the first reader is A in slot 2i + 2, behind the slot's barrier.
    int NT = 0;                                               // tiles this block processes (block-uniform)
    for (int kt = next_tile(-1); kt < nkt; kt = next_tile(kt)) ++NT;
    const int grp = wv >= NWV / 2 ? 1 : 0;
    int kt_stage = next_tile(-1), ord_stage = 0;
    if (NT > 0) {
        load_tile(kt_stage);
        store_tile(0, 0);
        kt_stage = next_tile(kt_stage);
        ord_stage = 1;
    }
    __syncthreads();
    int kt_x = next_tile(-1), kt_soft = 0;
    for (int p = 0; p < 2 * NT + 2; ++p) {
        const bool staging = ord_stage < NT && !(QL_PF_ABLATE & 8);
        QL_PF_STAMP(0);
        if (!(p & 1) && staging) load_tile(kt_stage);         // in flight under two slots of arithmetic
        const int pp = p - grp;
        if (pp >= 0 && pp <= 2 * NT) {
            const int ii = pp >> 1;
            if (!(pp & 1)) {
                if (ii > 0) second_product((ii - 1) % 3);
                if (ii < NT) {
                    first_product(ii & 1);
                    kt_soft = kt_x;
                    kt_x = next_tile(kt_x);
                }
            } else if (ii < NT) {
                const int t0 = kt_soft * kPfKeys;
                const int fl = !mask ? 2 : have_flags ? (int)frow[kt_soft] : 1;   // no mask at all: every tile is 'all zero'
                if (fl == 1 || t0 + kPfKeys > Tkv) softmax(std::true_type{}, t0, fl);
                else softmax(std::false_type{}, t0, fl);
            }
        }
        QL_PF_STAMP(1);
        if ((p & 1) && ord_stage < NT) {
            if (!(QL_PF_ABLATE & 8)) store_tile(ord_stage & 1, ord_stage % 3);
            kt_stage = next_tile(kt_stage);
            ++ord_stage;
        }
        QL_PF_STAMP(2);
        if (!(QL_PF_ABLATE & 16)) __syncthreads();
        QL_PF_STAMP(3);
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float inv = 1.0f / lacc[r][0];
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
    *q_block = kPfR * QL_PF_WAVES;
    *k_tile = kPfKeys;
}

template <typename T>
static int launch_prefill_attention(const void* Q, const void* Kc, const void* Vc, const float* mask, const uint8_t* flags, void* Out,
                                    int64_t B, int64_t S, int64_t Tkv, int64_t H, int64_t G, int64_t cap, int64_t ldm, hipStream_t st) {
    constexpr int NWV = QL_PF_WAVES, QB = kPfR * NWV;
    const int nqb = (int)((S + QB - 1) / QB), nkt = (int)((Tkv + kPfKeys - 1) / kPfKeys);
    if (nkt > kPfMaxTiles) return QL_ERR_UNSUPPORTED;
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

#ifdef QL_PF_STAMPS
extern "C" int qlinear_pf_stamps_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ql_pf_stamps), sizeof(unsigned long long) * 8 * 160 * 4);
}
#endif
