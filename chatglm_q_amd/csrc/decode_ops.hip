// Fused neighbours of the QLinear calls inside one decode step, gfx950.
//
// Once the 113 QLinear launches of a token run near the HBM rate, the ~50 tiny elementwise / reduction
// launches per layer that torch issues around them dominate a decode step (each costs the ~1.5 us
// dependent-launch boundary).  These four kernels replace them one-for-one with the SAME rounding
// sequence the model graph defines (chatglm_q/model.py lines cited per kernel; every intermediate the
// reference materialises in the activation dtype is rounded to it here too):
//   rmsnorm            chatglm_q/model.py:62-73
//   rope_kv_write      chatglm_q/model.py:139-155 (split, rotary on the first half of each head, cache write)
//   decode_attention   chatglm_q/model.py:157-175 (q / sqrt(d), q k^T, + mask, fp32 softmax, p v)
//   silu_mul           chatglm_q/model.py:200-201
// All are HBM/latency-bound row kernels: one 256-thread block per row / per head, 16-byte accesses.
#include "launch.h"
#include "ql_common.h"

// developer ablation switches for tools/microbench/attn_ablate.hip (always 0 in the library)
#ifndef QL_ATT_ABLATE
#define QL_ATT_ABLATE 0
#endif

namespace ql {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float wave_max(float v) { return wave_max_dpp(v); }
__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return r;
}

// 8 consecutive activations <-> fp32 (one 16-byte access for the 16-bit dtypes)
template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            T lo, hi;
            const uint16_t l16 = (uint16_t)(r[e] & 0xFFFFu), h16 = (uint16_t)(r[e] >> 16);
            __builtin_memcpy(&lo, &l16, 2);
            __builtin_memcpy(&hi, &h16, 2);
            v[2 * e] = (float)lo;
            v[2 * e + 1] = (float)hi;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = Act<T>::load(p + e);
    }
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        u32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const T lo = (T)v[2 * e], hi = (T)v[2 * e + 1];
            uint16_t l16, h16;
            __builtin_memcpy(&l16, &lo, 2);
            __builtin_memcpy(&h16, &hi, 2);
            r[e] = (u32)l16 | ((u32)h16 << 16);
        }
        *reinterpret_cast<u32x4*>(p) = r;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) Act<T>::store(p + e, v[e]);
    }
}

// ---------------------------------------------------------------------------------------------
// out = round(round(x * rsqrt(mean(x^2) + eps)) * w)        (+ optional residual stream update)
//   FUSE_ADD: x = round(x_in + delta) is written back to Hout first (the block's residual add,
//   chatglm_q/model.py:243,245) and normalised in the same pass: one launch instead of two.
// Row kept in registers (VPT x 8 values per thread): one read of the row, one barrier pair.
// ---------------------------------------------------------------------------------------------
// QUANT (round 3): the normalised row - already in registers - is ALSO emitted as int8 + one fp32 scale per row, exactly what
// act_quant_rows_kernel (w8a8.hip) would produce from the rounded output row (quantize_int8, chatglm_q/int8/quantizer.py:11-19:
// scale = max|x| / 127 clamped to 1e-10, true division, round half to even): the int8-activation GEMM that follows needs no
// quantiser launch.  Out may be null then (nobody else reads the 16-bit row).
__device__ __forceinline__ u32x2 quant8_pack(const float (&y)[8], float s) {
    u32x2 w = {0u, 0u};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float q = rintf(y[e] / s);
        q = fminf(fmaxf(q, -127.f), 127.f);
        w[e >> 2] |= ((u32)(int)q & 0xFFu) << (8 * (e & 3));
    }
    return w;
}
template <typename T, int VPT, bool FUSE_ADD, bool QUANT = false>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ X, const T* __restrict__ Delta,
                                                      const T* __restrict__ Wt, T* __restrict__ Hout, T* __restrict__ Out,
                                                      int dim, int64_t ldx, int64_t ldo, float eps,
                                                      int8_t* __restrict__ Aq = nullptr, float* __restrict__ a_scale = nullptr) {
    __shared__ float red[4];
    const T* x = X + (int64_t)blockIdx.x * ldx;
    T* o = Out + (int64_t)blockIdx.x * ldo;
    float v[VPT][8];
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int i = (threadIdx.x + u * 256) * 8;
        if (i < dim) {
            load8<T>(x + i, v[u]);
            if constexpr (FUSE_ADD) {
                float d[8];
                load8<T>(Delta + (int64_t)blockIdx.x * ldx + i, d);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][e] = Act<T>::round(v[u][e] + d[e]);
                store8<T>(Hout + (int64_t)blockIdx.x * ldx + i, v[u]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(v[u][e], v[u][e], ss);
        }
    }
    // the norm weights do not depend on the row: requested in front of the block reduction, not as a second round trip behind it (few-row
    // calls are a latency chain: 8 rows 5.1 -> 4.5 us under rocprof)
    u32x4 wraw[VPT];
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int i = (threadIdx.x + u * 256) * 8;
        if constexpr (sizeof(T) == 2) wraw[u] = *reinterpret_cast<const u32x4*>(Wt + (i < dim ? i : 0));
    }
    ss = block_sum_256(ss, red);
    const float r = rsqrtf(ss / (float)dim + eps);
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int i = (threadIdx.x + u * 256) * 8;
        if (i < dim) {
            float w[8], y[8];
            if constexpr (sizeof(T) == 2) unpack8<T>(wraw[u], w);
            else load8<T>(Wt + i, w);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = v[u][e] * r;
                // the reference rounds the fp32 product to the activation dtype (chatglm_q/model.py:139, two roundings); in this
                // instantiation hipcc otherwise folds multiply + conversion into v_fma_mixlo_f16 - ONE rounding of the exact
                // product, 1 ulp off in about one value of 40 000 (seen against the plain kernel, which keeps v_pk_mul_f32 + cvt)
                if constexpr (QUANT) asm volatile("" : "+v"(t));
                y[e] = Act<T>::round(t) * w[e];
            }
            if (!QUANT || Out) store8<T>(o + i, y);
            if constexpr (QUANT) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[u][e] = Act<T>::round(y[e]);         // the value the 16-bit row holds: what the quantiser would read
                    mx = fmaxf(mx, fabsf(v[u][e]));
                }
            }
        }
    }
    if constexpr (QUANT) {
        mx = block_max_256(mx, red);
        const float s = fmaxf(mx / 127.0f, 1e-10f);
        if (threadIdx.x == 0) a_scale[blockIdx.x] = s;
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int i = (threadIdx.x + u * 256) * 8;
            if (i < dim) *reinterpret_cast<u32x2*>(Aq + (int64_t)blockIdx.x * dim + i) = quant8_pack(v[u], s);
        }
    }
}

__device__ __forceinline__ int64_t clamp_pos(int64_t p, int capacity) { return p < 0 ? 0 : (p <= capacity ? p : capacity); }

// ---------------------------------------------------------------------------------------------
// fused_qkv (rows, (H + 2G) D): rotate q and k pairs (re, im) by (cos, sin) of the row's position, write
// q (rows, H D); k, v into the caches at row `write_index[s]`.
//   table: (max_pos, D/2, 2) in the activation dtype (second half of the pairs = (1, 0) pass-through)
// rows = B * S, row = b * S + s.  caches: (B, capacity, G, D).  One thread = 4 pairs = 8 values.
// ---------------------------------------------------------------------------------------------
// (x0 + i x1) (c + i s), one explicit contraction so that every kernel rotates bit-identically
__device__ __forceinline__ void rope_pair(float x0, float x1, float c, float s, float& y0, float& y1) {
    y0 = __builtin_fmaf(x0, c, -(x1 * s));
    y1 = __builtin_fmaf(x0, s, x1 * c);
}

template <typename T>
__global__ __launch_bounds__(256) void rope_kv_write_kernel(const T* __restrict__ QKV, const T* __restrict__ table,
                                                            const int64_t* __restrict__ pos, const int64_t* __restrict__ widx,
                                                            T* __restrict__ Qout, T* __restrict__ Kc, T* __restrict__ Vc,
                                                            int S, int H, int G, int D, int capacity, int64_t ldqkv) {
    const int row = blockIdx.x, b = row / S, s = row - b * S;
    const T* in = QKV + (int64_t)row * ldqkv;
    // Bounds (the ABI carries no table length): positions are 1-based counts of the unmasked tokens up to and including
    // the row (chatglm_q/model.py:307-308), so a valid position is at most its cache row + 1: positions are clamped to
    // [0, capacity] - the table holds capacity + 1 rows (checked by the host wrappers) - and a row index outside the
    // cache writes nothing.
    const T* cs = table + clamp_pos(pos[row], capacity) * D;     // D/2 pairs x 2 = D values
    const int64_t wrow = widx[s];
    const bool in_cache = wrow >= 0 && wrow < capacity;
    const int per_head = D / 8;                              // 8-value units per head
    const int n_units = (H + 2 * G) * per_head;
    for (int i = threadIdx.x; i < n_units; i += 256) {
        const int head = i / per_head, u = i - head * per_head;
        float x[8];
        load8<T>(in + head * D + u * 8, x);
        T* dst;
        if (head < H)
            dst = Qout + (int64_t)row * (H * D) + head * D;
        else if (head < H + G)
            dst = Kc + (((int64_t)b * capacity + wrow) * G + (head - H)) * D;
        else
            dst = Vc + (((int64_t)b * capacity + wrow) * G + (head - H - G)) * D;
        if (head >= H && !in_cache) continue;
        if (head < H + G) {
            float c[8], y[8];
            load8<T>(cs + u * 8, c);                         // (cos, sin) x 4 pairs
#pragma unroll
            for (int p = 0; p < 4; ++p) rope_pair(x[2 * p], x[2 * p + 1], c[2 * p], c[2 * p + 1], y[2 * p], y[2 * p + 1]);
            store8<T>(dst + u * 8, y);
        } else {
            store8<T>(dst + u * 8, x);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// one query position per block: block = (b, head).  q (B, H, D) [S == 1]; caches (B, capacity, G, D);
// mask (B, capacity) additive fp32.  out (B, H D).
//   qs = round(q / sqrt(D));  s_t = round(sum_d qs_d k_td) + mask_t;  p = round(softmax_fp32(s));
//   out_d = round(sum_t p_t v_td)
// QK: thread = position, the key row read with 16-byte loads, all issued before the first use.
// PV: thread = (8-wide d chunk, position slice): 16-byte loads of V, slices combined through LDS.
// ---------------------------------------------------------------------------------------------
template <typename T, int D, bool ROPE>
__global__ __launch_bounds__(256) void decode_attention_kernel(const T* __restrict__ Q, T* Kc, T* Vc,
                                                               const float* __restrict__ mask, T* __restrict__ Out, int H,
                                                               int G, int cap_full, float sqrt_d,
                                                               const T* __restrict__ table, const int64_t* __restrict__ pos,
                                                               const int64_t* __restrict__ widx, int64_t ldq,
                                                               int window, float* __restrict__ split_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Long contexts (split_out != nullptr): blockIdx.y owns the positions [t_lo, t_lo + capacity) and leaves
    // (local max, local exp-sum, unnormalised output) for attention_combine_kernel - a decode step has only B * H
    // blocks otherwise, and one block per head walking thousands of positions leaves most of the chip idle.
    const int t_lo = split_out ? (int)blockIdx.y * window : 0;
    const int capacity = split_out ? (cap_full - t_lo < window ? cap_full - t_lo : window) : cap_full;
    static_assert(2 * D <= 256, "one thread per q pair, k pair and v value");
    constexpr int CH = D / 8;                                 // 8-wide chunks per head row
    constexpr int SL = 256 / CH;                              // position slices in the PV phase
    float* sc = reinterpret_cast<float*>(smem);               // capacity scores / probabilities
    float* qs = sc + capacity;                                // D scaled query values
    float* red = qs + D;                                      // 4
    float* part = red + 4;                                    // SL x D partial outputs
    float* knew = part + SL * D;                              // ROPE: this step's rotated key / value of the head's group
    float* vnew = knew + D;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int g = h / (H / G);
    const T* kb = Kc + (((int64_t)b * cap_full + t_lo) * G + g) * D;
    const T* vb = Vc + (((int64_t)b * cap_full + t_lo) * G + g) * D;
    const float* mk = mask + (int64_t)b * cap_full + t_lo;
    // Thread (c, r) owns 16-byte chunk c of rows r, r + RP, r + 2 RP, ... in BOTH phases (Q.K and P.V).  A decode step
    // gives this kernel ~32 blocks, so its time is a chain of global round trips, not bandwidth: for the 16-bit
    // dtypes the key chunks, value chunks and mask entries of the first NP rows per thread (256 positions) are
    // requested before anything else - the rotary phase and the softmax then run under those loads.
    constexpr int RP = 256 / CH;
    constexpr bool kPre = sizeof(T) == 2 && !(QL_ATT_ABLATE & 32);
    constexpr int NP = kPre ? 256 / RP : 1;
    const int c = threadIdx.x % CH, r = threadIdx.x / CH;
    u32x4 kr[NP], vr[NP];
    float mr[NP];
    if constexpr (kPre) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int t = r + RP * i, tc = t < capacity ? t : capacity - 1;
            kr[i] = *reinterpret_cast<const u32x4*>(kb + (int64_t)tc * G * D + c * 8);
            mr[i] = mk[tc];
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int t = r + RP * i, tc = t < capacity ? t : capacity - 1;
            vr[i] = *reinterpret_cast<const u32x4*>(vb + (int64_t)tc * G * D + c * 8);
        }
    }
    int wrow = -1;
    if constexpr (ROPE) {
        // Q is the fused projection row (H q heads | G k heads | G v heads); rotate q and k exactly as
        // rope_kv_write_kernel does (rounded to T), keep k / v of the new position in LDS - the group's other
        // blocks must not read that cache row while block (h % (H/G) == 0) writes it
        wrow = (int)widx[0] - t_lo;                          // window-local; outside [0, capacity) for other windows
        const T* row = Q + (int64_t)b * ldq;
        const T* cs = table + clamp_pos(pos[b], cap_full) * D;
        const int d = threadIdx.x;
        if (d < D) {
            const int p = d < D / 2 ? d : d - D / 2;
            const T* x = d < D / 2 ? row + h * D : row + (H + g) * D;
            const float x0 = (QL_ATT_ABLATE & 1) ? 0.5f : Act<T>::load(x + 2 * p), x1 = (QL_ATT_ABLATE & 1) ? 0.25f : Act<T>::load(x + 2 * p + 1);
            const float c0 = (QL_ATT_ABLATE & 1) ? 0.6f : Act<T>::load(cs + 2 * p), c1 = (QL_ATT_ABLATE & 1) ? 0.8f : Act<T>::load(cs + 2 * p + 1);
            float y0, y1;
            rope_pair(x0, x1, c0, c1, y0, y1);
            y0 = Act<T>::round(y0);
            y1 = Act<T>::round(y1);
            if (d < D / 2) {
                qs[2 * p] = Act<T>::round(y0 / sqrt_d);
                qs[2 * p + 1] = Act<T>::round(y1 / sqrt_d);
            } else {
                knew[2 * p] = y0;
                knew[2 * p + 1] = y1;
            }
        } else if (d < 2 * D) {
            vnew[d - D] = Act<T>::load(row + (H + G + g) * D + (d - D));
        }
        __syncthreads();
        if (h % (H / G) == 0 && d < 2 * D && wrow >= 0 && wrow < capacity) {
            const int64_t at = (((int64_t)b * cap_full + t_lo + wrow) * G + g) * D;
            if (d < D) Act<T>::store(Kc + at + d, knew[d]);
            else Act<T>::store(Vc + at + d - D, vnew[d - D]);
        }
    } else {
        const T* q = Q + ((int64_t)b * H + h) * D;
        for (int d = threadIdx.x; d < D; d += 256) qs[d] = Act<T>::round(Act<T>::load(q + d) / sqrt_d);
        __syncthreads();
    }
    // Q.K: lane = 16-byte chunk c of key row t (a row = CH adjacent lanes: coalesced 2*D-byte reads; with one
    // thread per POSITION every load instruction touched 64 different cache lines and the loads alone cost 2 us),
    // 4 rows in flight per thread, partial dots combined across the CH lanes by DPP
    float mx = -INFINITY;
    {
        float qv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = qs[c * 8 + e];
        auto chunk_dot = [&](int t, const float (&kv)[8], float mt) {
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(qv[e], kv[e], acc);
            acc = group_sum<CH>(acc);
            // ROPE (one-launch decode step): cache rows behind the row this step writes hold nothing of the sequence (the reference
            // appends at the end of its cache, chatglm_q/model.py:148-151) - hidden whatever the mask says, as the 16-heads-per-group
            // kernel hides them (the dispatch paths must compute one function)
            const float sv = Act<T>::round(acc) + ((ROPE && t > wrow) ? -1e30f : mt);
            mx = fmaxf(mx, sv);                               // every lane of the group holds the same score
            if (c == 0) sc[t] = sv;
        };
        // loads are unconditional (a load under `if (t == wrow)` makes hipcc wait for it on the spot, serialising
        // the round trips); the row being written by this step is replaced AFTER the load by the LDS copy
        auto load_chunk = [&](int t, float (&kv)[8]) {
            if (QL_ATT_ABLATE & 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) kv[e] = 0.01f * (float)(c + e + t);
            } else {
                load8<T>(kb + (int64_t)t * G * D + c * 8, kv);
            }
        };
        auto fix_new = [&](int t, float (&kv)[8]) {
            if (ROPE && t == wrow) {
#pragma unroll
                for (int e = 0; e < 8; ++e) kv[e] = knew[c * 8 + e];
            }
        };
        int t = (QL_ATT_ABLATE & 16) ? capacity : r;
        if constexpr (kPre) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int tp = r + RP * i;
                if (tp < capacity && !(QL_ATT_ABLATE & 16)) {
                    float k0[8];
                    unpack8<T>(kr[i], k0);
                    fix_new(tp, k0);
                    chunk_dot(tp, k0, mr[i]);
                }
            }
            if (!(QL_ATT_ABLATE & 16)) t = r + RP * NP;
        }
        for (; t + 3 * RP < capacity; t += 4 * RP) {
            float k0[8], k1[8], k2[8], k3[8];
            load_chunk(t, k0);
            load_chunk(t + RP, k1);
            load_chunk(t + 2 * RP, k2);
            load_chunk(t + 3 * RP, k3);
            const float m0 = mk[t], m1 = mk[t + RP], m2 = mk[t + 2 * RP], m3 = mk[t + 3 * RP];   // unconditional too
            fix_new(t, k0);
            fix_new(t + RP, k1);
            fix_new(t + 2 * RP, k2);
            fix_new(t + 3 * RP, k3);
            chunk_dot(t, k0, m0);
            chunk_dot(t + RP, k1, m1);
            chunk_dot(t + 2 * RP, k2, m2);
            chunk_dot(t + 3 * RP, k3, m3);
        }
        for (; t < capacity; t += RP) {
            float k0[8];
            load_chunk(t, k0);
            const float m0 = mk[t];
            fix_new(t, k0);
            chunk_dot(t, k0, m0);
        }
    }
    mx = (QL_ATT_ABLATE & 4) ? 1.0f : block_max_256(mx, red);
    float sum = 0.f;
    for (int t = threadIdx.x; t < capacity; t += 256) {
        const float e = __expf(sc[t] - mx);
        sc[t] = e;
        sum += e;
    }
    sum = (QL_ATT_ABLATE & 4) ? 100.f : block_sum_256(sum, red);
    const float inv = 1.0f / sum;
    if (!split_out)                                           // split mode keeps exp(s - local max): normalised by the combine
        for (int t = threadIdx.x; t < capacity; t += 256) sc[t] = Act<T>::round(sc[t] * inv);
    __syncthreads();
    float pw = 0.f;
    if (ROPE && wrow >= 0 && wrow < capacity) {               // block-uniform
        // the new position's value comes from LDS: take its probability out of the table (0 x the cache row = 0)
        pw = sc[wrow];
        __syncthreads();
        if (threadIdx.x == 0) sc[wrow] = 0.f;
        __syncthreads();
    }
    const int sl = r;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    int t = (QL_ATT_ABLATE & 8) ? capacity : sl;
    if constexpr (kPre) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int tp = sl + SL * i;
            const float p = (tp < capacity && !(QL_ATT_ABLATE & 8)) ? sc[tp] : 0.f;
            float v[8];
            unpack8<T>(vr[i], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(p, v[e], o[e]);
        }
        if (!(QL_ATT_ABLATE & 8)) t = sl + SL * NP;
    }
    for (; t + 3 * SL < capacity; t += 4 * SL) {              // 4 independent 16-byte loads in flight
        float v0[8], v1[8], v2[8], v3[8];
        load8<T>(vb + (int64_t)t * G * D + c * 8, v0);
        load8<T>(vb + (int64_t)(t + SL) * G * D + c * 8, v1);
        load8<T>(vb + (int64_t)(t + 2 * SL) * G * D + c * 8, v2);
        load8<T>(vb + (int64_t)(t + 3 * SL) * G * D + c * 8, v3);
        const float p0 = sc[t], p1 = sc[t + SL], p2 = sc[t + 2 * SL], p3 = sc[t + 3 * SL];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o[e] = __builtin_fmaf(p3, v3[e], __builtin_fmaf(p2, v2[e], __builtin_fmaf(p1, v1[e], __builtin_fmaf(p0, v0[e], o[e]))));
    }
    for (; t < capacity; t += SL) {
        float v0[8];
        load8<T>(vb + (int64_t)t * G * D + c * 8, v0);
        const float p0 = sc[t];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(p0, v0[e], o[e]);
    }
    if (ROPE && sl == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(pw, vnew[c * 8 + e], o[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[sl * D + c * 8 + e] = o[e];
    __syncthreads();
    float* so = split_out ? split_out + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * (D + 2) : nullptr;
    if (so && threadIdx.x == 0) {
        so[0] = mx;
        so[1] = sum;
    }
    for (int d = threadIdx.x; d < D; d += 256) {
        float tot = 0.f;
        for (int s2 = 0; s2 < SL; ++s2) tot += part[s2 * D + d];
        if (so) so[2 + d] = tot;
        else Act<T>::store(Out + ((int64_t)b * H + h) * D + d, tot);
    }
}

// ---------------------------------------------------------------------------------------------
// Grouped-query decode attention on the matrix cores (16-bit dtypes, D = 128, 16 query heads per key/value group -
// ChatGLM2's geometry): one block = (sequence, key/value GROUP, 256-position window), so a cache row is read once for
// the group's 16 heads instead of once per head, and the 16 heads are the N dimension of 16x16 MFMA tiles:
//   S^T (positions x heads) = K (positions x d) Q^T          v_mfma_f32_16x16x32: A = key rows straight from the cache
//   O^T (d x heads)         = V^T (d x positions) P^T        v_mfma_f32_16x16x16: B = exp(S^T - max) in the C layout
//                                                            of the first product, A = ds_read_b64_tr_b16 of the
//                                                            wave's row-major value rows in LDS (pitch 288 B)
// Wave w owns positions [64 w, 64 w + 64) of the window with its own running maximum; the four waves' partial
// (max, exp-sum, O) are merged through LDS after ONE barrier - the same arithmetic attention_combine_kernel applies
// across windows.  Rounding points: rotary outputs, q / sqrt(d), the scores and the output are rounded to T as in
// decode_attention_kernel; P is rounded to T as exp(s - wave max) (the reference rounds exp(s - max) / sum: the same
// relative error, not the same bits), the exp-sum uses the unrounded values.
// Window fits one block (capacity <= 256): Out is written directly.  Otherwise split_out receives (max, sum, O) per
// head and window for attention_combine_kernel.  The step's new key / value row is taken from LDS (only this block
// touches the group's cache rows, so the row is written by the block whose window holds it).
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2v __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// Workgroup p of np prefetchers.  The attention launch's workgroup (att_blocks + p) runs on XCD (att_blocks + p) % 8 and
// consumer workgroup g of the next launch on XCD g % 8: p takes the consumers g = q, q + np', ... with
// q = 8 (p / 8) + (att_blocks + p) % 8 (np' = np rounded down to a multiple of 8), i.e. only ones of its own XCD, and
// reads their bytes once with plain (cacheable) 16-byte loads.  Measured on attention + o_proj (int4g32 4096 x 4096,
// capacity 256; tools/attention_prefetch.py): 10.8 us without; everything prefetched from the start 10.9 (the
// attention's own loads queue behind 9.4 MB); started ~1.7 us late and capped at ~7 MB 10.3 - against ROUND 2's attention kernel.
// Round 6 re-measured the cap on the whole decode step (tools/decode_ab.py, profiles/r06_prefetch_budget_ab.txt) against round 5's faster
// attention launch: the 7 MB no longer fit its shadow - the launch ended when the PREFETCHERS did - and cost the step 3 % (871 - 875 tok/s
// against 888 - 891 with no prefetch at all); 1 / 2 / 3 / 4 MB: 892 / 895 / 902 / 904, 5 MB and more 875: a cliff between 4 and 5 MB.  The
// cliff is the START DELAY's: with the prefetchers starting at once (the delay protected round 2's attention loads; round 5's kernel does not
// care) 5 / 6 / 7 / 9.4 MB run 898 / 901 / 904 / 899 - no cliff on either side.  7 MB from the start.
#ifndef QL_PF_BUDGET_KB
#define QL_PF_BUDGET_KB (7 << 10)
#endif
#ifndef QL_PF_SLEEP
#define QL_PF_SLEEP 0
#endif
#ifndef QL_PF_BLOCKS
#define QL_PF_BLOCKS 256        // prefetch workgroups at most (about one per CU)
#endif
constexpr int64_t kPrefetchBudget = (int64_t)QL_PF_BUDGET_KB << 10;   // bytes that fit in the attention's shadow
__device__ __forceinline__ void prefetch_blocks(const Prefetch& pf, int p, int np, int att_blocks) {
    const int np8 = np & ~7;
    if (p >= np8) return;
    const int q = 8 * (p >> 3) + ((att_blocks + p) & 7);
    const int64_t per_block = pf.block_bytes[0] + pf.block_bytes[1];
    const int64_t fit = kPrefetchBudget / (per_block > 0 ? per_block : 1);
    const int gmax = fit < pf.blocks ? (int)fit : pf.blocks;
    // let the attention workgroups' own requests reach the memory system first
    for (int i = 0; i < QL_PF_SLEEP; ++i) __builtin_amdgcn_s_sleep(8);
    u32 x = 0;
    for (int g = q; g < gmax; g += np8) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t nb = pf.block_bytes[r];
            if (nb == 0) continue;
            const char* src = pf.base[r] + (int64_t)g * nb;
            int64_t off = (int64_t)threadIdx.x * 16;
            const int64_t step = (int64_t)blockDim.x * 16;
            for (; off + 3 * step + 16 <= nb; off += 4 * step) {          // 4 independent loads in flight per thread
                const u32x4 a = *reinterpret_cast<const u32x4*>(src + off);
                const u32x4 b = *reinterpret_cast<const u32x4*>(src + off + step);
                const u32x4 c = *reinterpret_cast<const u32x4*>(src + off + 2 * step);
                const u32x4 d = *reinterpret_cast<const u32x4*>(src + off + 3 * step);
                x ^= a[0] ^ b[0] ^ c[0] ^ d[0];
            }
            for (; off + 16 <= nb; off += step) x ^= (*reinterpret_cast<const u32x4*>(src + off))[0];
        }
    }
    asm volatile("" ::"v"(x));                                // the loads are the point: keep them
}

// QL_ATT_STAMPS (developer build: tools/ab/build_variant.sh attstamps decode_ops.hip -DQL_ATT_STAMPS, tools/attention_timeline.py): lane 0 of
// the first and last wave of attention workgroup 0 records the 100 MHz wall clock at the stations of the kernel's dependent chain
// (QL_ATT_C: the shader clock counter, at entry and end: the clock the chain ran at).
#ifdef QL_ATT_STAMPS
__device__ unsigned long long ql_att_stamps[2 * 16];
#define QL_ATT_T(i, ...)                                                                                   \
    do {                                                                                                   \
        asm volatile("" ::__VA_ARGS__);                                                                    \
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wv == 0 || wv == NWV - 1))                 \
            ql_att_stamps[(wv == 0 ? 0 : 16) + (i)] = __builtin_amdgcn_s_memrealtime();                    \
    } while (0)
#define QL_ATT_C(i)                                                                                        \
    do {                                                                                                   \
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wv == 0 || wv == NWV - 1))                 \
            ql_att_stamps[(wv == 0 ? 0 : 16) + (i)] = __builtin_amdgcn_s_memtime();                        \
    } while (0)
#else
#define QL_ATT_T(i, ...)
#define QL_ATT_C(i)
#endif

template <typename T> struct AttMma;
template <> struct AttMma<f16> {
    typedef _Float16 v8 __attribute__((ext_vector_type(8)));
    typedef _Float16 v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 qk(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 pv(u32x2v a, u32x2v b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(v4, a), __builtin_bit_cast(v4, b), c, 0, 0, 0);
    }
};
template <> struct AttMma<__bf16> {
    typedef __bf16 v8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x4 qk(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 pv(u32x2v a, u32x2v b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    }
};
#ifdef QL_DEV_TUNING                                 // round 2's kernel: the A/B partner of decode_attention_group_kernel below (QLINEAR_ATTENTION_R2=1)
template <typename T, int NWV>
__global__ __launch_bounds__(NWV * 64) void decode_attention_mfma_kernel(const T* __restrict__ QKV, const int64_t* __restrict__ pos,
                                                                         const int64_t* __restrict__ widx, T* Kc, T* Vc,
                                                                         int hg, int att_blocks, int cap_full, int ldq32,
                                                                         const T* __restrict__ table,
                                                                         const float* __restrict__ mask, float sqrt_d,
                                                                         T* __restrict__ Out, float* __restrict__ split_out,
                                                                         Prefetch next) {
    const int H = hg & 0xFFFF, G = hg >> 16;                  // packed: the preloaded argument dwords are all taken
    // Workgroups past the attention ones (blockIdx.y == 0 only) warm the caches for the next launch: see Prefetch.
    if ((int)blockIdx.x >= att_blocks) {
        if (blockIdx.y == 0) prefetch_blocks(next, (int)blockIdx.x - att_blocks, (int)gridDim.x - att_blocks, att_blocks);
        return;
    }
    // Argument order: the leading 14 dwords are preloaded into SGPRs at wave launch (Makefile); they are what the
    // query / key / value loads need, so those are in flight before the scalar loads of the rest (and of pos / widx,
    // which live in device memory) have returned.  The rotary table row depends on pos: it is requested last.
    // NWV waves share the 256-position window: 64 (4 waves) or 32 (8 waves: half the dependent chain per wave)
    // positions each.
    static_assert(sizeof(T) == 2, "16-bit dtypes");
    static_assert(NWV == 4 || NWV == 8, "waves per block");
    constexpr int D = 128, HP = 16, WIN = 256, NTH = NWV * 64;
    constexpr int PW = WIN / NWV;                             // positions per wave
    constexpr int PT = PW / 16;                               // 16-position tiles per wave
    constexpr int NH = PW / 32;                               // 32-row value images per wave
    constexpr int QPT = HP * 64 / NTH;                        // query pairs rotated per thread
    constexpr int VP = 288;                                   // bytes per value row in LDS: rows 8 banks apart (tr reads conflict-free)
    constexpr int QP = 272;                                   // bytes per query head in LDS
    constexpr int OP = 132;                                   // floats per head of a wave's partial output
    __shared__ __attribute__((aligned(16))) unsigned char vimg[NWV][32 * VP];   // per wave: 32 value rows; later its partial O
    __shared__ __attribute__((aligned(16))) unsigned char qs[HP * QP];
    __shared__ __attribute__((aligned(16))) T knew[D];
    __shared__ __attribute__((aligned(16))) T vnew[D];
    __shared__ float mw[NWV][HP], lw[NWV][HP];
    static_assert(HP * OP * 4 <= 32 * VP, "partial output fits the wave's value image");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, q = lane >> 4;
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int h0 = g * HP;
    const int t_lo = (int)blockIdx.y * WIN;
    const int wlen = cap_full - t_lo < WIN ? cap_full - t_lo : WIN;
    const int p0 = wv * PW;                                   // the wave's first window-local position
    const int64_t pitch = (int64_t)G * D;
    const T* kb = Kc + (((int64_t)b * cap_full + t_lo) * G + g) * D;
    const T* vb = Vc + (((int64_t)b * cap_full + t_lo) * G + g) * D;
    const T* row = QKV + (int64_t)b * ldq32;

    // every load of the block up front: this step's q / k / v (just written: cache-hot), then keys, values, mask and
    // - once pos has arrived - the rotary table row
    u32 xq[QPT], cq[QPT];
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int idx = tid + NTH * k, hh = idx >> 6, p = idx & 63;
        xq[k] = *reinterpret_cast<const u32*>(row + (int64_t)(h0 + hh) * D + 2 * p);
    }
    const u32 xk = *reinterpret_cast<const u32*>(row + (int64_t)(H + g) * D + 2 * (tid & 63));
    const u32x4 xv = *reinterpret_cast<const u32x4*>(row + (int64_t)(H + G + g) * D + 8 * (tid & 15));
    u32x4 kf[PT][4];                                          // [position tile][d chunk j]: d = 32 j + 8 q .. + 7 of row li
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int r = p0 + 16 * pt + li, rc = r < wlen ? r : wlen - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[pt][j] = *reinterpret_cast<const u32x4*>(kb + rc * pitch + 32 * j + 8 * q);
    }
    u32x4 vr[8 * NH];                                         // row 4 i + q of the wave, 16-byte chunk li
#pragma unroll
    for (int i = 0; i < 8 * NH; ++i) {
        const int r = p0 + 4 * i + q, rc = r < wlen ? r : wlen - 1;
        vr[i] = *reinterpret_cast<const u32x4*>(vb + rc * pitch + 8 * li);
    }
    const float* mk = mask + (int64_t)b * cap_full + t_lo;
    float mr[PT][4];                                          // mask of position 16 pt + 4 q + e (the C layout's rows)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = p0 + 16 * pt + 4 * q + e;
            mr[pt][e] = mk[r < wlen ? r : wlen - 1];
        }
    const int wrow = (int)widx[0] - t_lo;                     // window-local row written by this step
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)                           // rows behind it: hidden whatever the mask says (see decode_attention_kernel)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (p0 + 16 * pt + 4 * q + e > wrow) mr[pt][e] = -1e30f;
    const bool has_new = wrow >= 0 && wrow < wlen;            // block-uniform
    const T* cs = table + clamp_pos(pos[b], cap_full) * D;
#pragma unroll
    for (int k = 0; k < QPT; ++k) cq[k] = *reinterpret_cast<const u32*>(cs + 2 * ((tid + NTH * k) & 63));
    // xk / xv are used under `tid < 64` / `tid < 80` only: left alone, the compiler sinks their loads into those
    // branches, where they are a fresh global round trip behind a queue drain in front of the first barrier
    u32 xk_p = xk, xv0 = xv[0], xv1 = xv[1], xv2 = xv[2], xv3 = xv[3];
    asm volatile("" : "+v"(xk_p), "+v"(xv0), "+v"(xv1), "+v"(xv2), "+v"(xv3));

    // rotary: 16 heads x 64 pairs (QPT per thread), the group's key pair (threads 0..63), the value row (threads 64..79)
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int idx = tid + NTH * k, hh = idx >> 6, p = idx & 63;
        float x0, x1, c0, c1, y0, y1;
        unpack2<T>(xq[k], x0, x1);
        unpack2<T>(cq[k], c0, c1);
        rope_pair(x0, x1, c0, c1, y0, y1);
        *reinterpret_cast<u32*>(qs + hh * QP + 4 * p) =
            pack2<T>(Act<T>::round(y0) / sqrt_d, Act<T>::round(y1) / sqrt_d);
    }
    if (tid < 64) {
        float x0, x1, c0, c1, y0, y1;
        unpack2<T>(xk_p, x0, x1);
        unpack2<T>(cq[0], c0, c1);                            // idx = tid: pair tid
        rope_pair(x0, x1, c0, c1, y0, y1);
        reinterpret_cast<u32*>(knew)[tid] = pack2<T>(y0, y1);
    } else if (tid < 80) {
        reinterpret_cast<u32x4*>(vnew)[tid - 64] = u32x4{xv0, xv1, xv2, xv3};
    }
    __syncthreads();
    if (has_new && tid < 32) {                                // the cache row of this step (read back from LDS by nobody else)
        const int64_t at = (((int64_t)b * cap_full + t_lo + wrow) * G + g) * D;
        if (tid < 16) *reinterpret_cast<u32x4*>(Kc + at + 8 * tid) = reinterpret_cast<const u32x4*>(knew)[tid];
        else *reinterpret_cast<u32x4*>(Vc + at + 8 * (tid - 16)) = reinterpret_cast<const u32x4*>(vnew)[tid - 16];
    }

    // S^T = K Q^T: lane (li, q) ends with scores of head li at positions 16 pt + 4 q + e
    const int wl = has_new ? wrow - p0 : -1;                  // wave-local new row (outside [0, PW): not this wave's)
    u32x4 qf[4], knf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        qf[j] = *reinterpret_cast<const u32x4*>(qs + li * QP + 64 * j + 16 * q);
        knf[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(knew) + 64 * j + 16 * q);
    }
    f32x4 s[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const bool is_new = wl == 16 * pt + li;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = AttMma<T>::qk(is_new ? knf[j] : kf[pt][j], qf[j], acc);
        s[pt] = acc;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = p0 + 16 * pt + 4 * q + e;
            const float sv = r < wlen ? Act<T>::round(s[pt][e]) + mr[pt][e] : -INFINITY;
            s[pt][e] = sv;
            mx = fmaxf(mx, sv);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float lsum = 0.f;
    u32x2v pf[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        float ev[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ev[e] = s[pt][e] == -INFINITY ? 0.f : __expf(s[pt][e] - mx);
            lsum += ev[e];
        }
        pf[pt] = u32x2v{pack2<T>(ev[0], ev[1]), pack2<T>(ev[2], ev[3])};
    }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);

    // O^T = V^T P^T, 32 value rows at a time through the wave's LDS image
    unsigned char* img = vimg[wv];
    const u32x4 vnf = reinterpret_cast<const u32x4*>(vnew)[li];
    f32x4 o[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = 32 * hf + 4 * i + q;               // wave-local row
            *reinterpret_cast<u32x4*>(img + (4 * i + q) * VP + 16 * li) = rl == wl ? vnf : vr[8 * hf + i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ptl = 0; ptl < 2; ++ptl)
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                const unsigned char* p = img + (16 * ptl + 4 * q + (li >> 2)) * VP + 2 * (16 * dt + 4 * (li & 3));
                const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(const_cast<unsigned char*>(p)));
                o[dt] = AttMma<T>::pv(__builtin_bit_cast(u32x2v, a), pf[2 * hf + ptl], o[dt]);
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // partial results of the wave: O[head li][d = 16 dt + 4 q + e], max and exp-sum per head
    float* ow = reinterpret_cast<float*>(img);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4*>(ow + li * OP + 16 * dt + 4 * q) = o[dt];
    if (q == 0) {
        mw[wv][li] = mx;
        lw[wv][li] = lsum;
    }
    __syncthreads();
    {
        constexpr int CW = HP * D / NTH;                      // d values per thread: 8 (4 waves) or 4 (8 waves)
        constexpr int CPH = D / CW;                           // threads per head
        const int hh = tid / CPH, ch = tid - hh * CPH;
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < NWV; ++w) m = fmaxf(m, mw[w][hh]);
        float l = 0.f, acc[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) acc[e] = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            const float mwv = mw[w][hh];
            const float fw = mwv == -INFINITY ? 0.f : __expf(mwv - m);
            l = __builtin_fmaf(lw[w][hh], fw, l);
            const float* src = reinterpret_cast<const float*>(vimg[w]) + hh * OP + CW * ch;
#pragma unroll
            for (int e4 = 0; e4 < CW; e4 += 4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(src + e4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e4 + e] = __builtin_fmaf(a[e], fw, acc[e4 + e]);
            }
        }
        const int64_t head = (int64_t)b * H + h0 + hh;
        if (split_out) {
            float* so = split_out + (head * gridDim.y + blockIdx.y) * (D + 2);
            if (ch == 0) {
                so[0] = m;
                so[1] = l;
            }
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int e = 0; e < CW; e += 2) *reinterpret_cast<f32x2*>(so + 2 + CW * ch + e) = f32x2{acc[e], acc[e + 1]};
        } else {
            const float inv = 1.0f / l;
            T* dst = Out + head * D + CW * ch;
            if constexpr (CW == 8) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = acc[e] * inv;
                store8<T>(dst, y);
            } else {
                typedef u32 u32x2s __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2s*>(dst) = u32x2s{pack2<T>(acc[0] * inv, acc[1] * inv), pack2<T>(acc[2] * inv, acc[3] * inv)};
            }
        }
    }
}
#endif   // QL_DEV_TUNING

// ---------------------------------------------------------------------------------------------
// Round 5: the group kernel rebuilt (the round-2 kernel above stays in the developer library: QLINEAR_ATTENTION_R2=1).
// Measured on the round-2 kernel (tools/attention_timeline.py: the 100 MHz clock stamped inside the launch, profiles/r05_attention_timeline.txt):
// two workgroups on two CUs; of 6.3 us launch to launch 5.5 are inside the kernel, and its data has only landed 2.3 - 2.9 us after entry -
// whatever is requested (the 4-byte position lands 1.0 us after entry in the first wave, 2.0 us in the last: a CU's address unit takes the
// waves' requests in arrival order, 128 KB of keys and values per CU queue in front), then a 2.6 us chain of 1 160 instructions per wave at
// two waves per SIMD.  What this kernel changes, and what each step measured (tools/ab/run_attention_ab.sh, profiles/r05_attention_ab.txt):
//   * instruction count (-0.23 us): workgroup -> (sequence, group) by one multiply with a host-made reciprocal, wave index / tile bases /
//     pointers in SGPRs, lanes add one 32-bit offset (global_load saddr form, immediates for the d chunks): ~90 instructions until the last
//     request is out (was 280); wave max / exp-sum across the four 16-lane rows by v_permlane16/32_swap (was four LDS bpermute round
//     trips); -inf handled once per wave (a safe maximum) and once per head in the merge; q / sqrt(d) as the correctly rounded quotient by
//     two fused steps on a host-rounded reciprocal (exhaustively equal to the division: tests/test_attention_scale_cpu.py);
//   * request ORDER (-0.12 us): widx, pos, queries, new key / value, keys, [pos has landed] table row, masks, values.  Requests retire in
//     order: the table row no longer waits behind the values, rotary -> barrier -> K Q^T -> softmax run while the values stream in; the
//     first barrier is a raw s_barrier behind an LDS wait (__syncthreads() would drain the values); widx / pos are VECTOR loads (as scalar
//     loads they cost a scalar-queue drain in front of the key requests);
//   * rows behind the step's own row hold nothing of this sequence (the reference appends at the END of its cache,
//     chatglm_q/model.py:148-151): their tiles' scores are -inf without MFMAs, their value / mask requests go to tile 0's rows (this CU's L1
//     has them: no bytes from L2; NOT a branch - a branch around a request costs a queue drain at its join); wave w owns the 16-position
//     tiles w, w + NWV, ...: a short context spreads over the waves;
//   * the new key / value row replaces its cache row in the ONE wave that owns its tile (wave-uniform branch).
// 6.30 -> 5.96 us at capacity 256, 9.48 -> 8.85 at 1152 - 4224 (window split + combine launch).  Measured and NOT adopted: the table row
// without the position's round trip (QL_ATT_NOPOS, a timing-only build: -0.09 us - a per-step row buffer would buy that), a barrier between
// the small first-needed requests and the bulk ones (QL_ATT_PREBAR: 0), waiting for widx to skip the keys' requests too (+0.5 us), 4 waves
// (+0.06).  Same rounding points as the round-2 kernel except that P = exp(s - wave max) is taken per wave of INTERLEAVED tiles.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float xrow_max(float x) {          // max over the lanes li, li + 16, li + 32, li + 48
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    x = fmaxf(a, b);
    a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ float xrow_sum(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    x = a + b;
    a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// x / d for a host-rounded r = RN(1 / d): q0 = x r, then one exact residual and one correction (Markstein): the correctly rounded quotient
__device__ __forceinline__ float div_by(float x, float d, float r) {
    const float q0 = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q0, d, x), r, q0);
}

template <typename T, int NWV>
__global__ __launch_bounds__(NWV * 64) void decode_attention_group_kernel(const T* __restrict__ QKV, const int64_t* __restrict__ widx,
                                                                          const int64_t* __restrict__ pos, T* Kc, T* Vc, int ga, int cap_full,
                                                                          int ldq32, unsigned ginv, const float* __restrict__ mask,
                                                                          const T* __restrict__ table,
                                                                          float sqrt_d, float rsqrt_d, T* __restrict__ Out,
                                                                          float* __restrict__ split_out, Prefetch next) {
    // The leading 14 argument dwords are preloaded into SGPRs at wave launch (Makefile): what the position / query / key / value loads need
    // (the mask and table pointers arrive by scalar load while those are issued).
    // ga = G | attention workgroups << 12 (B G of them along x, the prefetchers behind); ginv = floor(2^32 / G) + 1: x / G by one multiply.
    const int G = ga & 0xFFF, att_blocks = (int)((unsigned)ga >> 12);
    if ((int)blockIdx.x >= att_blocks) {                      // workgroups past the attention ones warm the caches for the next launch
        if (blockIdx.y == 0) prefetch_blocks(next, (int)blockIdx.x - att_blocks, (int)gridDim.x - att_blocks, att_blocks);
        return;
    }
    static_assert(sizeof(T) == 2, "16-bit dtypes");
    static_assert(NWV == 4 || NWV == 8, "waves per block");
    constexpr int D = 128, HP = 16, WIN = 256, NTH = NWV * 64;
    constexpr int PT = WIN / NWV / 16;                        // 16-position tiles per wave
    constexpr int NH = PT / 2;                                // 32-row value images per wave
    constexpr int QPT = HP * 64 / NTH;                        // query pairs rotated per thread
    constexpr int VP = 288;                                   // bytes per value row in LDS: rows 8 banks apart (tr reads conflict-free)
    constexpr int QP = 272;                                   // bytes per query head in LDS
    constexpr int OP = 132;                                   // floats per head of a wave's partial output
    __shared__ __attribute__((aligned(16))) unsigned char vimg[NWV][32 * VP];   // per wave: 32 value rows; later its partial O
    __shared__ __attribute__((aligned(16))) unsigned char qs[HP * QP];
    __shared__ __attribute__((aligned(16))) T knew[D];
    __shared__ __attribute__((aligned(16))) T vnew[D];
    __shared__ __attribute__((aligned(16))) float mw[HP][NWV], lw[HP][NWV];
    static_assert(HP * OP * 4 <= 32 * VP, "partial output fits the wave's value image");
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    QL_ATT_T(0, "v"(tid));
    QL_ATT_C(13);
    // widx[0] and pos[b] live in device memory.  As VECTOR loads (a laundered zero lane offset keeps the compiler from making them scalar
    // loads) they retire in order with everything else: a scalar load here cost a queue drain in front of the key loads (its destination
    // registers were reused), and the table row's address waits for exactly one load instead of for every scalar one.
    unsigned z0 = 0;
    asm volatile("" : "+v"(z0));
    const int b = G == 1 ? (int)blockIdx.x : (int)__umulhi(blockIdx.x, ginv), g = (int)blockIdx.x - b * G;
    const int w_abs_v = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(widx) + z0);       // (low words: |values| < 2^31)
    const int pos_v = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(pos + b) + z0);
    const int t_lo = (int)blockIdx.y * WIN;
    const int wlen = cap_full - t_lo < WIN ? cap_full - t_lo : WIN;
    const int pitch = G * D * 2;                              // bytes per cache row
    const int64_t win0 = (((int64_t)b * cap_full + t_lo) * G + g) * D;
    const char* kb = reinterpret_cast<const char*>(Kc + win0);
    const char* vb = reinterpret_cast<const char*>(Vc + win0);
    const T* row = QKV + (int64_t)b * ldq32;
    const float* mk = mask + (int64_t)b * cap_full + t_lo;

    // every load of the block up front.  The group's 16 query heads are 4 KB in a row: thread t takes pairs t, t + NTH, ...
    u32 xq[QPT];
    {
        const char* qb = reinterpret_cast<const char*>(row + g * (HP * D));
#pragma unroll
        for (int k = 0; k < QPT; ++k) xq[k] = *reinterpret_cast<const u32*>(qb + (unsigned)(4 * tid) + 4 * NTH * k);
    }
#ifdef QL_ATT_NOPOS                                          // timing ablation (results wrong): the table row without the position's round trip
    const u32 cq = *reinterpret_cast<const u32*>(reinterpret_cast<const char*>(table + (b + (pos_v & 0)) * D) + (unsigned)(4 * lane));
#endif
    // (every wave requests the new key pair and value chunk: two instructions; a branch around them costs a queue drain at its join)
    u32 xk = *reinterpret_cast<const u32*>(reinterpret_cast<const char*>(row + (HP * G + g) * D) + (unsigned)(4 * lane));
    u32x4 xv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(row + (HP * G + G + g) * D) + (unsigned)(16 * li));
#ifdef QL_ATT_PREBAR
    // the CU's address unit takes requests in arrival order: every wave's small first-needed requests are in its queue before any wave's
    // bulk ones (a barrier nobody waits long at: the waves enter together)
    asm volatile("s_barrier" ::: "memory");
#endif
    u32x4 kf[PT][4];                                          // [tile][d chunk j]: d = 32 j + 8 q .. + 7 of row li
    u32x4 vr[4 * PT];                                         // tile i >> 2, row 4 (i & 3) + q of it, 16-byte chunk li
    f32x4 mr[PT];                                             // mask of position 4 q + e of the tile
    // scalar base + one 32-bit lane offset (+ immediate): row clamped to the window (the ragged last tile of a cache), three instructions
    const int last = wlen - 1;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int t0 = 16 * (wv + NWV * pt);                  // the tile's first window-local position (scalar)
        const int r = t0 + li;
        const unsigned off = (unsigned)((r < last ? r : last) * pitch + 16 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[pt][j] = *reinterpret_cast<const u32x4*>(kb + off + 64 * j);
    }
    // The position's table row is needed first (rotary -> barrier -> scores) and the values last: requests return in order, so the row is
    // requested as soon as pos has landed - behind the keys, in front of the masks and the values - and the chain up to the softmax runs
    // while the values stream in.  widx went out in front of pos: it has landed too, so tiles behind the step's own row request nothing more.
#ifndef QL_ATT_NOPOS
    const int pos_b = __builtin_amdgcn_readfirstlane(pos_v);
    const T* cs = table + (pos_b < 0 ? 0 : pos_b <= cap_full ? pos_b : cap_full) * D;   // clamp_pos
    const u32 cq = *reinterpret_cast<const u32*>(reinterpret_cast<const char*>(cs) + (unsigned)(4 * lane));   // pair `lane` of the position's row
#endif
    // rows that exist: [0, nvalid) of the window (a write index outside the cache writes nothing and hides nothing)
    const int w_abs = __builtin_amdgcn_readfirstlane(w_abs_v);
    const int lim = w_abs >= 0 && w_abs < cap_full ? w_abs + 1 - t_lo : wlen;
    const int nvalid = lim < 0 ? 0 : lim < wlen ? lim : wlen;
    const int wrow = w_abs - t_lo;                            // window-local row written by this step
    const bool has_new = wrow >= 0 && wrow < wlen;            // block-uniform
    // wave-local new row 16 pt + (row in its tile) when the row's tile is one of this wave's, else -1
    const int wl = has_new && ((wrow >> 4) & (NWV - 1)) == wv ? 16 * ((wrow >> 4) / NWV) + (wrow & 15) : -1;
    bool tv[PT];                                              // the wave's tile pt holds a row that exists (scalar)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) tv[pt] = 16 * (wv + NWV * pt) < nvalid;
    // (no branch around a request: a join costs a queue drain.  A tile behind the step's row re-requests tile 0's rows instead - they are in
    // this CU's L1 by then: no bytes from L2)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int t0 = tv[pt] ? 16 * (wv + NWV * pt) : 0;
        if (wlen >= WIN) {                                    // (block-uniform; a ragged last window: clamped element by element)
            mr[pt] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(mk + t0) + (unsigned)(16 * q));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = t0 + 4 * q + e;
                mr[pt][e] = mk[r < last ? r : last];
            }
        }
    }
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int t0 = tv[pt] ? 16 * (wv + NWV * pt) : 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int r = t0 + 4 * a + q;
            const unsigned off = (unsigned)((r < last ? r : last) * pitch + 16 * li);
            vr[4 * pt + a] = *reinterpret_cast<const u32x4*>(vb + off);
        }
    }
    QL_ATT_T(15, "v"(tid));
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {                         // validity into the mask (while the loads fly)
        const int t0 = 16 * (wv + NWV * pt);
        if (tv[pt] && t0 + 16 > nvalid) {
#pragma unroll
            for (int e = 0; e < 4; ++e) mr[pt][e] = t0 + 4 * q + e < nvalid ? mr[pt][e] : -INFINITY;
        }
    }
    {                                                         // (keeps the two loads where they are: not sunk into the branches below)
        u32 x0 = xv[0], x1 = xv[1], x2 = xv[2], x3 = xv[3];
        asm volatile("" : "+v"(xk), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        xv = u32x4{x0, x1, x2, x3};
    }
    QL_ATT_T(1, "v"(tid));
#ifdef QL_ATT_STAMPS
    QL_ATT_T(12, "v"(xq[0]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    QL_ATT_T(2, "v"(cq));
#endif

    // rotary: 16 heads x 64 pairs (QPT per thread, all of pair `lane`), the group's key pair (wave 0), the value row (wave 1)
    float c0, c1;
    unpack2<T>(cq, c0, c1);
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        float x0, x1, y0, y1;
        unpack2<T>(xq[k], x0, x1);
        rope_pair(x0, x1, c0, c1, y0, y1);
        *reinterpret_cast<u32*>(qs + (wv + NWV * k) * QP + 4 * lane) =
            pack2<T>(div_by(Act<T>::round(y0), sqrt_d, rsqrt_d), div_by(Act<T>::round(y1), sqrt_d, rsqrt_d));
    }
    if (wv == 0) {
        float x0, x1, y0, y1;
        unpack2<T>(xk, x0, x1);
        rope_pair(x0, x1, c0, c1, y0, y1);
        reinterpret_cast<u32*>(knew)[lane] = pack2<T>(y0, y1);
    } else if (wv == 1) {
        if (lane < 16) reinterpret_cast<u32x4*>(vnew)[lane] = xv;
    }
    QL_ATT_T(3, "v"(tid));
    // the rotated queries / new row are in LDS; NOT __syncthreads(): its fence would wait for the values still in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    QL_ATT_T(4, "v"(tid));
    if (has_new && wv == NWV - 1 && lane < 32) {              // the cache row of this step (read back from LDS by nobody else)
        const int64_t at = win0 + (int64_t)wrow * (G * D);
        if (lane < 16) *reinterpret_cast<u32x4*>(Kc + at + 8 * lane) = reinterpret_cast<const u32x4*>(knew)[lane];
        else *reinterpret_cast<u32x4*>(Vc + at + 8 * (lane - 16)) = reinterpret_cast<const u32x4*>(vnew)[lane - 16];
    }

    // S^T = K Q^T: lane (li, q) ends with scores of head li at positions 4 q + e of each tile
    u32x4 qf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) qf[j] = *reinterpret_cast<const u32x4*>(qs + li * QP + 64 * j + 16 * q);
    if (wl >= 0) {                                            // this wave owns the new row's tile: the row comes from LDS
        const bool mine_k = li == (wl & 15);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
            if ((wl >> 4) == pt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 knf = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(knew) + 64 * j + 16 * q);
                    kf[pt][j] = mine_k ? knf : kf[pt][j];
                }
            }
    }
    f32x4 s[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (tv[pt]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = AttMma<T>::qk(kf[pt][j], qf[j], acc);
        }
        s[pt] = acc;
    }
    QL_ATT_T(5, "v"(s[0][0]), "v"(s[PT - 1][3]));
    float mx = -INFINITY;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
        if (tv[pt]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[pt][e] = Act<T>::round(s[pt][e]) + mr[pt][e];
                mx = fmaxf(mx, s[pt][e]);
            }
        }
    mx = xrow_max(mx);
    const float mxs = mx == -INFINITY ? 0.f : mx;             // a wave without a visible row: every exp below is exp(-inf) = 0
    float lsum = 0.f;
    u32x2v pf[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        pf[pt] = u32x2v{0u, 0u};
        if (tv[pt]) {
            float ev[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ev[e] = __expf(s[pt][e] - mxs);
                lsum += ev[e];
            }
            pf[pt] = u32x2v{pack2<T>(ev[0], ev[1]), pack2<T>(ev[2], ev[3])};
        }
    }
    lsum = xrow_sum(lsum);
    QL_ATT_T(6, "v"(lsum), "v"(pf[PT - 1]));

    // O^T = V^T P^T, 32 value rows (two tiles) at a time through the wave's LDS image
    if (wl >= 0) {                                            // (the values are first touched here: they streamed in behind everything above)
        const bool mine_v = q == (wl & 3);
        const u32x4 vnf = reinterpret_cast<const u32x4*>(vnew)[li];
#pragma unroll
        for (int i = 0; i < 4 * PT; ++i)
            if ((wl >> 2) == i) vr[i] = mine_v ? vnf : vr[i];
    }
    unsigned char* img = vimg[wv];
    f32x4 o[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
        if (!tv[2 * hf]) continue;                            // (tiles are valid in order: the second implies the first)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < 4 || tv[2 * hf + 1]) *reinterpret_cast<u32x4*>(img + (4 * i + q) * VP + 16 * li) = vr[8 * hf + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ptl = 0; ptl < 2; ++ptl) {
            if (!tv[2 * hf + ptl]) continue;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                const unsigned char* p = img + (16 * ptl + 4 * q + (li >> 2)) * VP + 2 * (16 * dt + 4 * (li & 3));
                const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(const_cast<unsigned char*>(p)));
                o[dt] = AttMma<T>::pv(__builtin_bit_cast(u32x2v, a), pf[2 * hf + ptl], o[dt]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    QL_ATT_T(7, "v"(o[0]), "v"(o[7]));
    // partial results of the wave: O[head li][d = 16 dt + 4 q + e], max and exp-sum per head
    float* ow = reinterpret_cast<float*>(img);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4*>(ow + li * OP + 16 * dt + 4 * q) = o[dt];
    if (q == 0) {
        mw[li][wv] = mx;
        lw[li][wv] = lsum;
    }
    QL_ATT_T(8, "v"(tid));
    __syncthreads();
    QL_ATT_T(9, "v"(tid));
    {
        constexpr int CW = HP * D / NTH;                      // d values per thread: 8 (4 waves) or 4 (8 waves)
        constexpr int CPH = D / CW;                           // threads per head
        const int hh = tid / CPH, ch = tid - hh * CPH;
        float mv[NWV], lv[NWV];
#pragma unroll
        for (int w4 = 0; w4 < NWV; w4 += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&mw[hh][w4]), c = *reinterpret_cast<const f32x4*>(&lw[hh][w4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) mv[w4 + e] = a[e], lv[w4 + e] = c[e];
        }
        float m = mv[0];
#pragma unroll
        for (int w = 1; w < NWV; ++w) m = fmaxf(m, mv[w]);
        const float ms = m == -INFINITY ? 0.f : m;            // a window without a visible row: every weight exp(-inf) = 0
        float l = 0.f, acc[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) acc[e] = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            const float fw = __expf(mv[w] - ms);
            l = __builtin_fmaf(lv[w], fw, l);
            const float* src = reinterpret_cast<const float*>(vimg[w]) + hh * OP + CW * ch;
#pragma unroll
            for (int e4 = 0; e4 < CW; e4 += 4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(src + e4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e4 + e] = __builtin_fmaf(a[e], fw, acc[e4 + e]);
            }
        }
        const int64_t head = ((int64_t)b * G + g) * HP + hh;
        if (split_out) {
            float* so = split_out + (head * gridDim.y + blockIdx.y) * (D + 2);
            if (ch == 0) {
                so[0] = m;
                so[1] = l;
            }
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int e = 0; e < CW; e += 2) *reinterpret_cast<f32x2*>(so + 2 + CW * ch + e) = f32x2{acc[e], acc[e + 1]};
        } else {
            const float inv = 1.0f / l;
            T* dst = Out + head * D + CW * ch;
            if constexpr (CW == 8) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = acc[e] * inv;
                store8<T>(dst, y);
            } else {
                typedef u32 u32x2s __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2s*>(dst) = u32x2s{pack2<T>(acc[0] * inv, acc[1] * inv), pack2<T>(acc[2] * inv, acc[3] * inv)};
            }
        }
    }
    QL_ATT_T(10, "v"(tid));
#ifdef QL_ATT_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    QL_ATT_T(11, "v"(tid));
    QL_ATT_C(14);
#endif
}

// ---------------------------------------------------------------------------------------------
// out = sum_w o_w e^(m_w - m) / sum_w l_w e^(m_w - m): the windows of the split modes above, one block per head.
// The windows are independent loads: thread = (window slice, d pair), 4 windows in flight per thread (a loop over
// the windows with one load each was a chain of nwin global round trips: 15 us at 32 windows).
// Up to 8 windows (capacities up to 2048 - the usual ones): ONE wave per head, thread = d pair, every window's (max, sum) and pair requested up
// front - one memory round trip, no LDS, no barrier (the general kernel below: the maxima, a block reduction, then the records - two round trips and
// three barriers; 2.9 -> 2.3 us per launch, round 5).  Windows are summed in order.
template <typename T>
__global__ __launch_bounds__(64) void attention_combine8_kernel(const float* __restrict__ part, T* __restrict__ Out, int nwin) {
    constexpr int D = 128;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float* p = part + (int64_t)blockIdx.x * nwin * (D + 2);
    const int dd = threadIdx.x;
    f32x2 ml[8], x[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const float* rec = p + (int64_t)(w < nwin ? w : nwin - 1) * (D + 2);       // (past the end: the last record again, weight 0)
        ml[w] = *reinterpret_cast<const f32x2*>(rec);
        x[w] = *reinterpret_cast<const f32x2*>(rec + 2 + 2 * dd);
    }
    float m = -INFINITY;
#pragma unroll
    for (int w = 0; w < 8; ++w) m = fmaxf(m, w < nwin ? ml[w][0] : -INFINITY);
    const float ms = m == -INFINITY ? 0.f : m;
    float l = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const float f = w < nwin ? __expf(ml[w][0] - ms) : 0.f;                   // exp(-inf) = 0: a window without a visible row
        l = __builtin_fmaf(ml[w][1], f, l);
        o0 = __builtin_fmaf(x[w][0], f, o0);
        o1 = __builtin_fmaf(x[w][1], f, o1);
    }
    const float inv = 1.0f / l;
    *reinterpret_cast<u32*>(Out + (int64_t)blockIdx.x * D + 2 * dd) = pack2<T>(o0 * inv, o1 * inv);
}

// 9 .. 32 windows (capacities up to 8192): four waves per head, wave s takes the windows s, s + 4, ... (eight at most) exactly as above - one memory
// round trip - and the four (max, sum, pair) results are merged through LDS behind ONE barrier, in slice order.
template <typename T>
__global__ __launch_bounds__(256) void attention_combine32_kernel(const float* __restrict__ part, T* __restrict__ Out, int nwin) {
    constexpr int D = 128;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4c __attribute__((ext_vector_type(4)));
    __shared__ f32x4c mrg[4][64];                             // (max, sum, o0, o1) per slice and d pair
    const float* p = part + (int64_t)blockIdx.x * nwin * (D + 2);
    const int dd = threadIdx.x & 63, ws = threadIdx.x >> 6;
    f32x2 ml[8], x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int w = ws + 4 * i;
        const float* rec = p + (int64_t)(w < nwin ? w : nwin - 1) * (D + 2);
        ml[i] = *reinterpret_cast<const f32x2*>(rec);
        x[i] = *reinterpret_cast<const f32x2*>(rec + 2 + 2 * dd);
    }
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) m = fmaxf(m, ws + 4 * i < nwin ? ml[i][0] : -INFINITY);
    const float ms = m == -INFINITY ? 0.f : m;
    float l = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float f = ws + 4 * i < nwin ? __expf(ml[i][0] - ms) : 0.f;
        l = __builtin_fmaf(ml[i][1], f, l);
        o0 = __builtin_fmaf(x[i][0], f, o0);
        o1 = __builtin_fmaf(x[i][1], f, o1);
    }
    mrg[ws][dd] = f32x4c{m, l, o0, o1};
    __syncthreads();
    if (ws != 0) return;
    float mm = -INFINITY;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) mm = fmaxf(mm, mrg[s2][dd][0]);
    const float mms = mm == -INFINITY ? 0.f : mm;
    float lt = 0.f, t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
        const f32x4c r = mrg[s2][dd];
        const float f = __expf(r[0] - mms);                   // exp(-inf) = 0: a slice without a visible row
        lt = __builtin_fmaf(r[1], f, lt);
        t0 = __builtin_fmaf(r[2], f, t0);
        t1 = __builtin_fmaf(r[3], f, t1);
    }
    const float inv = 1.0f / lt;
    *reinterpret_cast<u32*>(Out + (int64_t)blockIdx.x * D + 2 * dd) = pack2<T>(t0 * inv, t1 * inv);
}

template <typename T, int D>
__global__ __launch_bounds__(256) void attention_combine_kernel(const float* __restrict__ part, T* __restrict__ Out, int nwin) {
    constexpr int HD = D / 2, NS = 256 / HD;                  // d pairs, window slices
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ float red[4];
    __shared__ float acc[NS][D + 1];
    const float* p = part + (int64_t)blockIdx.x * nwin * (D + 2);
    const int tid = threadIdx.x, ws = tid / HD, dd = tid - ws * HD;
    float m = -INFINITY;
    for (int w = tid; w < nwin; w += 256) m = fmaxf(m, p[(int64_t)w * (D + 2)]);
    m = block_max_256(m, red);
    float l = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll 4
    for (int w = ws; w < nwin; w += NS) {
        const float* rec = p + (int64_t)w * (D + 2);
        const f32x2 ml = *reinterpret_cast<const f32x2*>(rec);
        const f32x2 x = *reinterpret_cast<const f32x2*>(rec + 2 + 2 * dd);
        const float f = ml[0] == -INFINITY ? 0.f : __expf(ml[0] - m);
        l = __builtin_fmaf(ml[1], f, l);
        o0 = __builtin_fmaf(x[0], f, o0);
        o1 = __builtin_fmaf(x[1], f, o1);
    }
    acc[ws][2 * dd] = o0;
    acc[ws][2 * dd + 1] = o1;
    if (dd == 0) acc[ws][D] = l;
    __syncthreads();
    if (tid < D) {
        float o = 0.f, lt = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            o += acc[s2][tid];
            lt += acc[s2][D];
        }
        Act<T>::store(Out + (int64_t)blockIdx.x * D + tid, o / lt);
    }
}

// ---------------------------------------------------------------------------------------------
// Prefill attention's middle: P = round(softmax_fp32(scores + mask)) row by row (chatglm_q/model.py:166-170: the
// additive mask promotes the fp16 scores to fp32, the softmax runs in fp32, its result is cast to the activation
// dtype).  As separate torch kernels that is 16 bytes of traffic per score (add, softmax, cast); here 4.
// One wave per row; scores (rows, T) with row stride lds, mask row = row % mask_rows (the S query positions of a
// chunk; every head shares them), mask row stride ldm.  T <= 64 * 32 is held in registers, longer rows loop.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void masked_softmax_kernel(const T* __restrict__ Sc, const float* __restrict__ mask,
                                                             T* __restrict__ P, int64_t rows, int Tn, int mask_rows,
                                                             int64_t lds, int64_t ldm, int64_t ldp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* s = Sc + row * lds;
    const float* mk = mask ? mask + (row % mask_rows) * ldm : nullptr;
    T* p = P + row * ldp;
    constexpr int NR = 32;                                    // values per lane kept in registers
    if (Tn <= 64 * NR) {
        float v[NR];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int t = lane + 64 * i;
            v[i] = t < Tn ? Act<T>::load(s + t) + (mk ? mk[t] : 0.f) : -INFINITY;
            mx = fmaxf(mx, v[i]);
        }
        mx = wave_max_dpp(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            v[i] = __expf(v[i] - mx);                         // exp(-inf) = 0 past the end of the row
            sum += v[i];
        }
        const float inv = 1.0f / wave_sum(sum);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int t = lane + 64 * i;
            if (t < Tn) Act<T>::store(p + t, v[i] * inv);
        }
        return;
    }
    float mx = -INFINITY;
    for (int t = lane; t < Tn; t += 64) mx = fmaxf(mx, Act<T>::load(s + t) + (mk ? mk[t] : 0.f));
    mx = wave_max_dpp(mx);
    float sum = 0.f;
    for (int t = lane; t < Tn; t += 64) sum += __expf(Act<T>::load(s + t) + (mk ? mk[t] : 0.f) - mx);
    const float inv = 1.0f / wave_sum(sum);
    for (int t = lane; t < Tn; t += 64) Act<T>::store(p + t, __expf(Act<T>::load(s + t) + (mk ? mk[t] : 0.f) - mx) * inv);
}

// out = round(round(silu(h)) * gate), (h, gate) = halves of a (rows, 2 hidden) matrix; 8 values per thread
// SiLU(h) * gate of one row per block, row in registers, emitted as int8 + scale (and as the 16-bit row when Out is given): the
// quantising producer in front of an int8-activation w_out.  Same arithmetic as silu_mul_kernel, then quantize_int8 on the
// rounded values.
template <typename T, int VPT>
__global__ __launch_bounds__(256) void silu_mul_quant_kernel(const T* __restrict__ In, T* __restrict__ Out, int hidden, int64_t ldin,
                                                             int64_t ldo, int8_t* __restrict__ Aq, float* __restrict__ a_scale) {
    __shared__ float red[4];
    const T* in = In + (int64_t)blockIdx.x * ldin;
    float v[VPT][8];
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int i = (threadIdx.x + u * 256) * 8;
        if (i < hidden) {
            float h[8], g[8], y[8];
            load8<T>(in + i, h);
            load8<T>(in + hidden + i, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = Act<T>::round(h[e] / (1.0f + __expf(-h[e]))) * g[e];
            if (Out) store8<T>(Out + (int64_t)blockIdx.x * ldo + i, y);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[u][e] = Act<T>::round(y[e]);
                mx = fmaxf(mx, fabsf(v[u][e]));
            }
        }
    }
    mx = block_max_256(mx, red);
    const float s = fmaxf(mx / 127.0f, 1e-10f);
    if (threadIdx.x == 0) a_scale[blockIdx.x] = s;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int i = (threadIdx.x + u * 256) * 8;
        if (i < hidden) *reinterpret_cast<u32x2*>(Aq + (int64_t)blockIdx.x * hidden + i) = quant8_pack(v[u], s);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void silu_mul_kernel(const T* __restrict__ In, T* __restrict__ Out, int hidden,
                                                       int64_t ldin, int64_t ldo) {
    const T* in = In + (int64_t)blockIdx.y * ldin;
    T* o = Out + (int64_t)blockIdx.y * ldo;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= hidden) return;
    float h[8], g[8], y[8];
    load8<T>(in + i, h);
    load8<T>(in + hidden + i, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = Act<T>::round(h[e] / (1.0f + __expf(-h[e]))) * g[e];
    store8<T>(o + i, y);
}

// ---------------------------------------------------------------------------------------------
// greedy step bookkeeping in ONE launch (block = batch row): tok = argmax(logits row) (lowest index wins
// ties, as torch.argmax), pos += 1; block 0 also advances the shared cache write index and unmasks that
// position in every row's mask (the next position may attend to itself).
// Replaces torch's argmax (a 23 us single-block reduction at 65024 logits), a copy, an index_fill and two adds.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void greedy_advance_kernel(const T* __restrict__ logits, int N, int64_t ldl,
                                                              int64_t* __restrict__ tok, int64_t* __restrict__ write_index,
                                                              int64_t* __restrict__ pos, float* __restrict__ mask, int capacity) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const T* row = logits + (int64_t)blockIdx.x * ldl;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    int i0 = 0;
    if constexpr (sizeof(T) == 2) {
        // 16-byte chunks, 8 per thread requested before the first compare (one block reads the whole row: with
        // scalar 2-byte loads in a dependent loop this kernel took 24 us for 65024 logits)
        if ((reinterpret_cast<uintptr_t>(row) & 15) == 0) {
            const int nch = N >> 3;
            for (int cbase = 0; cbase < nch; cbase += 8 * 1024) {
                u32x4 r[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ch = cbase + u * 1024 + threadIdx.x;
                    r[u] = *reinterpret_cast<const u32x4*>(row + (int64_t)(ch < nch ? ch : nch - 1) * 8);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ch = cbase + u * 1024 + threadIdx.x;
                    if (ch < nch) {
                        float v[8];
                        unpack8<T>(r[u], v);
#pragma unroll
                        for (int e = 0; e < 8; ++e)           // ascending index: strict > keeps the lowest index on ties
                            if (v[e] > best) { best = v[e]; besti = ch * 8 + e; }
                    }
                }
            }
            i0 = nch << 3;
        }
    }
    for (int i = i0 + threadIdx.x; i < N; i += 1024) {
        const float v = Act<T>::load(row + i);
        if (v > best || (v == best && i < besti)) { best = v; besti = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(besti, off, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
        tok[blockIdx.x] = besti == 0x7fffffff ? 0 : besti;    // an all-NaN row compares false everywhere: stay in range
        pos[blockIdx.x] += 1;
        if (blockIdx.x == 0) {                       // the shared write index and every row's mask: one writer, no race
            const int64_t nw = write_index[0] + 1;
            write_index[0] = nw;
            if (nw < capacity)
                for (int b = 0; b < (int)gridDim.x; ++b) mask[(int64_t)b * capacity + nw] = 0.f;
        }
    }
}

#define QL_DT(dtype, CALL)                                        \
    switch (dtype) {                                              \
    case QL_DTYPE_F32: { typedef float T; CALL; break; }          \
    case QL_DTYPE_F16: { typedef f16 T; CALL; break; }            \
    case QL_DTYPE_BF16: { typedef __bf16 T; CALL; break; }        \
    default: return QL_ERR_BAD_DTYPE;                             \
    }

template <typename T, bool FUSE_ADD>
static int launch_rmsnorm(const void* X, const void* Delta, const void* W, void* Hout, void* Out, int64_t rows, int64_t dim,
                          int64_t ldx, int64_t ldo, float eps, hipStream_t st) {
    const int vpt = (int)((dim / 8 + 255) / 256);
#define QL_RMS(V)                                                                                                  \
    rmsnorm_kernel<T, V, FUSE_ADD><<<(unsigned)rows, 256, 0, st>>>((const T*)X, (const T*)Delta, (const T*)W, (T*)Hout, \
                                                                   (T*)Out, (int)dim, ldx, ldo, eps)
    if (vpt <= 1) QL_RMS(1);
    else if (vpt <= 2) QL_RMS(2);
    else if (vpt <= 4) QL_RMS(4);
    else if (vpt <= 8) QL_RMS(8);
    else return QL_ERR_UNSUPPORTED;
#undef QL_RMS
    return finish_launch();
}

template <typename T, bool FUSE_ADD>
static int launch_rmsnorm_quant(const void* X, const void* Delta, const void* W, void* Hout, void* Out, int8_t* Aq, float* a_scale,
                                int64_t rows, int64_t dim, int64_t ldx, int64_t ldo, float eps, hipStream_t st) {
    const int vpt = (int)((dim / 8 + 255) / 256);
#define QL_RMSQ(V)                                                                                                       \
    rmsnorm_kernel<T, V, FUSE_ADD, true><<<(unsigned)rows, 256, 0, st>>>((const T*)X, (const T*)Delta, (const T*)W, (T*)Hout, \
                                                                         (T*)Out, (int)dim, ldx, ldo, eps, Aq, a_scale)
    if (vpt <= 1) QL_RMSQ(1);
    else if (vpt <= 2) QL_RMSQ(2);
    else if (vpt <= 4) QL_RMSQ(4);
    else if (vpt <= 8) QL_RMSQ(8);
    else return QL_ERR_UNSUPPORTED;
#undef QL_RMSQ
    return finish_launch();
}

int rmsnorm_quant(int dtype, const void* X, const void* Delta, const void* W, void* Hout, void* Out, int8_t* Aq, float* a_scale,
                  int64_t rows, int64_t dim, int64_t ldx, int64_t ldo, float eps, hipStream_t st) {
    if (dtype == QL_DTYPE_F32) return QL_ERR_UNSUPPORTED;
    if (Delta) {
        QL_DT(dtype, return (launch_rmsnorm_quant<T, true>(X, Delta, W, Hout, Out, Aq, a_scale, rows, dim, ldx, ldo, eps, st)))
    } else {
        QL_DT(dtype, return (launch_rmsnorm_quant<T, false>(X, Delta, W, Hout, Out, Aq, a_scale, rows, dim, ldx, ldo, eps, st)))
    }
    return QL_ERR_BAD_DTYPE;
}

template <typename T>
static int launch_silu_mul_quant(const void* In, void* Out, int8_t* Aq, float* a_scale, int64_t rows, int64_t hidden, int64_t ldin,
                                 int64_t ldo, hipStream_t st) {
    const int vpt = (int)((hidden / 8 + 255) / 256);
#define QL_SMQ(V) silu_mul_quant_kernel<T, V><<<(unsigned)rows, 256, 0, st>>>((const T*)In, (T*)Out, (int)hidden, ldin, ldo, Aq, a_scale)
    if (vpt <= 1) QL_SMQ(1);
    else if (vpt <= 2) QL_SMQ(2);
    else if (vpt <= 4) QL_SMQ(4);
    else if (vpt <= 7) QL_SMQ(7);
    else if (vpt <= 8) QL_SMQ(8);
    else return QL_ERR_UNSUPPORTED;
#undef QL_SMQ
    return finish_launch();
}

int silu_mul_quant(int dtype, const void* In, void* Out, int8_t* Aq, float* a_scale, int64_t rows, int64_t hidden, int64_t ldin,
                   int64_t ldo, hipStream_t st) {
    if (dtype == QL_DTYPE_F32) return QL_ERR_UNSUPPORTED;
    QL_DT(dtype, return (launch_silu_mul_quant<T>(In, Out, Aq, a_scale, rows, hidden, ldin, ldo, st)))
    return QL_ERR_BAD_DTYPE;
}

int rmsnorm(int dtype, const void* X, const void* Delta, const void* W, void* Hout, void* Out, int64_t rows, int64_t dim,
            int64_t ldx, int64_t ldo, float eps, hipStream_t st) {
    if (Delta) {
        QL_DT(dtype, return (launch_rmsnorm<T, true>(X, Delta, W, Hout, Out, rows, dim, ldx, ldo, eps, st)))
    } else {
        QL_DT(dtype, return (launch_rmsnorm<T, false>(X, Delta, W, Hout, Out, rows, dim, ldx, ldo, eps, st)))
    }
    return QL_ERR_BAD_DTYPE;
}

int rope_kv_write(int dtype, const void* QKV, const void* table, const int64_t* pos, const int64_t* widx, void* Qout,
                  void* Kc, void* Vc, int64_t B, int64_t S, int64_t H, int64_t G, int64_t D, int64_t capacity,
                  int64_t ldqkv, hipStream_t st) {
    QL_DT(dtype, (rope_kv_write_kernel<T><<<(unsigned)(B * S), 256, 0, st>>>((const T*)QKV, (const T*)table, pos, widx, (T*)Qout,
                                                                              (T*)Kc, (T*)Vc, (int)S, (int)H, (int)G, (int)D,
                                                                              (int)capacity, ldqkv)))
    return finish_launch();
}

constexpr int kAttnWindow = 256;                          // positions per block in split mode (= the prefetched tile)
template <typename T, bool ROPE>
static int launch_attention(const void* Q, void* Kc, void* Vc, const float* mask, void* Out, int64_t B, int64_t H, int64_t G,
                            int64_t D, int64_t capacity, const void* table, const int64_t* pos, const int64_t* widx,
                            int64_t ldq, float* split_ws, const Prefetch& pf, hipStream_t st) {
    const float sq = sqrtf((float)D);
    const int nwin = split_ws ? (int)((capacity + kAttnWindow - 1) / kAttnWindow) : 1;
    if constexpr (ROPE && sizeof(T) == 2) {
        // 16 heads per key/value group: the group kernel on the matrix cores (QLINEAR_DISPATCH=nogroupattn: per-head kernels)
        const bool one_window = capacity <= kAttnWindow;
        if (D == 128 && H == 16 * G && G < 4096 && B * G < (1 << 19) && ldq % 8 == 0 && ldq <= 0x7fffffff && (one_window || split_ws) &&
            !(dispatch_flags() & QL_D_NOGROUPATTN)) {
            // prefetch workgroups: about one per CU (they only issue loads); none without a descriptor, none when the
            // attention itself fills the chip (windows of a long context, large batches: measured slower)
            const int att_blocks = (int)(B * G);
            const int npf = pf.blocks > 0 && one_window && att_blocks <= 16 ? (pf.blocks < QL_PF_BLOCKS ? ((pf.blocks + 7) & ~7) : QL_PF_BLOCKS) : 0;
            dim3 gridg((unsigned)(att_blocks + npf), (unsigned)(one_window ? 1 : nwin));
#ifdef QL_DEV_TUNING
            if (QL_TUNE("QLINEAR_ATTENTION_R2", 0)) {         // round 2's kernel (8 waves x 32 positions: 6.3 us against 6.6 for 4 x 64)
                const int hg = (int)H | ((int)G << 16);
                if (QL_TUNE("QLINEAR_ATTENTION_WAVES", 8) == 4)
                    decode_attention_mfma_kernel<T, 4><<<gridg, 256, 0, st>>>((const T*)Q, pos, widx, (T*)Kc, (T*)Vc, hg, att_blocks,
                                                                             (int)capacity, (int)ldq, (const T*)table, mask, sq,
                                                                             (T*)Out, one_window ? nullptr : split_ws, pf);
                else
                    decode_attention_mfma_kernel<T, 8><<<gridg, 512, 0, st>>>((const T*)Q, pos, widx, (T*)Kc, (T*)Vc, hg, att_blocks,
                                                                             (int)capacity, (int)ldq, (const T*)table, mask, sq,
                                                                             (T*)Out, one_window ? nullptr : split_ws, pf);
            } else
#endif
            {
                const int ga = (int)G | (att_blocks << 12);
                const unsigned ginv = (unsigned)(0x100000000ull / (unsigned long long)G) + 1u;   // x / G = (x * ginv) >> 32 for x G < 2^32
                const float rsq = (float)(1.0 / (double)sq);  // RN(1 / sqrt(d)) for the kernel's two-step division
#ifdef QL_DEV_TUNING
                if (QL_TUNE("QLINEAR_ATTENTION_WAVES", 8) == 4)
                    decode_attention_group_kernel<T, 4><<<gridg, 256, 0, st>>>((const T*)Q, widx, pos, (T*)Kc, (T*)Vc, ga, (int)capacity, (int)ldq, ginv,
                                                                              mask, (const T*)table, sq, rsq, (T*)Out,
                                                                              one_window ? nullptr : split_ws, pf);
                else
#endif
                    decode_attention_group_kernel<T, 8><<<gridg, 512, 0, st>>>((const T*)Q, widx, pos, (T*)Kc, (T*)Vc, ga, (int)capacity, (int)ldq, ginv,
                                                                              mask, (const T*)table, sq, rsq, (T*)Out,
                                                                              one_window ? nullptr : split_ws, pf);
            }
            const int rc = finish_launch();
            if (rc != 0 || one_window) return rc;
            if (nwin <= 8) attention_combine8_kernel<T><<<(unsigned)(B * H), 64, 0, st>>>(split_ws, (T*)Out, nwin);
            else if (nwin <= 32) attention_combine32_kernel<T><<<(unsigned)(B * H), 256, 0, st>>>(split_ws, (T*)Out, nwin);
            else attention_combine_kernel<T, 128><<<(unsigned)(B * H), 256, 0, st>>>(split_ws, (T*)Out, nwin);
            return finish_launch();
        }
    }
    dim3 grid((unsigned)(B * H), (unsigned)nwin);
#define QL_ATT(DD)                                                                                                  \
    {                                                                                                               \
        const size_t lds = (size_t)((split_ws ? kAttnWindow : capacity) + DD + 4 + (256 / (DD / 8)) * DD + 2 * DD) * sizeof(float); \
        decode_attention_kernel<T, DD, ROPE><<<grid, 256, lds, st>>>(                                               \
            (const T*)Q, (T*)Kc, (T*)Vc, mask, (T*)Out, (int)H, (int)G, (int)capacity, sq, (const T*)table, pos, widx, ldq, \
            kAttnWindow, split_ws);                                                                                 \
        if (split_ws) {                                                                                             \
            const int rc = finish_launch();                                                                         \
            if (rc != 0) return rc;                                                                                 \
            attention_combine_kernel<T, DD><<<(unsigned)(B * H), 256, 0, st>>>(split_ws, (T*)Out, nwin);             \
        }                                                                                                           \
    }
    if (D == 128) QL_ATT(128)
    else if (D == 64) QL_ATT(64)
    else if (D == 32) QL_ATT(32)
    else return QL_ERR_UNSUPPORTED;
#undef QL_ATT
    return finish_launch();
}

int decode_attention(int dtype, const void* Q, const void* Kc, const void* Vc, const float* mask, void* Out, int64_t B,
                     int64_t H, int64_t G, int64_t D, int64_t capacity, hipStream_t st) {
    QL_DT(dtype, return (launch_attention<T, false>(Q, (void*)Kc, (void*)Vc, mask, Out, B, H, G, D, capacity, nullptr, nullptr,
                                                    nullptr, 0, nullptr, Prefetch{}, st)))
    return QL_ERR_BAD_DTYPE;
}

size_t decode_attention_split_bytes(int64_t B, int64_t H, int64_t D, int64_t capacity) {
    return (size_t)(B * H * ((capacity + kAttnWindow - 1) / kAttnWindow) * (D + 2)) * sizeof(float);
}

int decode_attention_rope(int dtype, const void* QKV, const void* table, const int64_t* pos, const int64_t* widx, void* Kc,
                          void* Vc, const float* mask, void* Out, int64_t B, int64_t H, int64_t G, int64_t D,
                          int64_t capacity, int64_t ldqkv, float* split_ws, const Prefetch& pf, hipStream_t st) {
    QL_DT(dtype, return (launch_attention<T, true>(QKV, Kc, Vc, mask, Out, B, H, G, D, capacity, table, pos, widx, ldqkv, split_ws,
                                                   pf, st)))
    return QL_ERR_BAD_DTYPE;
}

int greedy_advance(int dtype, const void* logits, int64_t B, int64_t N, int64_t ldl, int64_t* tok, int64_t* write_index,
                   int64_t* pos, float* mask, int64_t capacity, hipStream_t st) {
    QL_DT(dtype, (greedy_advance_kernel<T><<<(unsigned)B, 1024, 0, st>>>((const T*)logits, (int)N, ldl, tok, write_index, pos, mask,
                                                                          (int)capacity)))
    return finish_launch();
}

int masked_softmax(int dtype, const void* Sc, const float* mask, void* P, int64_t rows, int64_t Tn, int64_t mask_rows, int64_t lds,
                   int64_t ldm, int64_t ldp, hipStream_t st) {
    QL_DT(dtype, (masked_softmax_kernel<T><<<(unsigned)((rows + 3) / 4), 256, 0, st>>>((const T*)Sc, mask, (T*)P, rows, (int)Tn,
                                                                                       (int)mask_rows, lds, ldm, ldp)))
    return finish_launch();
}

int silu_mul(int dtype, const void* In, void* Out, int64_t rows, int64_t hidden, int64_t ldin, int64_t ldo, hipStream_t st) {
    dim3 grid((unsigned)((hidden / 8 + 255) / 256), (unsigned)rows);
    QL_DT(dtype, (silu_mul_kernel<T><<<grid, 256, 0, st>>>((const T*)In, (T*)Out, (int)hidden, ldin, ldo)))
    return finish_launch();
}

}  // namespace ql

#ifdef QL_ATT_STAMPS
extern "C" int qlinear_att_stamps_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ql::ql_att_stamps), sizeof(unsigned long long) * 32);
}
#endif
