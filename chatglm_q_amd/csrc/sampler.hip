// The reference's decoding mode - top-k / top-p sampling (chatglm_q/decoder.py:12-27, called by generate() at :85) - as ONE launch
// behind lm_head, so that the sampled step can live in the captured HIP graph like the greedy one (greedy_advance_kernel):
//
//     probs = softmax(logits.float() / temperature)              fp32
//     probs, indices = sort(probs, descending)[:top_k]           ties: lowest index first (a stable sort)
//     probs[(cumsum(probs) - probs) > top_p] = 0
//     probs /= probs.sum()
//     token = indices[multinomial(probs, 1)]
//
// One workgroup of 1024 threads per logits row, which is walked twice (the second time from L2): (1) scale / maxima, (2) exp-sum +
// collection of every value >= a first threshold into LDS.  No sort of the row, nothing of it kept in registers (64 values per thread
// at 1024 threads per workgroup = the whole register budget: hipcc spilled 66 registers per lane):
//   * first threshold t0 = min over G >= top_k lane groups of the group's maximum: at least G distinct values are >= t0, and for
//     logits in no particular order only ~G ln G are (a few hundred of 65 024);
//   * the k-th largest of THOSE is found by a histogram select in ordered-key space on the LDS list (256 bins between the bounds,
//     repeated on the crossing bin until at most top_k + 28 values are left or the bin is one key wide);
//   * the finalists are ranked by counting ((value desc, index asc) is a strict total order), the top k land sorted in LDS;
//   * one wave does cumsum / top-p cut / renormalisation and one inverse-CDF draw from a counter-based generator (Philox4x32-10,
//     key = seed, counter = the row's own draw counter in device memory: a replayed graph draws fresh numbers).
// Rows that defeat the first threshold (sorted logits, most of the row at -inf, huge tie groups: more than 2048 values >= t0) take
// an exact bisection instead, one walk over the row per step - slow (tens of microseconds) and still exact, ties by lowest index.
// The bookkeeping of a decode step (token -> tok, pos += 1, write_index += 1, unmask the new position) is greedy_advance_kernel's.
#include "launch.h"
#include "ql_common.h"

// QL_SAMPLER_KEEP=1 (A/B builds only): 2-byte rows stay in registers between the two walks (walk_regs) instead of being read twice.
// Measured slower, not faster (profiles/r06_sampler_timeline.txt): 64 raw halves + the walk's working set do not fit the 128 registers a
// 1024-thread workgroup leaves a lane - hipcc spills 20 - 160 registers into the second walk (22.9 - 48 us against 18.1 for two reads).
#ifndef QL_SAMPLER_KEEP
#define QL_SAMPLER_KEEP 0
#endif

namespace ql {
namespace {

constexpr int SB = 1024;        // threads per workgroup
constexpr int SU = 8;           // 16-byte chunks a thread requests before it uses the first
constexpr int S_CAP = 2048;     // candidates the selection phases handle (more: the bisection path)
constexpr int S_SEG = 128;      // the first collection writes wave-private LDS segments of S_SEG entries (16 x 128 = 2048 slots)
constexpr int S_KMAX = 1024;    // largest top_k served
constexpr int S_SLACK = 28;     // refinement stops at top_k + S_SLACK finalists

struct SamplerShared {
    float cx[16 * S_SEG];       // candidates: every value >= the first threshold, unordered: wave w's at [S_SEG w, S_SEG w + wn[w]),
    int ci[16 * S_SEG];         // or (bisection path) contiguous from 0
    u32 fk[S_CAP];              // finalists (ordered key >= the refined threshold), unordered
    int fi[S_CAP];
    float sx[S_KMAX];           // the top k in order
    int si[S_KMAX];
    int hist[256];
    float wmax[16], wmin[16], wsum[16];
    int wcnt[16], wn[16];
    int cnt, fcnt, cross_bin, cross_above;
};

// float -> u32 with the same order (and -0 == +0); NaN sorts above +inf, where torch.sort(descending) puts it
__device__ __forceinline__ u32 okey(float x) {
    x += 0.0f;
    const u32 b = f32_as_u32(x);
    return b ^ ((u32)((int)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float okey_inv(u32 k) { return u32_as_f32((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }

__device__ __forceinline__ float wave_min_dpp(float v) { return -wave_max_dpp(-v); }
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// sum over the block, result in every thread (two barriers: the scratch may be reused right away)
__device__ __forceinline__ int block_sum_i(int v, int* wcnt) {
    v = wave_sum_i(v);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += wcnt[w];
    __syncthreads();
    return t;
}

// Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1) -> first output word
__device__ __forceinline__ u32 philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const u32 hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const u32 hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const u32 n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

template <typename T>
__device__ __forceinline__ void load_chunk8(const T* p, float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        unpack8<T>(*reinterpret_cast<const u32x4*>(p), v);
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    }
}

// x = logit / temperature in fp32 (chatglm_q/decoder.py:14): q = l r, one residual correction (r = 1 / T rounded once): the quotient
// rounded to nearest but for rare half-ulp cases, at 3 instructions per logit instead of the ~10 of the full division sequence
struct Scale {
    float t, r;
    bool on;
    template <bool ON>
    __device__ __forceinline__ float apply(float l) const {
        if constexpr (!ON) return l;
        const float q = l * r;
        const float y = __builtin_fmaf(__builtin_fmaf(-t, q, l), r, q);
        return (y == y) ? y : q;                             // infinities: the residual is inf - inf
    }
};

// One walk over a logits row by the whole workgroup: chunk c (logits 8 c .. 8 c + 7) belongs to thread c % 1024 of round c / 1024;
// SU chunks per thread are requested before the first is used.  f(c, v): the chunk's scaled values, -inf past the row's end.
template <typename T, bool SCALED, typename F>
__device__ __forceinline__ void walk_row(const T* __restrict__ row, int N, const Scale& sc, F f) {
    const int nch = (N + 7) >> 3, tid = threadIdx.x;
    const bool vec = (reinterpret_cast<uintptr_t>(row) & 15) == 0;
    if constexpr (sizeof(T) == 2) {
        if (vec) {
            // 16-bit logits, aligned row: the requests in flight are held as the raw 16-byte words (32 registers, not 64 converted floats:
            // with those the kernel spilled) and converted chunk by chunk; a chunk that straddles the row's end is read element-wise
            const int nfull = N >> 3;
            for (int c0 = 0; c0 < nch; c0 += SU * SB) {
                u32x4 raw[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int c = c0 + u * SB + tid;
                    raw[u] = *reinterpret_cast<const u32x4*>(row + (size_t)(c < nfull ? c : (nfull > 0 ? nfull - 1 : 0)) * 8);
                }
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int c = c0 + u * SB + tid;
                    if (c < nch) {
                        float v[8];
                        if (c < nfull) {
                            unpack8<T>(raw[u], v);
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = c * 8 + e < N ? Act<T>::load(row + c * 8 + e) : -INFINITY;
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = sc.template apply<SCALED>(v[e]);
                        f(c, v);
                    }
                }
            }
            return;
        }
    }
    for (int c0 = 0; c0 < nch; c0 += SU * SB) {
        float v[SU][8];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int c = c0 + u * SB + tid;
            if (vec && c * 8 + 8 <= N) {
                load_chunk8<T>(row + c * 8, v[u]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][e] = c * 8 + e < N ? Act<T>::load(row + c * 8 + e) : -INFINITY;
            }
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int c = c0 + u * SB + tid;
            if (c < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][e] = sc.template apply<SCALED>(v[u][e]);
                f(c, v[u]);
            }
        }
    }
}

// The row held in registers as the raw 16-bit words (KEEP form: 2-byte logits, N <= 65 536, N % 8 == 0, 16-byte aligned rows): 8 x 16 bytes
// per thread, converted again on every walk - one memory pass per launch instead of two (a CU streams a 130 KB row at ~40 GB/s: 3.3 us a pass).
template <typename T, bool SCALED, typename F>
__device__ __forceinline__ void walk_regs(const u32x4 (&raw)[SU], int N, const Scale& sc, F f) {
    static_assert(sizeof(T) == 2, "16-bit logits");
    const int nch = N >> 3, tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < SU; ++u) {
        const int c = u * SB + tid;
        if (c < nch) {
            float v[8];
            unpack8<T>(raw[u], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = sc.template apply<SCALED>(v[e]);
            f(c, v);
        }
    }
}

// developer build only (make strace; tools/sampler_timeline.py): thread 0 stamps the 100 MHz clock at the phase boundaries into u_out[2..]
#ifdef QL_SAMPLER_TRACE
#define S_STAMP(i) do { if (threadIdx.x == 0 && u_out) reinterpret_cast<unsigned long long*>(u_out)[1 + (i)] = wall_clock64(); } while (0)
#else
#define S_STAMP(i) do { } while (0)
#endif

template <typename T, bool KEEP>
__global__ __launch_bounds__(SB) void top_p_sample_kernel(const T* __restrict__ logits, int N, int64_t ldl, int top_k_h, float top_p_h,
                                                          float temperature_h, const float* __restrict__ dparams,
                                                          unsigned long long* __restrict__ rng, int64_t* __restrict__ tok,
                                                          int64_t* __restrict__ write_index, int64_t* __restrict__ pos,
                                                          float* __restrict__ mask, int capacity, float* __restrict__ probs_out,
                                                          int64_t* __restrict__ index_out, float* __restrict__ u_out, int out_ld) {
    __shared__ SamplerShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const T* row = logits + (int64_t)blockIdx.x * ldl;

    // What the LAST statements of the launch need from memory - the generator's state, the position counters - is requested by thread 0
    // NOW: fetched at the end each of these was a global round trip (~1 us) in the launch's tail (the first version's final wave: 2.1 us).
    unsigned long long seed = 0ull, ctr = 0ull;
    int64_t pos_old = 0, widx_old = 0;
    if (tid == 0) {
        if (rng) { seed = rng[0]; ctr = rng[1 + blockIdx.x]; }
        if (pos) pos_old = pos[blockIdx.x];
        if (write_index && blockIdx.x == 0) widx_old = write_index[0];
    }
    // parameters: device-resident ones (a captured graph serves any setting) override the launch's
    int k = top_k_h;
    float top_p = top_p_h, temperature = temperature_h;
    if (dparams) { k = (int)dparams[0]; top_p = dparams[1]; temperature = dparams[2]; }
    k = k < 1 ? 1 : k;
    k = k > S_KMAX ? S_KMAX : k;
    k = k > N ? N : k;
    const Scale sc = {temperature, 1.0f / temperature, temperature != 1.0f};

    S_STAMP(0);
    // ---- pass 1: thread maxima -> the row's maximum M and the first threshold t0 --------------------------------------------------
    u32x4 raw[KEEP ? SU : 1];
    if constexpr (KEEP) {
        const int nch = N >> 3;
#pragma unroll
        for (int u = 0; u < SU; ++u) {                       // every request before the first use; clamped, not predicated
            const int c = u * SB + tid;
            raw[u] = *reinterpret_cast<const u32x4*>(row + (c < nch ? c : nch - 1) * 8);
        }
    }
    auto walk = [&](auto f) {                                // temperature == 1 (the reference's default): no per-logit scaling code at all
        if (sc.on) {
            if constexpr (KEEP) walk_regs<T, true>(raw, N, sc, f);
            else walk_row<T, true>(row, N, sc, f);
        } else {
            if constexpr (KEEP) walk_regs<T, false>(raw, N, sc, f);
            else walk_row<T, false>(row, N, sc, f);
        }
    };
    float tm = -INFINITY;
    walk([&](int, const float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) tm = fmaxf(tm, v[e]);
    });
    // First threshold t0: a value with at least k logits >= it, as tight as one histogram round makes it.  The 1024 thread maxima are 1024
    // DISTINCT logits, so any t with at least k of THEM >= t qualifies; t = the lower edge of the bin in which the count of thread maxima,
    // taken from the top, crosses k (256 bins in ordered-key space between the smallest thread maximum and the row maximum).  For logits in
    // no particular order about k + k^2 / 2048 + (a bin's worth) logits are >= t0 (105 - 110 of 65 024 at k = 100): the second walk then
    // almost never leaves its arithmetic, and the selection behind it usually has nothing left to refine.  (First version: the minimum over
    // k-ish lane groups of the group maximum - ~k ln k candidates, 700 at k = 100: pass 2 took the append branch on every other element.)
    {
        const float wmx = wave_max_dpp(tm), wmn = wave_min_dpp(tm);
        if (lane == 0) { sh.wmax[wv] = wmx; sh.wmin[wv] = wmn; }
        if (tid < 256) sh.hist[tid] = 0;
        if (tid == 0) sh.cnt = 0;
    }
    __syncthreads();
    float M = sh.wmax[0], tmin = sh.wmin[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) { M = fmaxf(M, sh.wmax[w]); tmin = fminf(tmin, sh.wmin[w]); }
    float t0;
    {
        const u32 lo0 = okey(tmin), span = okey(M) - lo0;
        const int shift = span < 256u ? 0 : (32 - __builtin_clz(span)) - 8;
        atomicAdd(&sh.hist[(okey(tm) - lo0) >> shift], 1);
        __syncthreads();
        if (wv == 0) {                                       // suffix sums over the bins, lane l owns bins 4 l .. 4 l + 3 (as in the selection below)
            const int h0 = sh.hist[4 * lane], h1 = sh.hist[4 * lane + 1], h2 = sh.hist[4 * lane + 2], h3 = sh.hist[4 * lane + 3];
            const int mine = h0 + h1 + h2 + h3;
            int suf = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_down(suf, off, 64);
                if (lane + off < 64) suf += o;
            }
            const int after = suf - mine, s3 = after + h3, s2 = s3 + h2, s1 = s2 + h1, s0 = s1 + h0;
            if (s0 >= k && after < k) sh.cross_bin = 4 * lane + (s3 >= k ? 3 : s2 >= k ? 2 : s1 >= k ? 1 : 0);
        }
        __syncthreads();
        t0 = okey_inv(lo0 + ((u32)sh.cross_bin << shift));
    }

    S_STAMP(1);
    // ---- pass 2 (the row again, from L2): softmax denominator + every value >= t0 into the LDS list -----------------------------
    constexpr float L2E = 1.4426950408889634f;
    const float mb = -M * L2E;                               // exp(x - M) = 2^(x L2E - M L2E): one fma + v_exp_f32; the rounding of M L2E
    float z = 0.f;                                           // is one common factor of every term and leaves p = e / Z alone
    // collection without atomics: a wave appends to ITS segment of the list - position = the wave's running count (an SGPR) + the number of
    // hitting lanes below this one (v_mbcnt of the compare's own mask); one counter for 700 hits serialised them (~3 us of the first version)
    int wbase = 0;
    walk([&](int c, const float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            z += __builtin_amdgcn_exp2f(__builtin_fmaf(v[e], L2E, mb));
            const bool hit = v[e] >= t0 && (KEEP || c * 8 + e < N);
            const unsigned long long m = __ballot(hit);
            if (m) {                                         // wave-uniform
                if (hit) {
                    const int p = wbase + (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                    if (p < S_SEG) { sh.cx[wv * S_SEG + p] = v[e]; sh.ci[wv * S_SEG + p] = c * 8 + e; }
                }
                wbase += __popcll(m);
            }
        }
    });
    z = wave_sum(z);
    if (lane == 0) { sh.wsum[wv] = z; sh.wn[wv] = wbase < S_SEG ? wbase : S_SEG; sh.wcnt[wv] = wbase; }
    __syncthreads();
    float Z = 0.f;
    int cnt = 0;
    bool seg_overflow = false;
#pragma unroll
    for (int w = 0; w < 16; ++w) { Z += sh.wsum[w]; cnt += sh.wcnt[w]; seg_overflow |= sh.wcnt[w] > S_SEG; }
    bool contig = false;                                     // list layout: wave segments, or (bisection path) contiguous from 0
    __syncthreads();                                         // wcnt is scratch of the bisection path
    S_STAMP(2);
    u32 lo = okey(t0);
    const u32 kmax = okey(M);

    // ---- the rare rows: more than S_CAP values >= t0.  Exact bisection, one walk over the row per step ---------------------------
    if (cnt > S_CAP || seg_overflow) {                       // uniform
        contig = true;
        u32 hi = kmax;                                       // invariant: count(key >= lo) >= k > count(key > hi)
        int c_lo = cnt;                                      // (a full segment with a small total: no bisection, the list is collected again, contiguous)
        // A masked row (constrained decoding: most logits at -inf) lands here with lo = key(-inf) every token: ask first whether fewer than k
        // logits are finite - then the k-th value IS -inf and the 31 bisection steps between -inf and the maximum (4 us each) are one
        if (c_lo > S_CAP && lo == okey(-INFINITY) && lo < hi) {
            int n = 0;
            walk([&](int c, const float (&v)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) n += (v[e] > -INFINITY && c * 8 + e < N) ? 1 : 0;
            });
            n = block_sum_i(n, sh.wcnt);
            if (n < k) hi = lo;                              // ties at -inf decide: the compaction below
            else { lo = lo + 1; c_lo = n; }                  // every finite value is >= key(-inf) + 1
        }
        while (lo < hi && c_lo > S_CAP) {
            const u32 mid = lo + ((hi - lo - 1) >> 1) + 1;   // lo < mid <= hi
            int n = 0;
            walk([&](int c, const float (&v)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) n += (okey(v[e]) >= mid && c * 8 + e < N) ? 1 : 0;
            });
            n = block_sum_i(n, sh.wcnt);
            if (n >= k) { lo = mid; c_lo = n; } else hi = mid - 1;
        }
        __syncthreads();
        if (tid == 0) sh.cnt = 0;
        __syncthreads();
        if (c_lo <= S_CAP) {                                 // every value >= lo fits the list
            walk([&](int c, const float (&v)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (okey(v[e]) >= lo && c * 8 + e < N) {
                        const int p = atomicAdd(&sh.cnt, 1);
                        sh.cx[p] = v[e]; sh.ci[p] = c * 8 + e;
                    }
            });
            __syncthreads();
            cnt = sh.cnt;
        } else {                                             // lo == hi: the k-th value itself, tied more than S_CAP times
            int gt = 0;                                      // everything above it (fewer than k) ...
            walk([&](int c, const float (&v)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (okey(v[e]) > lo && c * 8 + e < N) {
                        const int p = atomicAdd(&sh.cnt, 1);
                        sh.cx[p] = v[e]; sh.ci[p] = c * 8 + e;
                        ++gt;
                    }
            });
            gt = block_sum_i(gt, sh.wcnt);
            const int need_t = k - gt;                       // ... plus that many of the ties, lowest indices first: rounds of 1024
            const int nch = (N + 7) >> 3;                    // chunks in thread order are index order
            int base = 0;
            for (int c0 = 0; c0 < nch && base < need_t; c0 += SB) {
                const int c = c0 + tid;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = c * 8 + e < N ? sc.template apply<true>(Act<T>::load(row + c * 8 + e)) : -INFINITY;
                int n = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) n += (okey(v[e]) == lo && c * 8 + e < N) ? 1 : 0;
                int inc = n;                                 // inclusive scan over the wave, then over the waves
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_up(inc, off, 64);
                    if (lane >= off) inc += o;
                }
                if (lane == 63) sh.wcnt[wv] = inc;
                __syncthreads();
                int before = base, total = 0;
#pragma unroll
                for (int w = 0; w < 16; ++w) { before += w < wv ? sh.wcnt[w] : 0; total += sh.wcnt[w]; }
                int p = before + inc - n;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (okey(v[e]) == lo && c * 8 + e < N) {
                        if (p < need_t) { sh.cx[gt + p] = v[e]; sh.ci[gt + p] = c * 8 + e; }
                        ++p;
                    }
                base += total;
                __syncthreads();
            }
            cnt = k;
        }
    }

    S_STAMP(3);
    // ---- histogram select on the list: the k-th largest key --------------------------------------------------------------------
    {
        u32 hi = kmax;
        int need = k, above = 0;
        u32 tk = lo;
        int F = cnt;
        while (cnt > k + S_SLACK) {                           // (a list this short is ranked as it is) every quantity that steers the loop is block-uniform
            const u32 span = hi - lo;
            const int shift = span < 256u ? 0 : (32 - __builtin_clz(span)) - 8;
            if (tid < 256) sh.hist[tid] = 0;
            __syncthreads();
            for (int p = tid; p < 16 * S_SEG; p += SB) {
                if (!(contig ? p < cnt : (p & (S_SEG - 1)) < sh.wn[p / S_SEG])) continue;
                const u32 key = okey(sh.cx[p]);
                if (key >= lo && key <= hi) atomicAdd(&sh.hist[(key - lo) >> shift], 1);
            }
            __syncthreads();
            if (wv == 0) {                                    // suffix sums S(b) = hist[b] + hist[b + 1] + ...: lane l owns bins 4 l .. 4 l + 3
                const int h0 = sh.hist[4 * lane], h1 = sh.hist[4 * lane + 1], h2 = sh.hist[4 * lane + 2], h3 = sh.hist[4 * lane + 3];
                const int mine = h0 + h1 + h2 + h3;
                int suf = mine;                               // inclusive suffix scan over the lanes
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_down(suf, off, 64);
                    if (lane + off < 64) suf += o;
                }
                const int after = suf - mine;                 // bins right of mine
                const int s3 = after + h3, s2 = s3 + h2, s1 = s2 + h1, s0 = s1 + h0;
                // the crossing bin b*: the highest b with S(b) >= need (S does not increase with b; S(0) >= need)
                if (s0 >= need && after < need) {
                    int b, ab;
                    if (s3 >= need) { b = 3; ab = after; }
                    else if (s2 >= need) { b = 2; ab = s3; }
                    else if (s1 >= need) { b = 1; ab = s2; }
                    else { b = 0; ab = s1; }
                    sh.cross_bin = 4 * lane + b;
                    sh.cross_above = ab;                      // S(b* + 1)
                }
            }
            __syncthreads();
            const int b = sh.cross_bin, ab = sh.cross_above, hb = sh.hist[b];
            tk = lo + ((u32)b << shift);
            F = above + ab + hb;
            if (F <= k + S_SLACK || shift == 0) break;
            above += ab;
            need -= ab;
            const unsigned long long top = (unsigned long long)tk + ((1ull << shift) - 1ull);
            lo = tk;
            hi = top < (unsigned long long)hi ? (u32)top : hi;
            __syncthreads();                                  // hist / cross_* are rewritten by the next round
        }

        S_STAMP(4);
        // ---- finalists (key >= tk) compacted, ranked by counting; the top k land in order ---------------------------------------
        if (tid == 0) sh.fcnt = 0;
        __syncthreads();
        for (int p = tid; p < 16 * S_SEG; p += SB) {             // uniform trip count; one counter update per wave and round
            const bool valid = contig ? p < cnt : (p & (S_SEG - 1)) < sh.wn[p / S_SEG];
            const u32 key = valid ? okey(sh.cx[p]) : 0u;
            const bool hit = valid && key >= tk;
            const unsigned long long m = __ballot(hit);
            if (m) {
                int base = 0;
                if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&sh.fcnt, (int)__popcll(m));
                base = __shfl(base, (int)__builtin_ctzll(m), 64);
                if (hit) {
                    const int q = base + (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                    sh.fk[q] = key; sh.fi[q] = sh.ci[p];
                }
            }
        }
        __syncthreads();
        F = sh.fcnt;
        int ltpe = 0;                                         // log2 of the threads per finalist (groups stay inside a wave)
        while (ltpe < 6 && ((2 * F) << ltpe) <= SB) ++ltpe;
        const int tpe = 1 << ltpe, per_round = SB >> ltpe;
        for (int e0 = 0; e0 < F; e0 += per_round) {           // uniform trip count
            const int ent = e0 + (tid >> ltpe), part = tid & (tpe - 1);
            int r = 0;
            u32 myk = 0;
            int myi = 0;
            if (ent < F) {
                myk = sh.fk[ent]; myi = sh.fi[ent];
                for (int j = part; j < F; j += tpe) {
                    const u32 kj = sh.fk[j];
                    const int ij = sh.fi[j];
                    r += (kj > myk || (kj == myk && ij < myi)) ? 1 : 0;
                }
            }
            for (int off = 1; off < tpe; off <<= 1) r += __shfl_xor(r, off, 64);
            if (ent < F && part == 0 && r < k) { sh.sx[r] = okey_inv(myk); sh.si[r] = myi; }
        }
        __syncthreads();
    }

    S_STAMP(5);
    // ---- one wave: probabilities, cumulative sums, top-p cut, renormalisation, the draw ---------------------------------------------
    if (wv == 0) {
        const int per = (k + 63) >> 6;                        // lane l owns sorted entries per * l .. per * l + per - 1
        float p[S_KMAX / 64], c[S_KMAX / 64];
        float run = 0.f;
#pragma unroll
        for (int i = 0; i < S_KMAX / 64; ++i) {
            p[i] = 0.f;
            if (i < per) {                                    // uniform
                const int j = per * lane + i;
                if (j < k) p[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sh.sx[j], L2E, mb)) / Z;
                run += p[i];
            }
            c[i] = run;
        }
        float inc = run;                                      // cumsum(probs): the lanes' totals scanned, the lane's own run added
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        const float excl = inc - run;
        float kept = 0.f;
        int nkept = 0;
#pragma unroll
        for (int i = 0; i < S_KMAX / 64; ++i) {
            const int j = per * lane + i;
            c[i] += excl;
            const bool keep = (i < per && j < k) && !((c[i] - p[i]) > top_p);
            if (!keep) p[i] = 0.f;
            kept += p[i];
            nkept += keep ? 1 : 0;
        }
        const float s = wave_sum(kept);
        nkept = wave_sum_i(nkept);                            // the kept entries are a prefix of the order (cumsum - p does not decrease)
        // one uniform draw; the row's own counter advances by one per launch
        float uni = 0.f;
        if (lane == 0) {
            if (rng) rng[1 + blockIdx.x] = ctr + 1ull;
            const u32 r = philox4x32_10((u32)ctr, (u32)(ctr >> 32), blockIdx.x, 0u, (u32)seed, (u32)(seed >> 32));
            uni = (float)(r >> 8) * 0x1p-24f;
        }
        uni = __shfl(uni, 0, 64);
        const float target = uni * s;                         // inverse CDF on the un-normalised kept mass: entry j with cum(j-1) <= u s < cum(j)
        int jstar = 0x7fffffff;
#pragma unroll
        for (int i = S_KMAX / 64 - 1; i >= 0; --i) {
            const int j = per * lane + i;
            if (p[i] > 0.f && c[i] > target) jstar = j;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_xor(jstar, off, 64);
            jstar = o < jstar ? o : jstar;
        }
        if (jstar >= nkept) jstar = nkept > 0 ? nkept - 1 : 0;   // u s rounded up to the whole mass: the last kept entry
        if (probs_out) {
#pragma unroll
            for (int i = 0; i < S_KMAX / 64; ++i)
                if (i < per) {
                    const int j = per * lane + i;
                    if (j < k) {
                        probs_out[(int64_t)blockIdx.x * out_ld + j] = p[i] / s;
                        index_out[(int64_t)blockIdx.x * out_ld + j] = sh.si[j];
                    }
                }
        }
        S_STAMP(6);
        if (lane == 0) {
            if (u_out) u_out[blockIdx.x] = uni;
            tok[blockIdx.x] = sh.si[jstar];
            if (pos) pos[blockIdx.x] = pos_old + 1;
            if (write_index && blockIdx.x == 0) {             // the shared write index and every row's mask: one writer (greedy_advance_kernel)
                const int64_t nw = widx_old + 1;
                write_index[0] = nw;
                if (mask && nw < capacity)
                    for (int b = 0; b < (int)gridDim.x; ++b) mask[(int64_t)b * capacity + nw] = 0.f;
            }
        }
    }
}

}  // namespace

int top_p_sample(int dtype, const void* logits, int64_t B, int64_t N, int64_t ldl, int64_t top_k, float top_p, float temperature,
                 const float* dparams, uint64_t* rng_state, int64_t* tok, int64_t* write_index, int64_t* pos, float* mask,
                 int64_t capacity, float* probs_out, int64_t* index_out, float* u_out, int64_t out_ld, hipStream_t st) {
    if (!logits || !tok) return QL_ERR_NULL_POINTER;
    if (B < 1 || N < 1 || ldl < N) return QL_ERR_BAD_SHAPE;
    if (N > (int64_t)1 << 30) return QL_ERR_UNSUPPORTED;
    if (!dparams && (top_k < 1 || !(temperature > 0.f))) return QL_ERR_BAD_SHAPE;
    if (!dparams && top_k > S_KMAX && N > S_KMAX) return QL_ERR_UNSUPPORTED;
    if ((probs_out == nullptr) != (index_out == nullptr)) return QL_ERR_BAD_SHAPE;
    // (A/B builds) the row in registers between the two walks: 2-byte logits, whole 16-byte chunks, aligned rows, at most 8 chunks per thread
    [[maybe_unused]] const bool keep = QL_SAMPLER_KEEP && dtype != QL_DTYPE_F32 && N % 8 == 0 && N <= 8 * SB * SU && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && (ldl % 8) == 0;
    switch (dtype) {
#define QL_SAMPLE(TT, KEEP)                                                                                                        \
    top_p_sample_kernel<TT, KEEP><<<(unsigned)B, SB, 0, st>>>((const TT*)logits, (int)N, ldl, (int)(top_k > S_KMAX ? S_KMAX : top_k), top_p, \
                                                              temperature, dparams, (unsigned long long*)rng_state, tok, write_index, pos, \
                                                              mask, (int)capacity, probs_out, index_out, u_out, (int)out_ld)
    case QL_DTYPE_F32: QL_SAMPLE(float, false); break;
#if QL_SAMPLER_KEEP
    case QL_DTYPE_F16: if (keep) QL_SAMPLE(f16, true); else QL_SAMPLE(f16, false); break;
    case QL_DTYPE_BF16: if (keep) QL_SAMPLE(__bf16, true); else QL_SAMPLE(__bf16, false); break;
#else
    case QL_DTYPE_F16: QL_SAMPLE(f16, false); break;
    case QL_DTYPE_BF16: QL_SAMPLE(__bf16, false); break;
#endif
#undef QL_SAMPLE
    default: return QL_ERR_BAD_DTYPE;
    }
    return finish_launch();
}

}  // namespace ql
