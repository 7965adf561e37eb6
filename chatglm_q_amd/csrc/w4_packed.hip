// int4g32 decode-shape GEMV on the derived streaming layout (gfx950 / MI355X).
//
// Layout ("packed", built once per weight by w4_repack_kernel from the canonical buffers):
//   weights : Wt[n][g] = one 16-byte unit holding the 32 nibbles of column n, group g; n padded to
//             a multiple of 4.  Word j of the unit holds k = 8j .. 8j+7 at nibble positions
//             p(kk) = (kk >> 1) + 4 (kk & 1), i.e. nibbles (p, p+4) are the k-adjacent pair
//             (2p, 2p+1): one v_and_or_b32 yields a k-pair ready for v_dot2c_f32_f16.
//   scales  : Sp[t][g][c], t = n / 4, c = n % 4 (one 8-byte load per lane per group for fp16).
//
// Work decomposition: one WAVE owns 4 output columns for ALL of K; lane l of the wave walks the
// groups l, l+64, ... .  Every weight load instruction is 64 lanes x 16 B = 1 KiB contiguous, no
// partial sums ever leave the wave (6 cross-lane steps at the end), no cross-workgroup reduction.
// The activation row is shared by the block's 4 waves through LDS (16-byte chunks, XOR-swizzled so
// that the 64-byte-strided per-lane reads are bank-conflict free).
//
// Latency structure (the whole 1x4096->4096 problem is ~37 KB per CU, i.e. one HBM round trip):
// the loads of the first TWO group tiles are issued before anything else waits, later tiles are
// issued two iterations ahead of their use; activations are the OLDEST entries of the vector-memory
// queue so that their s_waitcnt does not drain the weight loads behind them.
#include <stdlib.h>

#include "launch.h"
#include "w4_dequant.h"
#include "w4_splice.h"

// how the one-row kernels request their weight units: QL_GEMV_NT 1 = non-temporal loads (rounds 1 - 5), 0 = default loads (round 6's
// pure-read sweep, bench.py roofline.floor.pure_read_sweep_us: default loads stream the same bytes 2 - 4 % faster than `nt` ones)
#ifndef QL_GEMV_NT
#define QL_GEMV_NT 1
#endif
#if QL_GEMV_NT
#define QL_GEMV_W_LOAD(p) __builtin_nontemporal_load(p)
#else
#define QL_GEMV_W_LOAD(p) (*(p))
#endif

namespace ql {

__device__ __forceinline__ int packed_pos(int kk) { return (kk >> 1) + 4 * (kk & 1); }

template <typename T>
__global__ __launch_bounds__(256) void w4_repack_kernel(const uint8_t* __restrict__ Wq, const T* __restrict__ S,
                                                        u32x4* __restrict__ Wt, T* __restrict__ Sp, int N, int Npad,
                                                        int G) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    if (n >= Npad) return;
    u32x4 out = {0x88888888u, 0x88888888u, 0x88888888u, 0x88888888u};   // q == 0 for padded columns
    float sc = 0.f;
    if (n < N) {
        u32 words[4] = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const u32 b = Wq[((int64_t)g * 16 + r) * N + n];
            const int k0 = 2 * r, k1 = 2 * r + 1;                         // k within the group
            words[k0 >> 3] |= (b & 0xFu) << (4 * packed_pos(k0 & 7));
            words[k1 >> 3] |= (b >> 4) << (4 * packed_pos(k1 & 7));
        }
        out = u32x4{words[0], words[1], words[2], words[3]};
        sc = Act<T>::load(S + (int64_t)g * N + n);
    }
    Wt[(int64_t)n * G + g] = out;
    Act<T>::store(Sp + ((int64_t)(n >> 2) * G + g) * 4 + (n & 3), sc);
}

// The inverse (round 5): part 1 -> the canonical buffers, byte for byte (the repack is a bijection on the N real columns: padded
// columns and their zero scales are dropped).  Lets a host that keeps only derived layouts resident serve state_dict() / checkpoints
// (chatglm_q/loader.py:90-104's buffer contract) - qlinear_w4g32_unpack_gemv.
template <typename T>
__global__ __launch_bounds__(256) void w4_unpack_kernel(const u32x4* __restrict__ Wt, const T* __restrict__ Sp, uint8_t* __restrict__ Wq,
                                                        T* __restrict__ S, int N, int G) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    if (n >= N) return;
    const u32x4 unit = Wt[(int64_t)n * G + g];
    const u32 words[4] = {unit[0], unit[1], unit[2], unit[3]};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k0 = 2 * r, k1 = 2 * r + 1;
        const u32 lo = (words[k0 >> 3] >> (4 * packed_pos(k0 & 7))) & 0xFu;
        const u32 hi = (words[k1 >> 3] >> (4 * packed_pos(k1 & 7))) & 0xFu;
        Wq[((int64_t)g * 16 + r) * N + n] = (uint8_t)(lo | (hi << 4));
    }
    S[(int64_t)g * N + n] = Sp[((int64_t)(n >> 2) * G + g) * 4 + (n & 3)];
}

// part 2 of the derived layout (launch.h): tile-major copy for the MFMA kernels, built from part 1
template <typename T>
__global__ __launch_bounds__(256) void w4_tile_kernel(const u32x4* __restrict__ Wt, const T* __restrict__ Sp,
                                                      u32x4* __restrict__ Wm, T* __restrict__ Sm, int Npad, int G,
                                                      int ksteps, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (ct * ksteps + kt) * 64 + lane
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int64_t step = idx >> 6;
    const int kt = (int)(step % ksteps), ct = (int)(step / ksteps);
    const int n = ct * 32 + (lane & 31), g = 2 * kt + (lane >> 5);
    u32x4 w = {0x88888888u, 0x88888888u, 0x88888888u, 0x88888888u};
    T sc = (T)0.f;
    if (n < Npad && g < G) {
        w = Wt[(int64_t)n * G + g];
        sc = Sp[((int64_t)(n >> 2) * G + g) * 4 + (n & 3)];
    }
    Wm[idx] = w;
    Sm[idx] = sc;
}

// ---------------------------------------------------------------------------------------------
// 16-bit activation kernel (fp16 / bf16)
// ---------------------------------------------------------------------------------------------
// Two arithmetic modes, selected per call (QL_FLAG_STRICT_ROUNDING):
//
//  strict: every dequantised weight (n - 8) * s is rounded to the activation dtype before it meets the
//      activation - bit-for-bit the reference's rounding sequence (chatglm_q/int4/triton_ops.py:72-73).
//      fp16, per 8 weights: 1 shift, 4 v_and_or, 4 exact offset removals, 4 v_pk_mul_f16, 4 v_dot2c.
//      bf16 (no packed bf16 multiply): exact fp32 product per weight, v_cvt_pk_bf16_f32, 4 v_dot2c per 8 weights.
//
//  exact-dequant (default): the offset form produced by the exponent splice is fed to v_dot2c AS IS
//      and the group's affine correction is applied once per (column, group) in fp32:
//          sum_k a_k (n_k - 8) s  =  s * u * ( sum_k a_k v_k  -  o * sum_k a_k ),   v_k = splice(n_k)
//      fp16: v = 0x6400 | (n << 4) = 16 (64 + n)   -> u = 1/16, o = 1152   (nibble at mantissa bits 4..7)
//      bf16: v = 0x4300 | n        = 128 + n       -> u = 1,    o = 136    (nibble at mantissa bits 0..3)
//      Products a_k v_k are exact in fp32; the cancellation costs < 2^-17 relative per group
//      (offset/step = 64 resp. 128), so this mode is CLOSER to real-number arithmetic than the
//      reference's per-weight fp16 rounding and differs from it by that rounding only (~2e-4 relative,
//      inside the 1e-3 tolerance).  Per 8 weights: 3 shifts, 4 v_and_or, 4 v_dot2c.
//
// LDS image of the activation rows: 16-byte chunk cc of row m (k = 8 cc .. 8 cc + 7) lives at chunk
// position  m * K/8 + 4 g + (j ^ ((g >> 2) & 3))  with g = cc >> 2 (its group), j = cc & 3.
// A lane reads the 4 chunks of ITS group (64-byte lane stride); the XOR spreads each 16-lane
// ds_read_b128 service group over all 16 four-bank slots.
template <int MB, bool A_LDS>
struct PackedTile16 {
    u32x4 w[4];
    u32x2 s;
    u32x4 a[A_LDS ? 1 : MB][A_LDS ? 1 : 4];   // activations only when they are NOT staged in LDS
};

// VAR is a developer knob (tools/microbench): 0 = product kernel; the other values strip parts of the
// kernel to attribute time.  Only VAR == 0 is instantiated unless QL_DEV_VARIANTS is defined.
//   1: no math (loaded words are XOR-folded), 2: math with a constant activation (no LDS reads),
//   4: as 2 and no activation staging at all (no staging loads, no LDS writes, no barrier)
//   5: as 1 and no activation staging at all (weight / scale loads, reduction and epilogue only)
//   6: as 5 plus the staging loads, LDS writes and barrier (staged data unused); 7: as 5 plus the staging LOADS only
// KS: the block's 4 waves cover 4/KS column quads x KS slices of K (combined through LDS at the end):
// shapes with few columns but a long K (w_out: 13696 -> 4096) get 4x the workgroups and 4x the loads
// in flight per column instead of one wave walking 7 tiles in sequence.
// PRO: activation prologue applied while the row is staged (decode step, one row), so that the small ops in
// front of a QLinear call cost no launch of their own.  Rounding sequence as the graph's:
//   PRO_SILU     the input row is (h | gate), 2K wide: staged = round(round(silu(h)) * gate)      model.py:200-201
//   PRO_ADDNORM  hnew = round(x + delta) (delta optional), written to pro.hout by block 0;
//                staged = round(round(hnew * rsqrt(mean(hnew^2) + eps)) * ln_weight)               model.py:62-73,243-245
// (enum PRO_* and struct Prologue: launch.h)

// QL_SPAN_PROBE (developer build `make span`, never in libqlinear_hip.so): every wave of the 16-bit GEMV stamps the
// constant 100 MHz s_memrealtime counter when it starts and when its sums are complete; min / max over the launch give
// the KERNEL's own span, without the dependent-launch boundary that HIP events and rocprofv3 durations contain
// ("boundary vs kernel" measured, not inferred).  Two plain 8-byte stores per wave into its own slot (a first
// version used atomicMin / atomicMax on two words: 2 048 contended 64-bit atomics made the 4 us kernel take 45 us).
#ifdef QL_SPAN_PROBE
constexpr int kSpanWaves = 1 << 16;                       // one (start, end) slot per wave: plain stores, no contention
__device__ unsigned long long ql_span_slots[2 * kSpanWaves];
#define QL_SPAN_SLOT() (((int)(by * gridDim.x + bx) * 4 + (int)(threadIdx.x >> 6)) & (kSpanWaves - 1))
#define QL_SPAN_START() do { if ((threadIdx.x & 63) == 0) ql_span_slots[2 * QL_SPAN_SLOT()] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#define QL_SPAN_END() do { if ((threadIdx.x & 63) == 0) ql_span_slots[2 * QL_SPAN_SLOT() + 1] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define QL_SPAN_START() do { } while (0)
#define QL_SPAN_END() do { } while (0)
#endif

// The kernel body is a device function taking its block coordinates as arguments, so that ONE launch can run two
// projections (w4_mlp_pair_kernel below: blocks [0, nA) = the first, the rest = the second, chained through counters).
// CHAIN: 0 none; 1 = this projection's outputs feed a chained consumer in the same launch (outputs stored, block arrives on
// chain->cnt); 2 = this projection's activation row is produced in the same launch (weight tiles requested first, then the
// block waits for the producers, then stages the row).
#ifndef QL_CHAIN_ABLATE
#define QL_CHAIN_ABLATE 0         // timing experiments (results wrong): 1 = consumers do not wait, 2 = no acquire fence
#endif
struct ChainArgs {
    unsigned* cnt;        // 64 slot counters, 64 bytes apart (cnt[16 * s]); done counter cnt[16 * 64]; error word cnt[16 * 65]; top counter cnt[16 * 66]
    int producers;        // blocks of the producing projection
    int consumers;        // blocks of the consuming projection (the last one to pass its wait resets the counters)
};
constexpr int kChainSlots = 64;
constexpr int kChainWords = 16 * (kChainSlots + 3);

template <typename T, int MB, int ACH, int KS, bool STRICT, int VAR = 0, int PRO = PRO_NONE, int CHAIN = 0>   // ACH: 16-byte A chunks staged per thread; 0 = A from global
__device__ __forceinline__ void w4_packed_gemv_16_body(const T* __restrict__ A, const u32x4* __restrict__ Wt,
                                                       const T* __restrict__ Sp, const void* pro_delta,
                                                       const void* pro_ln_weight, int N, int K, int M, int lda32,
                                                       const T* __restrict__ bias, T* __restrict__ C, int64_t ldc,
                                                       void* pro_hout, float pro_eps, int pro_gate,
                                                       const T* __restrict__ resid, const int bx, const int by,
                                                       const ChainArgs chain = ChainArgs{nullptr, 0, 0}) {
    // resid (one row, nullable): the residual stream the output is added to in the epilogue - out = round(y + resid),
    // y = rounded sum (+ bias) - so that the NEXT projection's RMSNorm prologue needs no delta operand
    // (chatglm_q/model.py:243,245: hidden = hidden + sublayer(...)).
    // Argument order: everything the first loads need sits in the leading 14 dwords, which are preloaded into SGPRs at
    // wave launch (-amdgpu-kernarg-preload-count, Makefile); the rest costs a scalar-load round trip that used to
    // sit in front of the first weight load (4096 x 4096: 4.25 -> 3.97 us).
    const int G = K >> 5;
    const int64_t lda = lda32;
    const Prologue pro{pro_delta, pro_ln_weight, pro_hout, pro_eps, pro_gate};
    static_assert(PRO == PRO_NONE || (MB == 1 && ACH > 0), "prologues exist for the one-row LDS-staged kernel");
    constexpr bool A_LDS = ACH > 0;
    typedef Splice<T> SP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QW = 4 / KS;                 // column quads per block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    QL_SPAN_START();
    const int ks = wave % KS;
    const int t_raw = bx * QW + wave / KS;
    const bool wave_active = t_raw * 4 < N;
    const int t = wave_active ? t_raw : 0;     // inactive waves shadow quad 0 (they must reach the barriers)
    const int m0 = by * MB;
    const int gs = (G + KS - 1) / KS;          // groups per K slice
    const int g_begin = ks * gs;
    const int g_end = min(G, g_begin + gs);
    const int iters = g_end > g_begin ? (g_end - g_begin + 63) >> 6 : 0;
    const int cpr = K >> 3;                    // 16-byte chunks per activation row

    // splice constants live in registers so that (w & mask) | magic is ONE v_and_or_b32
    // (two 32-bit literals cannot be encoded in one VOP3 instruction)
    u32 k_mask, k_mask_odd, k_magic;
    asm volatile("s_mov_b32 %0, %1" : "=s"(k_mask) : "i"(SP::kMask));
    asm volatile("s_mov_b32 %0, %1" : "=s"(k_mask_odd) : "i"(SP::kMaskOdd));
    asm volatile("v_mov_b32 %0, %1" : "=v"(k_magic) : "i"(SP::kMagic));

    const T* arow[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) arow[m] = A + (int64_t)((m0 + m < M) ? (m0 + m) : (M - 1)) * lda;

    // (1) activation staging loads first: oldest in the VM queue
    u32x4 areg[A_LDS ? ACH : 1];
    constexpr bool kNorm = PRO == PRO_ADDNORM || PRO == PRO_NORM;
    u32x4 xreg[PRO != PRO_NONE ? ACH : 1], yreg[PRO == PRO_ADDNORM ? ACH : 1];   // prologue operands
    auto issue_staging_loads = [&] {
        if constexpr (A_LDS && VAR != 4 && VAR != 5) {
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                // unconditional (clamped) load: a load under a per-element condition makes hipcc branch
                // around it and drain the queue (vmcnt(0)) per element
                const int c = min(tid + i * 256, MB * cpr - 1);
                const int m = MB == 1 ? 0 : c / cpr, cc = c - m * cpr;
                areg[i] = *reinterpret_cast<const u32x4*>(arow[m] + cc * 8);
                if constexpr (PRO == PRO_SILU) xreg[i] = *reinterpret_cast<const u32x4*>(arow[0] + K + cc * 8);
                if constexpr (kNorm) xreg[i] = *reinterpret_cast<const u32x4*>((const T*)pro.ln_weight + cc * 8);
                if constexpr (PRO == PRO_ADDNORM)
                    yreg[i] = *reinterpret_cast<const u32x4*>((pro.delta ? (const T*)pro.delta : arow[0]) + cc * 8);
            }
        }
    };
    if constexpr (CHAIN != 2) issue_staging_loads();

    const u32x4* wbase = Wt + (int64_t)t * 4 * G;
    const T* sbase = Sp + (int64_t)t * G * 4;

    auto load_tile = [&](int it) {
        PackedTile16<MB, A_LDS> tl;
        const int g = g_begin + it * 64 + lane;
        const int gc = g < g_end ? g : G - 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) tl.w[c] = QL_GEMV_W_LOAD(wbase + (int64_t)c * G + gc);
        tl.s = *reinterpret_cast<const u32x2*>(sbase + (int64_t)gc * 4);   // masked at use, not here:
                                                                           // touching it now would drain the queue
        if constexpr (!A_LDS) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) tl.a[m][j] = *reinterpret_cast<const u32x4*>(arow[m] + gc * 32 + 8 * j);
        }
        return tl;
    };

    // (2) two weight tiles in flight before anything waits
    PackedTile16<MB, A_LDS> t0 = load_tile(0);
    PackedTile16<MB, A_LDS> t1;
    if (iters > 1) t1 = load_tile(1);          // wave-uniform branch around the whole tile

    // Epilogue operands of this wave's column quad (bias, residual) requested behind the weight tiles: fetched at
    // the end they are a global round trip in the tail of every wave (the residual epilogue measured no gain that way);
    // in front of the tiles, their pointers - late kernel arguments - would put a scalar-load wait before the first
    // weight load of every wave (plain forward 3.97 -> 4.41 us).  Unconditional loads from a safe address; used only when the quad is whole and the rows are 8-byte aligned.
    const bool quad_early = MB == 1 && t * 4 + 3 < N && (((uintptr_t)bias | (uintptr_t)resid | (uintptr_t)C) & 7) == 0;
    const int tq = quad_early ? t * 4 : 0;
    u32x2 bias_q = {0u, 0u}, resid_q = {0u, 0u};
    if (bias || resid) {                                      // kernel-uniform: plain forwards issue no extra loads
        bias_q = *reinterpret_cast<const u32x2*>((bias && quad_early ? bias : (const T*)Sp) + tq);
        resid_q = *reinterpret_cast<const u32x2*>((resid && quad_early ? resid : (const T*)Sp) + tq);
    }


    if constexpr (CHAIN == 2) {
        // The activation row is being produced by lower-numbered blocks of this launch (dispatched before this one, so they
        // make progress whatever this block does).  This block's weight tiles are already in flight; wave 0 polls the 64
        // arrival counters (one per lane, relaxed agent-scope loads, s_sleep between polls, bounded), ONE acquire, then the
        // block stages the row.  The last consumer block to get here resets the counters for the next launch.
        if (wave == 0) {
            // ONE word is polled (by one lane): the top counter, which the last arriver of each of the 64 slot counters
            // increments (1 024 consumer blocks polling 64 words each slowed the producers down 5x: 105 us for the pair)
            const unsigned want = (unsigned)(chain.producers < kChainSlots ? chain.producers : kChainSlots);
            unsigned spins = 0;
            for (; !(QL_CHAIN_ABLATE & 1);) {
                const unsigned v = __hip_atomic_load(chain.cnt + 16 * (kChainSlots + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= want) break;
                if (++spins > (1u << 16)) {                       // tens of ms: give up loudly instead of hanging the GPU
                    if (lane == 0) __hip_atomic_store(chain.cnt + 16 * (kChainSlots + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
            }
            if (!(QL_CHAIN_ABLATE & 2)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (lane == 0) {
                const unsigned d = __hip_atomic_fetch_add(chain.cnt + 16 * kChainSlots, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (d == (unsigned)chain.consumers - 1) {
                    for (int i = 0; i <= kChainSlots + 2; ++i)
                        if (i != kChainSlots + 1) __hip_atomic_store(chain.cnt + 16 * i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
        issue_staging_loads();
    }
    // (3) stage the activations (waits only for the staging loads, which are older than the tiles)
    if constexpr (PRO == PRO_SILU) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            float h[8], g[8];
            unpack8<T>(areg[i], h);
            unpack8<T>(xreg[i], g);
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = Act<T>::round(h[e] / (1.0f + __expf(-h[e]))) * g[e];
            areg[i] = pack8<T>(h);
        }
    }
    if constexpr (kNorm) {
        float* nred = reinterpret_cast<float*>(smem + (((size_t)K * sizeof(T) + 15) & ~(size_t)15) + 4 * 4 * sizeof(float));
        float hv[ACH][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            unpack8<T>(areg[i], hv[i]);
            const int c = tid + i * 256;
            if constexpr (PRO == PRO_ADDNORM) {
                if (pro.delta) {
                    float d[8];
                    unpack8<T>(yreg[i], d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[i][e] = Act<T>::round(hv[i][e] + d[e]);
                }
            }
            if (c < cpr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(hv[i][e], hv[i][e], ss);
                if constexpr (PRO == PRO_ADDNORM) {
                    if (bx == 0 && pro.hout) *reinterpret_cast<u32x4*>((T*)pro.hout + c * 8) = pack8<T>(hv[i]);
                }
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) nred[wave] = ss;
        __syncthreads();
        const float r = rsqrtf(((nred[0] + nred[1]) + (nred[2] + nred[3])) / (float)K + pro.eps);
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            float w[8];
            unpack8<T>(xreg[i], w);
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[i][e] = Act<T>::round(hv[i][e] * r) * w[e];
            areg[i] = pack8<T>(hv[i]);
        }
    }
    if constexpr (A_LDS && VAR == 7) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) asm volatile("" ::"v"(areg[i][0]));   // keep the loads
    }
    if constexpr (A_LDS && VAR != 4 && VAR != 5 && VAR != 7) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int c = tid + i * 256;
            if (c < MB * cpr) {
                const int m = MB == 1 ? 0 : c / cpr, cc = c - m * cpr;
                *reinterpret_cast<u32x4*>(smem + ((int64_t)m * cpr + a_chunk_pos(cc >> 2, cc & 3)) * 16) = areg[i];
            }
        }
        __syncthreads();
    }

    float acc[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[m][c] = 0.f;

    auto compute_tile = [&](const PackedTile16<MB, A_LDS>& tl, int it) {
        const int g = g_begin + it * 64 + lane;
        const int gc = g < g_end ? g : G - 1;
        u32 av[MB][16];                                       // av[m][i] = (a[2i], a[2i+1]) of this lane's group
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 x;
                if constexpr (VAR == 2 || VAR == 4 || VAR == 5 || VAR == 6 || VAR == 7)
                    x = u32x4{SP::kOnes, SP::kOnes, SP::kOnes, SP::kOnes};
                else if constexpr (A_LDS)
                    x = *reinterpret_cast<const u32x4*>(smem + ((int64_t)m * cpr + a_chunk_pos(gc, j)) * 16);
                else
                    x = tl.a[m][j];
                av[m][4 * j + 0] = x[0];
                av[m][4 * j + 1] = x[1];
                av[m][4 * j + 2] = x[2];
                av[m][4 * j + 3] = x[3];
            }
        const u32x2 sv = g < g_end ? tl.s : u32x2{0u, 0u};    // out-of-range lanes contribute 0
        if constexpr (VAR == 1 || VAR == 5 || VAR == 6 || VAR == 7) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[0][c] += u32_as_f32((tl.w[c][0] ^ tl.w[c][1] ^ tl.w[c][2] ^ tl.w[c][3] ^ sv[0] ^ sv[1] ^ av[0][c]) &
                                        0x3fffffffu);
            return;
        }
        if constexpr (STRICT && Act<T>::code == QL_DTYPE_BF16) {
            // bf16 strict: (n - 8) * s evaluated exactly in fp32 (a small integer times a bf16 value) and rounded ONCE to
            // bf16 - torch's `int8 * bf16` product, the reference's per-weight rounding (chatglm_q/int4/triton_ops.py:72-73,
            // chatglm_q/int4/qlinear.py:30-32) - then v_dot2c_f32_bf16 with fp32 accumulation.  2^23 | n is the exact float
            // 8388608 + n: one v_and_or and one subtraction per weight instead of an integer conversion.
            // Round 6: nibble -> float by v_cvt_f32_ubyteN on the even / odd nibbles of a word spread into bytes (2 ands + 1 shift per 8
            // weights instead of a shift + and_or + subtraction per weight), and (n - 8) s as ONE fma n s + (-8 s): n s has <= 12, (n - 8) s
            // <= 11 significant bits and -8 s is a power-of-two multiple of s - every step exact, the value rounded to bf16 is the one the
            // subtract-then-multiply form rounded: bit-equal results at ~3 instead of ~5 VALU instructions per weight.
            const float sc[4] = {SP::lo(sv[0]), SP::hi(sv[0]), SP::lo(sv[1]), SP::hi(sv[1])};
            const float m8[4] = {-8.0f * sc[0], -8.0f * sc[1], -8.0f * sc[2], -8.0f * sc[3]};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32 w = tl.w[c][j];
                    const u32 ev = w & 0x0F0F0F0Fu, od = (w >> 4) & 0x0F0F0F0Fu;   // nibble positions 0, 2, 4, 6 / 1, 3, 5, 7 as bytes
#pragma unroll
                    for (int i = 0; i < 4; ++i) {                        // pair i: k = 8 j + 2 i (position i), + 1 (position i + 4)
                        const u32 src = (i & 1) ? od : ev;               // position i -> byte i / 2 of its word, position i + 4 -> byte i / 2 + 2
                        const float qe = (float)((src >> (8 * (i >> 1))) & 0xFFu);
                        const float qo = (float)((src >> (8 * (i >> 1) + 16)) & 0xFFu);
                        const u32 pr = pack2<__bf16>(__builtin_fmaf(qe, sc[c], m8[c]), __builtin_fmaf(qo, sc[c], m8[c]));
#pragma unroll
                        for (int m = 0; m < MB; ++m) acc[m][c] = SP::dot(pr, av[m][4 * j + i], acc[m][c]);
                    }
                }
        } else if constexpr (STRICT) {
            const h2 k1032 = {(f16)1032.0f, (f16)1032.0f};
            const h2 kInv16 = {(f16)0.0625f, (f16)0.0625f};
            const h2 kM72 = {(f16)-72.0f, (f16)-72.0f};
            u32 k_mask_hi;
            asm volatile("s_mov_b32 %0, 0x00F000F0" : "=s"(k_mask_hi));
            const h2 s2[4] = {as_h2((sv[0] & 0xFFFFu) | (sv[0] << 16)), as_h2((sv[0] >> 16) | (sv[0] & 0xFFFF0000u)),
                              as_h2((sv[1] & 0xFFFFu) | (sv[1] << 16)), as_h2((sv[1] >> 16) | (sv[1] & 0xFFFF0000u))};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32 w = tl.w[c][j], w8 = w >> 8;
                    // exact (n - 8) as fp16 pairs, then ONE rounding per weight in the scale multiply
                    const h2 w0 = (as_h2((w & k_mask) | k_magic) - k1032) * s2[c];
                    const h2 w1 = (as_h2((w & k_mask_hi) | k_magic) * kInv16 + kM72) * s2[c];
                    const h2 w2 = (as_h2((w8 & k_mask) | k_magic) - k1032) * s2[c];
                    const h2 w3 = (as_h2((w8 & k_mask_hi) | k_magic) * kInv16 + kM72) * s2[c];
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        float v = acc[m][c];
                        v = __builtin_amdgcn_fdot2(w0, as_h2(av[m][4 * j + 0]), v, false);
                        v = __builtin_amdgcn_fdot2(w1, as_h2(av[m][4 * j + 1]), v, false);
                        v = __builtin_amdgcn_fdot2(w2, as_h2(av[m][4 * j + 2]), v, false);
                        v = __builtin_amdgcn_fdot2(w3, as_h2(av[m][4 * j + 3]), v, false);
                        acc[m][c] = v;
                    }
                }
        } else {
            float corr[MB];                                   // offset * sum of this group's activations
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                float e = 0.f, o = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    e = SP::dot(SP::kOnes, av[m][i], e);
                    o = SP::dot(SP::kOnes, av[m][i + 1], o);
                }
                corr[m] = SP::offset(e, o);
            }
            const float sc[4] = {SP::lo(sv[0]), SP::hi(sv[0]), SP::lo(sv[1]), SP::hi(sv[1])};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float e[MB], o[MB];                           // two independent dot chains per row
#pragma unroll
                for (int m = 0; m < MB; ++m) e[m] = o[m] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32 w = tl.w[c][j];
                    const u32 wh = w >> 8;
                    const u32 x0 = (w & k_mask) | k_magic;
                    const u32 x1 = ((SP::kSplitChains ? w : (w >> 4)) & k_mask_odd) | k_magic;
                    const u32 x2 = (wh & k_mask) | k_magic;
                    const u32 x3 = ((SP::kSplitChains ? wh : (w >> 12)) & k_mask_odd) | k_magic;
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        e[m] = SP::dot(x0, av[m][4 * j + 0], e[m]);
                        o[m] = SP::dot(x1, av[m][4 * j + 1], o[m]);
                        e[m] = SP::dot(x2, av[m][4 * j + 2], e[m]);
                        o[m] = SP::dot(x3, av[m][4 * j + 3], o[m]);
                    }
                }
#pragma unroll
                for (int m = 0; m < MB; ++m) acc[m][c] = __builtin_fmaf(sc[c], SP::combine(e[m], o[m]) - corr[m], acc[m][c]);
            }
        }
    };

    // ping-pong over two register tiles: a tile's registers are refilled (for two iterations later) right
    // after its math, so one tile is always in flight under the other tile's math and no register copy ever
    // touches a pending load.  The steady-state loop contains only UNCONDITIONAL loads of REAL tiles: a load
    // under `if (more tiles)` leaves hipcc unable to count the loads issued after a given one (it then waits
    // vmcnt(0), i.e. for the tile it just issued), and a dummy load past the end keeps the wave alive until
    // it returns (measured +15 %).  The last round is peeled instead.
    const int full = iters >> 1;
    for (int r = 0; r + 1 < full; ++r) {
        compute_tile(t0, 2 * r);
        t0 = load_tile(2 * r + 2);
        compute_tile(t1, 2 * r + 1);
        t1 = load_tile(2 * r + 3);
    }
    if (full > 0) {
        compute_tile(t0, 2 * full - 2);
        if (iters & 1) t0 = load_tile(2 * full);
        compute_tile(t1, 2 * full - 1);
    }
    if (iters & 1) compute_tile(t0, iters - 1);

#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[m][c] = wave_sum(acc[m][c]);

    if constexpr (KS > 1) {
        // combine the K slices of a quad: slice 0 of each quad adds its partners' sums
        float* red = reinterpret_cast<float*>(smem + (A_LDS ? (((size_t)MB * K * sizeof(T) + 15) & ~(size_t)15) : 0));
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int c = 0; c < 4; ++c) red[(wave * MB + m) * 4 + c] = acc[m][c];
        }
        __syncthreads();
        if (ks == 0 && lane == 0) {
#pragma unroll
            for (int p = 1; p < KS; ++p)
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[m][c] += red[((wave + p) * MB + m) * 4 + c];
        }
    }

    QL_SPAN_END();
    auto epilogue = [&] {
    if (wave_active && ks == 0 && lane == 0) {
        float bq[4] = {0.f, 0.f, 0.f, 0.f}, rq[4] = {0.f, 0.f, 0.f, 0.f};
        if (quad_early) {
            unpack2<T>(bias_q[0], bq[0], bq[1]);
            unpack2<T>(bias_q[1], bq[2], bq[3]);
            unpack2<T>(resid_q[0], rq[0], rq[1]);
            unpack2<T>(resid_q[1], rq[2], rq[3]);
        }
        if (PRO != PRO_NONE && pro.gate_epilogue) {
            // SiLU(h) * gate on the quad's (h0, h1, gate0, gate1) sums; N is a multiple of 4 here
            float y[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                y[c] = Act<T>::round(acc[0][c]);
                if (bias) y[c] = Act<T>::round(y[c] + (quad_early ? bq[c] : Act<T>::load(bias + t * 4 + c)));
            }
            if constexpr (CHAIN == 1) {
                // write-through (agent-scope) 4-byte store of the pair: visible to the consumer blocks without a release fence
                // (a buffer_wbl2 per producer block - 1 712 of them - made the one-launch MLP take 105 us instead of 23)
                const float o0 = Act<T>::round(Act<T>::round(y[0] / (1.0f + __expf(-y[0]))) * y[2]);
                const float o1 = Act<T>::round(Act<T>::round(y[1] / (1.0f + __expf(-y[1]))) * y[3]);
                __hip_atomic_store(reinterpret_cast<u32*>(C + t * 2), pack2<T>(o0, o1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
                Act<T>::store(C + t * 2 + i, Act<T>::round(Act<T>::round(y[i] / (1.0f + __expf(-y[i]))) * y[i + 2]));
            return;
        }
        if (quad_early) {                                     // one row, whole quad: one 8-byte store
            float y[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                y[c] = Act<T>::round(acc[0][c]);
                if (bias) y[c] = resid ? Act<T>::round(y[c] + bq[c]) : y[c] + bq[c];
                if (resid) y[c] = y[c] + rq[c];
            }
            *reinterpret_cast<u32x2*>(C + t * 4) = u32x2{pack2<T>(y[0], y[1]), pack2<T>(y[2], y[3])};
            return;
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m0 + m >= M) break;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int n = t * 4 + c;
                if (n >= N) continue;
                if (resid) {                                  // one row: residual add as a third rounded operation
                    float y = Act<T>::round(acc[m][c]);
                    if (bias) y = Act<T>::round(y + Act<T>::load(bias + n));
                    Act<T>::store(C + n, y + Act<T>::load(resid + n));
                } else {
                    store_out<T>(C + (int64_t)(m0 + m) * ldc + n, acc[m][c], bias ? bias + n : nullptr);
                }
            }
        }
    }
    };
    epilogue();
    if constexpr (CHAIN == 1) {
        // outputs -> visible to the consumer blocks of this launch (cdna_hip_programming.md Guideline 16, R1): the payload went
        // out write-through, every storing wave drains its stores, block barrier, one lane arrives
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int slot = bx & (kChainSlots - 1);
            const unsigned want = (unsigned)(chain.producers + kChainSlots - 1 - slot) / kChainSlots;   // producers of this slot
            if (__hip_atomic_fetch_add(chain.cnt + 16 * slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == want)
                __hip_atomic_fetch_add(chain.cnt + 16 * (kChainSlots + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// QL_GEMV_MIN_WAVES (A/B builds): minimum waves per SIMD the one-row (MB == 1) instantiations are compiled for, i.e. a register cap of 512 / n
#ifndef QL_GEMV_MIN_WAVES
#define QL_GEMV_MIN_WAVES 1
#endif
template <typename T, int MB, int ACH, int KS, bool STRICT, int VAR = 0, int PRO = PRO_NONE>
__global__ __launch_bounds__(256, (MB == 1 ? QL_GEMV_MIN_WAVES : 1)) void w4_packed_gemv_16_kernel(const T* __restrict__ A, const u32x4* __restrict__ Wt,
                                                                const T* __restrict__ Sp, const void* pro_delta,
                                                                const void* pro_ln_weight, int N, int K, int M, int lda32,
                                                                const T* __restrict__ bias, T* __restrict__ C, int64_t ldc,
                                                                void* pro_hout, float pro_eps, int pro_gate,
                                                                const T* __restrict__ resid = nullptr) {
    w4_packed_gemv_16_body<T, MB, ACH, KS, STRICT, VAR, PRO>(A, Wt, Sp, pro_delta, pro_ln_weight, N, K, M, lda32, bias, C, ldc, pro_hout,
                                                              pro_eps, pro_gate, resid, (int)blockIdx.x, (int)blockIdx.y);
}

// ---------------------------------------------------------------------------------------------
// generic dtype kernel (fp32 / bf16): same decomposition, per-nibble dequant, activations from global
// ---------------------------------------------------------------------------------------------
template <typename T>
struct PackedTileG {
    u32x4 w[4];
    float s[4];
};

template <typename T, int MB>
__global__ __launch_bounds__(256) void w4_packed_gemv_kernel(const T* __restrict__ A, const u32x4* __restrict__ Wt,
                                                             const T* __restrict__ Sp, const T* __restrict__ bias,
                                                             T* __restrict__ C, int M, int N, int K, int G,
                                                             int64_t lda, int64_t ldc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 4 + wave;
    if (t * 4 >= N) return;                    // no barriers in this kernel
    const int m0 = blockIdx.y * MB;
    const int iters = (G + 63) >> 6;
    const u32x4* wbase = Wt + (int64_t)t * 4 * G;
    const T* sbase = Sp + (int64_t)t * G * 4;
    const T* arow[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) arow[m] = A + (int64_t)((m0 + m < M) ? (m0 + m) : (M - 1)) * lda;

    auto load_tile = [&](int it) {
        PackedTileG<T> tl;
        const int g = it * 64 + lane;
        const int gc = g < G ? g : G - 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            tl.w[c] = QL_GEMV_W_LOAD(wbase + (int64_t)c * G + gc);
            tl.s[c] = Act<T>::load(sbase + (int64_t)gc * 4 + c);         // masked at use
        }
        return tl;
    };

    float acc[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[m][c] = 0.f;

    auto compute_tile = [&](PackedTileG<T>& t0, int it) {
        const int g = it * 64 + lane;
        const int gc = g < G ? g : G - 1;
        float m8s[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (g >= G) t0.s[c] = 0.f;                                   // out-of-range lanes contribute 0
            m8s[c] = -8.0f * t0.s[c];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                float a[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) a[m] = Act<T>::load(arow[m] + gc * 32 + 8 * j + kk);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float wq = dequant_nibble<T>(t0.w[c][j], packed_pos(kk), t0.s[c], m8s[c]);
#pragma unroll
                    for (int m = 0; m < MB; ++m) acc[m][c] = __builtin_fmaf(a[m], wq, acc[m][c]);
                }
            }
    };
    // two register tiles; steady-state loads unconditional and real, last round peeled
    // (see w4_packed_gemv_16_kernel)
    PackedTileG<T> ta = load_tile(0), tb;
    if (iters > 1) tb = load_tile(1);
    const int full = iters >> 1;
    for (int r = 0; r + 1 < full; ++r) {
        compute_tile(ta, 2 * r);
        ta = load_tile(2 * r + 2);
        compute_tile(tb, 2 * r + 1);
        tb = load_tile(2 * r + 3);
    }
    if (full > 0) {
        compute_tile(ta, 2 * full - 2);
        if (iters & 1) ta = load_tile(2 * full);
        compute_tile(tb, 2 * full - 1);
    }
    if (iters & 1) compute_tile(ta, iters - 1);

#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[m][c] = wave_sum(acc[m][c]);

    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m0 + m >= M) break;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int n = t * 4 + c;
                if (n < N) store_out<T>(C + (int64_t)(m0 + m) * ldc + n, acc[m][c], bias ? bias + n : nullptr);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <typename T>
static int launch_w4_repack_gemv(const uint8_t* Wq, const void* S, void* gemv, int64_t N, int64_t K, hipStream_t st) {
    const int64_t G = K / 32, Npad = (N + 3) & ~(int64_t)3;
    u32x4* Wt = (u32x4*)gemv;
    T* Sp = (T*)((char*)gemv + Npad * G * 16);
    dim3 grid((unsigned)((Npad + 255) / 256), (unsigned)G);
    w4_repack_kernel<T><<<grid, 256, 0, st>>>(Wq, (const T*)S, Wt, Sp, (int)N, (int)Npad, (int)G);
    return finish_launch();
}

template <typename T>
static int launch_w4_unpack_gemv(const void* gemv, uint8_t* Wq, void* S, int64_t N, int64_t K, hipStream_t st) {
    const int64_t G = K / 32, Npad = (N + 3) & ~(int64_t)3;
    const u32x4* Wt = (const u32x4*)gemv;
    const T* Sp = (const T*)((const char*)gemv + Npad * G * 16);
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)G);
    w4_unpack_kernel<T><<<grid, 256, 0, st>>>(Wt, Sp, Wq, (T*)S, (int)N, (int)G);
    return finish_launch();
}

// part 2 (tile-major) from part 1: the two may be one buffer (tiled = gemv + off_wm) or two allocations
template <typename T>
static int launch_w4_tile(const void* gemv, void* tiled, int64_t N, int64_t K, hipStream_t st) {
    const W4Layout L = w4_layout(N, K, sizeof(T));
    const int64_t total = L.ctiles * L.ksteps * 64;
    w4_tile_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>((const u32x4*)gemv, (const T*)((const char*)gemv + L.off_sp),
                                                                      (u32x4*)tiled, (T*)((char*)tiled + (L.off_sm - L.off_wm)),
                                                                      (int)L.Npad, (int)L.G, (int)L.ksteps, total);
    return finish_launch();
}

struct PackedArgs {
    const void* A;
    const void* packed;
    const void* bias;
    void* C;
    int M, N, K;
    int64_t lda, ldc;
    bool strict;
    hipStream_t st;
    const void* resid = nullptr;   // one-row residual epilogue (w4_packed_residual)
};

#ifdef QL_DEV_VARIANTS
static int dev_variant() { return QL_TUNE("QL_VARIANT", 0); }
#endif

// K slices per block: as many as keep every lane of a wave busy (>= 64 groups per slice) while the
// grid is still small (fewer than ~4 blocks per CU); developer build: QLINEAR_W4_KSPLIT overrides for measurements.
// Every block stages its MB activation rows (MB * K values) into LDS.  Once that tile is large enough to limit the
// blocks per CU (> 40 KB: w_out with 2 rows, 55 KB) more, smaller blocks only multiply the staging traffic (1024
// blocks x 55 KB = 56 MB against 28 MB of weights: 20.4 us instead of 13.9), so the split is halved until the staged
// bytes no longer exceed the weight bytes.  Small tiles keep the finer split (qkv_proj, 4 rows: 11.5 vs 14.1 us).
#ifndef QL_KS_BLOCK_BOUND
#define QL_KS_BLOCK_BOUND 1024
#endif
static int choose_ksplit(int64_t quads, int64_t G, int mb = 1) {
    const int forced = QL_TUNE("QLINEAR_W4_KSPLIT", 0);
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    int ks = 1;
    while (ks < 4 && G / (ks * 2) >= 64 && quads * ks / 4 < QL_KS_BLOCK_BOUND) ks *= 2;
    if (mb > 1 && mb * G * 64 > 40 * 1024)
        while (ks > 1 && (quads * ks / 4) * mb * (G * 32) * 2 > quads * 4 * G * 16) ks /= 2;
    return ks;
}

int w4_gemv_ksplit(int64_t quads, int64_t G) { return choose_ksplit(quads, G); }

template <typename T, int MB, int ACH, int KS, bool STRICT>
static int launch_16(const PackedArgs& p) {
    const int64_t G = p.K / 32, Npad = (p.N + 3) & ~(int64_t)3;
    const u32x4* Wt = (const u32x4*)p.packed;
    const T* Sp = (const T*)((const char*)p.packed + Npad * G * 16);
    const int quads = (int)(Npad / 4);
    constexpr int QW = 4 / KS;
    dim3 grid((unsigned)((quads + QW - 1) / QW), (unsigned)((p.M + MB - 1) / MB));
    const size_t lds = (ACH > 0 ? (((size_t)MB * p.K * sizeof(T) + 15) & ~(size_t)15) : 0) +
                       (KS > 1 ? (size_t)4 * MB * 4 * sizeof(float) : 0);
#ifdef QL_DEV_VARIANTS
    if constexpr (MB == 1 && !STRICT) {
        if (dev_variant() == 1) {
            w4_packed_gemv_16_kernel<T, 1, ACH, KS, false, 1><<<grid, 256, lds, p.st>>>((const T*)p.A, Wt, Sp, nullptr, nullptr, p.N, p.K, p.M, (int)p.lda, (const T*)p.bias, (T*)p.C, p.ldc, nullptr, 0.f, 0);
            return finish_launch();
        }
        if (dev_variant() == 2) {
            w4_packed_gemv_16_kernel<T, 1, ACH, KS, false, 2><<<grid, 256, lds, p.st>>>((const T*)p.A, Wt, Sp, nullptr, nullptr, p.N, p.K, p.M, (int)p.lda, (const T*)p.bias, (T*)p.C, p.ldc, nullptr, 0.f, 0);
            return finish_launch();
        }
        if (dev_variant() == 4) {
            w4_packed_gemv_16_kernel<T, 1, ACH, KS, false, 4><<<grid, 256, lds, p.st>>>((const T*)p.A, Wt, Sp, nullptr, nullptr, p.N, p.K, p.M, (int)p.lda, (const T*)p.bias, (T*)p.C, p.ldc, nullptr, 0.f, 0);
            return finish_launch();
        }
        if (dev_variant() == 6) {
            w4_packed_gemv_16_kernel<T, 1, ACH, KS, false, 6><<<grid, 256, lds, p.st>>>((const T*)p.A, Wt, Sp, nullptr, nullptr, p.N, p.K, p.M, (int)p.lda, (const T*)p.bias, (T*)p.C, p.ldc, nullptr, 0.f, 0);
            return finish_launch();
        }
        if (dev_variant() == 7) {
            w4_packed_gemv_16_kernel<T, 1, ACH, KS, false, 7><<<grid, 256, lds, p.st>>>((const T*)p.A, Wt, Sp, nullptr, nullptr, p.N, p.K, p.M, (int)p.lda, (const T*)p.bias, (T*)p.C, p.ldc, nullptr, 0.f, 0);
            return finish_launch();
        }
        if (dev_variant() == 5) {
            w4_packed_gemv_16_kernel<T, 1, ACH, KS, false, 5><<<grid, 256, lds, p.st>>>((const T*)p.A, Wt, Sp, nullptr, nullptr, p.N, p.K, p.M, (int)p.lda, (const T*)p.bias, (T*)p.C, p.ldc, nullptr, 0.f, 0);
            return finish_launch();
        }
    }
#endif
    w4_packed_gemv_16_kernel<T, MB, ACH, KS, STRICT><<<grid, 256, lds, p.st>>>((const T*)p.A, Wt, Sp, nullptr, nullptr, p.N, p.K, p.M,
                                                                               (int)p.lda, (const T*)p.bias, (T*)p.C, p.ldc, nullptr, 0.f, 0,
                                                                               (const T*)p.resid);
    return finish_launch(QL_K_W4_GEMV);
}

template <typename T, int MB, int KS, bool STRICT>
static int launch_16_ach(const PackedArgs& p) {
    const int64_t chunks = (int64_t)MB * (p.K / 8);
    const bool lds_ok = (size_t)MB * p.K * sizeof(T) <= 60 * 1024;
#ifdef QL_DEV_VARIANTS
    if (dev_variant() == 3) return launch_16<T, MB, 0, KS, STRICT>(p);     // activations straight from global
#endif
    if (lds_ok && chunks <= 2 * 256) return launch_16<T, MB, 2, KS, STRICT>(p);
    if (lds_ok && chunks <= 4 * 256) return launch_16<T, MB, 4, KS, STRICT>(p);
    if (lds_ok && chunks <= 8 * 256) return launch_16<T, MB, 8, KS, STRICT>(p);
    return launch_16<T, MB, 0, KS, STRICT>(p);
}

template <typename T, int MB, bool STRICT>
static int launch_16_mb(const PackedArgs& p) {
    const int64_t Npad = (p.N + 3) & ~(int64_t)3;
    switch (choose_ksplit(Npad / 4, p.K / 32, MB)) {
    case 4: return launch_16_ach<T, MB, 4, STRICT>(p);
    case 2: return launch_16_ach<T, MB, 2, STRICT>(p);
    default: return launch_16_ach<T, MB, 1, STRICT>(p);
    }
}

template <typename T, bool STRICT>
static int launch_16_any(const PackedArgs& p) {
    if (p.M == 1) return launch_16_mb<T, 1, STRICT>(p);
    if (p.M == 2) return launch_16_mb<T, 2, STRICT>(p);
    return launch_16_mb<T, 4, STRICT>(p);
}

template <typename T, int MB>
static int launch_generic_mb(const PackedArgs& p) {
    const int64_t G = p.K / 32, Npad = (p.N + 3) & ~(int64_t)3;
    const u32x4* Wt = (const u32x4*)p.packed;
    const T* Sp = (const T*)((const char*)p.packed + Npad * G * 16);
    const int quads = (int)(Npad / 4);
    dim3 grid((unsigned)((quads + 3) / 4), (unsigned)((p.M + MB - 1) / MB));
    w4_packed_gemv_kernel<T, MB><<<grid, 256, 0, p.st>>>((const T*)p.A, Wt, Sp, (const T*)p.bias, (T*)p.C, p.M, p.N,
                                                         p.K, (int)G, p.lda, p.ldc);
    return finish_launch(QL_K_W4_GEMV);
}

// one-row forward with an activation prologue (decode step)
template <typename T, int ACH, int KS, int PRO, bool STRICT>
static int launch_16_pro(const PackedArgs& p, const Prologue& pro) {
    const int64_t G = p.K / 32, Npad = (p.N + 3) & ~(int64_t)3;
    const u32x4* Wt = (const u32x4*)p.packed;
    const T* Sp = (const T*)((const char*)p.packed + Npad * G * 16);
    const int quads = (int)(Npad / 4);
    constexpr int QW = 4 / KS;
    dim3 grid((unsigned)((quads + QW - 1) / QW), 1);
    const size_t lds = (((size_t)p.K * sizeof(T) + 15) & ~(size_t)15) + 4 * 4 * sizeof(float) + 4 * sizeof(float);
    w4_packed_gemv_16_kernel<T, 1, ACH, KS, STRICT, 0, PRO><<<grid, 256, lds, p.st>>>(
        (const T*)p.A, Wt, Sp, pro.delta, pro.ln_weight, p.N, p.K, 1, (int)p.lda, (const T*)p.bias, (T*)p.C, p.ldc, pro.hout, pro.eps,
        pro.gate_epilogue);
    return finish_launch(QL_K_W4_GEMV);
}

template <typename T, int PRO, bool STRICT>
static int launch_16_pro_any(const PackedArgs& p, const Prologue& pro) {
    const int64_t chunks = p.K / 8, Npad = (p.N + 3) & ~(int64_t)3;
    if ((size_t)p.K * sizeof(T) > 60 * 1024 || chunks > 8 * 256) return QL_ERR_UNSUPPORTED;
    const int ks = choose_ksplit(Npad / 4, p.K / 32);
#define QL_PRO(ACH_)                                                        \
    switch (ks) {                                                           \
    case 4: return launch_16_pro<T, ACH_, 4, PRO, STRICT>(p, pro);          \
    case 2: return launch_16_pro<T, ACH_, 2, PRO, STRICT>(p, pro);          \
    default: return launch_16_pro<T, ACH_, 1, PRO, STRICT>(p, pro);         \
    }
    if (chunks <= 2 * 256) { QL_PRO(2) }
    if (chunks <= 4 * 256) { QL_PRO(4) }
    QL_PRO(8)
#undef QL_PRO
}

#ifdef QL_DEV_EXPERIMENTS      // libqlinear_hip_dev.so only (include/qlinear_hip_dev.h): measured 27.5 vs 23.0 us, profiles/r02_mlp_pair.txt
// ---------------------------------------------------------------------------------------------
// The MLP of a decode step in ONE launch (experiment): blocks [0, nA) run the first projection (RMSNorm
// prologue, SiLU * gate epilogue on the gate-interleaved copy) and publish its (1, hidden) row; the remaining blocks run
// the second projection (residual epilogue): their weight tiles are requested at once, then they wait for the row.
// Workgroups are dispatched in index order, so every producer is resident or done before the first consumer starts:
// the waits cannot deadlock (they are bounded all the same).
// ---------------------------------------------------------------------------------------------
template <typename T>
struct PairArgs {
    const T* x;            // (1, Ka) hidden state
    const T* ln_weight;
    float eps;
    const u32x4* Wa; const T* Sa; const T* bias_a; int Na, Ka;     // gate-interleaved first projection (Na packed columns)
    T* mid;                // (1, Na / 2)
    const u32x4* Wb; const T* Sb; const T* bias_b; int Nb, Kb;     // second projection, Kb == Na / 2
    const T* resid; T* out;
    int nA;
    ChainArgs chain;
};

template <typename T, int ACH_A, int KS_A, int ACH_B, int KS_B>
__global__ __launch_bounds__(256) void w4_mlp_pair_kernel(const PairArgs<T> p) {
    if ((int)blockIdx.x < p.nA)
        w4_packed_gemv_16_body<T, 1, ACH_A, KS_A, false, 0, PRO_NORM, 1>(p.x, p.Wa, p.Sa, nullptr, p.ln_weight, p.Na, p.Ka, 1, p.Ka, p.bias_a,
                                                                      p.mid, p.Na, nullptr, p.eps, 1, nullptr, (int)blockIdx.x, 0, p.chain);
    else
        w4_packed_gemv_16_body<T, 1, ACH_B, KS_B, false, 0, PRO_NONE, 2>(p.mid, p.Wb, p.Sb, nullptr, nullptr, p.Nb, p.Kb, 1, p.Kb, p.bias_b,
                                                                      p.out, p.Nb, nullptr, 0.f, 0, p.resid, (int)blockIdx.x - p.nA, 0,
                                                                      p.chain);
}

size_t w4_mlp_pair_workspace_bytes() { return sizeof(unsigned) * kChainWords; }

template <typename T>
static int launch_mlp_pair(const void* x, const void* ln_weight, float eps, const void* packed_a, const void* bias_a, int64_t Na,
                           int64_t Ka, const void* packed_b, const void* bias_b, int64_t Nb, int64_t Kb, const void* resid,
                           void* mid, void* out, void* ws, hipStream_t st) {
    const int64_t Ga = Ka / 32, Gb = Kb / 32, NpA = (Na + 3) & ~(int64_t)3, NpB = (Nb + 3) & ~(int64_t)3;
    const int ks_a = choose_ksplit(NpA / 4, Ga), ks_b = choose_ksplit(NpB / 4, Gb);
    // the one combination a ChatGLM2-6B layer needs (4096 -> 2 x 13696 -> 4096); everything else: two launches
    if (!(Ka / 8 <= 2 * 256 && ks_a == 1 && Kb / 8 > 4 * 256 && Kb / 8 <= 8 * 256 && ks_b == 4 && (size_t)Kb * sizeof(T) <= 60 * 1024))
        return QL_ERR_UNSUPPORTED;
    PairArgs<T> p;
    p.x = (const T*)x; p.ln_weight = (const T*)ln_weight; p.eps = eps;
    p.Wa = (const u32x4*)packed_a; p.Sa = (const T*)((const char*)packed_a + NpA * Ga * 16); p.bias_a = (const T*)bias_a;
    p.Na = (int)Na; p.Ka = (int)Ka; p.mid = (T*)mid;
    p.Wb = (const u32x4*)packed_b; p.Sb = (const T*)((const char*)packed_b + NpB * Gb * 16); p.bias_b = (const T*)bias_b;
    p.Nb = (int)Nb; p.Kb = (int)Kb; p.resid = (const T*)resid; p.out = (T*)out;
    const int nA = (int)((NpA / 4 + 3) / 4), nB = (int)(NpB / 4);           // KS_A = 1: 4 quads per block; KS_B = 4: one quad per block
    p.nA = nA;
    p.chain = ChainArgs{(unsigned*)ws, nA, nB};
    const size_t lds_a = (((size_t)Ka * sizeof(T) + 15) & ~(size_t)15) + 4 * 4 * sizeof(float) + 4 * sizeof(float);
    const size_t lds_b = (((size_t)Kb * sizeof(T) + 15) & ~(size_t)15) + (size_t)4 * 4 * sizeof(float);
    w4_mlp_pair_kernel<T, 2, 1, 8, 4><<<(unsigned)(nA + nB), 256, lds_a > lds_b ? lds_a : lds_b, st>>>(p);
    return finish_launch();
}

int w4_mlp_pair(int dtype, const void* x, const void* ln_weight, float eps, const void* packed_a, const void* bias_a, int64_t Na,
                int64_t Ka, const void* packed_b, const void* bias_b, int64_t Nb, int64_t Kb, const void* resid, void* mid,
                void* out, void* ws, hipStream_t st) {
    if (dtype == QL_DTYPE_F16) return launch_mlp_pair<f16>(x, ln_weight, eps, packed_a, bias_a, Na, Ka, packed_b, bias_b, Nb, Kb, resid, mid, out, ws, st);
    if (dtype == QL_DTYPE_BF16) return launch_mlp_pair<__bf16>(x, ln_weight, eps, packed_a, bias_a, Na, Ka, packed_b, bias_b, Nb, Kb, resid, mid, out, ws, st);
    return QL_ERR_BAD_DTYPE;
}

#endif  // QL_DEV_EXPERIMENTS

template <typename T, bool STRICT>
static int w4_packed_fused_t(int kind, const PackedArgs& p, const Prologue& pro) {
    if (kind == PRO_SILU) return launch_16_pro_any<T, PRO_SILU, STRICT>(p, pro);
    if (kind == PRO_ADDNORM)
        return !pro.delta && !pro.hout ? launch_16_pro_any<T, PRO_NORM, STRICT>(p, pro) : launch_16_pro_any<T, PRO_ADDNORM, STRICT>(p, pro);
    return QL_ERR_UNSUPPORTED;
}

int w4_packed_fused(int dtype, int kind, bool gate_epilogue, bool strict, const void* A, const void* packed, const void* bias, void* C,
                    int64_t N, int64_t K, const void* delta, const void* ln_weight, void* hout, float eps, hipStream_t st) {
    const PackedArgs p{A, packed, bias, C, 1, (int)N, (int)K, K, N, strict, st};
    const Prologue pro{delta, ln_weight, hout, eps, gate_epilogue ? 1 : 0};
    if (dtype == QL_DTYPE_F16) return strict ? w4_packed_fused_t<f16, true>(kind, p, pro) : w4_packed_fused_t<f16, false>(kind, p, pro);
    if (dtype == QL_DTYPE_BF16) return strict ? w4_packed_fused_t<__bf16, true>(kind, p, pro) : w4_packed_fused_t<__bf16, false>(kind, p, pro);
    return QL_ERR_BAD_DTYPE;
}

int w4_unpack_gemv(int dtype, const void* gemv, uint8_t* Wq, void* S, int64_t N, int64_t K, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F32: return launch_w4_unpack_gemv<float>(gemv, Wq, S, N, K, st);
    case QL_DTYPE_F16: return launch_w4_unpack_gemv<f16>(gemv, Wq, S, N, K, st);
    case QL_DTYPE_BF16: return launch_w4_unpack_gemv<__bf16>(gemv, Wq, S, N, K, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

int w4_repack_gemv(int dtype, const uint8_t* Wq, const void* S, void* gemv, int64_t N, int64_t K, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F32: return launch_w4_repack_gemv<float>(Wq, S, gemv, N, K, st);
    case QL_DTYPE_F16: return launch_w4_repack_gemv<f16>(Wq, S, gemv, N, K, st);
    case QL_DTYPE_BF16: return launch_w4_repack_gemv<__bf16>(Wq, S, gemv, N, K, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

int w4_tile(int dtype, const void* gemv, void* tiled, int64_t N, int64_t K, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F32: return launch_w4_tile<float>(gemv, tiled, N, K, st);
    case QL_DTYPE_F16: return launch_w4_tile<f16>(gemv, tiled, N, K, st);
    case QL_DTYPE_BF16: return launch_w4_tile<__bf16>(gemv, tiled, N, K, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

int w4_repack(int dtype, const uint8_t* Wq, const void* S, void* packed, int64_t N, int64_t K, hipStream_t st) {
    const int rc = w4_repack_gemv(dtype, Wq, S, packed, N, K, st);
    if (rc != 0) return rc;
    return w4_tile(dtype, packed, (char*)packed + w4_layout(N, K, dtype == QL_DTYPE_F32 ? 4 : 2).off_wm, N, K, st);
}

// Which kernel serves M rows.  The VALU GEMV does 4 rows per pass at about 4x the issue cost of one row; the MFMA
// kernels (w4_fewrow.hip up to 32 rows, w4_gemm.hip above) cost the same for any row count of a tile.  Measured
// (ChatGLM2-6B shapes, fp16, us):
//   rows        4096->4608  4096->4096  4096->27392  13696->4096
//   2  GEMV         6.8         6.1        19.2         14.2
//   3-4 GEMV       11.4         8.3        48           29
//   3-16 few-row    9.2-10.4    8.7-9.3    20-22.4      14.3-15.3
// so the MFMA path starts at 3 rows.  QLINEAR_GEMV_MAX_ROWS forces the limit.
void w4_gemv_blocks(int64_t N, int64_t K, int64_t* w_block_bytes, int64_t* s_block_bytes, int64_t* s_offset, int64_t* blocks) {
    const int64_t G = K / 32, Npad = (N + 3) & ~(int64_t)3, quads = Npad / 4;
    const int qw = 4 / choose_ksplit(quads, G);               // column quads per workgroup of the one-row kernel
    *w_block_bytes = (int64_t)qw * 4 * G * 16;
    *s_block_bytes = (int64_t)qw * G * 8;                     // 16-bit scales: 4 per (quad, group)
    *s_offset = Npad * G * 16;
    *blocks = (quads + qw - 1) / qw;
}

// one-row forward whose output is added to the residual stream: C = round(y + resid), y = rounded sum (+ bias)
int w4_packed_residual(int dtype, bool strict, const void* A, const void* packed, const void* bias, const void* resid, void* C, int64_t N,
                       int64_t K, hipStream_t st) {
    PackedArgs p{A, packed, bias, C, 1, (int)N, (int)K, K, N, strict, st};
    p.resid = resid;
    if (dtype == QL_DTYPE_F16) return strict ? launch_16_any<f16, true>(p) : launch_16_any<f16, false>(p);
    if (dtype == QL_DTYPE_BF16) return strict ? launch_16_any<__bf16, true>(p) : launch_16_any<__bf16, false>(p);
    return QL_ERR_UNSUPPORTED;
}

// 2..4 rows in the default arithmetic: the 4x4x4-MFMA kernel on part 1 (w4_rows4.hip).  Measured against what served those
// row counts before (us, fp16, ChatGLM2-6B shapes qkv / o / w_in / w_out; tools/gemv_rows.py):
//   rows   GEMV (2) / few-row MFMA (3, 4)     rows4
//   2      6.6 / 5.9 / 19.1 / 13.1            5.9 / 5.2 / 15.9 / 12.0
//   3      8.8 / 8.4 / 19.6 / 13.8            6.7 / 5.4 / 16.6 / (15.4)
//   4      9.1 / 8.4 / 19.3 / 13.6            6.9 / 5.5 / 17.1 / (16.5)
// (one row: 4.7 / 4.4 / 13.0 / 9.1 on the GEMV, 6.0 / 5.3 / 15.9 / 13.9 here) - so: 2 rows always, 3 and 4 rows while the
// staged rows stay small (K <= 8192: w_out's 13696-deep rows are the case that loses).  QLINEAR_ROWS4_MIN / _MAX move the
// row range (MAX = 0 turns the kernel off).
bool w4_rows4_serves(int dtype, int64_t M, int64_t N, int64_t K, int64_t lda, bool strict) {
    // 2..4 rows: measured against the GEMV (2 rows) and the few-row kernel (3, 4 rows), profiles/r02_rows_2_to_4.txt
    const int r4_min = QL_TUNE("QLINEAR_ROWS4_MIN", 2), r4_max = QL_TUNE("QLINEAR_ROWS4_MAX", 4);
    return !strict && !(dispatch_flags() & QL_D_NOROWS4) && M >= r4_min && M <= r4_max && (M <= 2 || M * K * 2 <= 64 * 1024) && w4_rows4_supported(dtype, M, N, K, lda);
}

int w4_rows4_gated(int dtype, const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                   int64_t lda, int64_t ldc, hipStream_t st) {
    const int64_t Npad = (N + 3) & ~(int64_t)3;
    return w4_rows4(dtype, choose_ksplit(Npad / 4, K / 32, (int)M), A, packed, bias, C, M, N, K, lda, ldc, st, true);
}

int w4_rows4_fused(int dtype, bool gate, const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                   const void* delta, const void* ln_weight, void* hout, float eps, hipStream_t st) {
    const int64_t Npad = (N + 3) & ~(int64_t)3;
    return w4_rows4(dtype, choose_ksplit(Npad / 4, K / 32, (int)M), A, packed, bias, C, M, N, K, K, gate ? N / 2 : N, st, gate, delta,
                    ln_weight, hout, eps);
}

bool w4_rows_use_gemm(int64_t M, int64_t N, int64_t K) {
    const int forced = QL_TUNE("QLINEAR_GEMV_MAX_ROWS", -1);
    if (forced >= 0) return M > forced;
    (void)N;
    (void)K;
    return M > 2;
}

size_t w4_packed_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (!w4_rows_use_gemm(M, N, K)) return 0;
    if (w4_rows16_serves(QL_DTYPE_F16, M, N, K, K)) return 0;   // one launch, no slabs (unaligned rows fall to the few-row kernel, which then runs unsplit)
    return w4_fewrow_supported(M, N, K) ? w4_fewrow_workspace_bytes(M, N, K) : w4_packed_gemm_workspace_bytes(M, N, K);
}

// any row count on the tile-major part alone (fp16 / bf16): the MFMA kernels
int w4_tiled(int dtype, const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
             int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st) {
    if (dtype != QL_DTYPE_F16 && dtype != QL_DTYPE_BF16) return QL_ERR_BAD_DTYPE;
    if (w4_fewrow_supported(M, N, K)) return w4_fewrow(dtype, A, tiled, bias, C, M, N, K, lda, ldc, ws, ws_bytes, st);
    return w4_packed_gemm(dtype, A, tiled, bias, C, M, N, K, lda, ldc, ws, ws_bytes, st);
}

int w4_packed(int dtype, const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N,
              int64_t K, int64_t lda, int64_t ldc, bool strict, void* ws, size_t ws_bytes, hipStream_t st) {
    if (w4_rows4_serves(dtype, M, N, K, lda, strict)) {
        const int64_t Npad = (N + 3) & ~(int64_t)3;
        return w4_rows4(dtype, choose_ksplit(Npad / 4, K / 32, (int)M), A, packed, bias, C, M, N, K, lda, ldc, st);
    }
    // 3..16 rows: 16x16x32 MFMA on part 1, one launch (w4_rows16.hip; always the reference's rounding sequence)
    if (w4_rows_use_gemm(M, N, K) && w4_rows16_serves(dtype, M, N, K, K)) {
        // (the caller was told part 1 serves this call - qlinear_w4g32_rows_on_tiled - and may hold nothing else: no silent detour to part 2)
        if (lda % 8 != 0 || lda > 0x7fffffff || ((uintptr_t)A & 15) != 0) return QL_ERR_MISALIGNED;
        return w4_rows16(dtype, A, packed, bias, C, M, N, K, lda, ldc, st);
    }
    // many rows: the MFMA GEMM (always the reference's rounding sequence); it needs 16-byte aligned rows
    if (w4_rows_use_gemm(M, N, K) && (dtype == QL_DTYPE_F16 || dtype == QL_DTYPE_BF16))
        return w4_tiled(dtype, A, (const char*)packed + w4_layout(N, K, 2).off_wm, bias, C, M, N, K, lda, ldc, ws, ws_bytes, st);
    if (lda > 0x7fffffff) return QL_ERR_UNSUPPORTED;          // the GEMV kernels take the row stride as 32 bits
    const PackedArgs p{A, packed, bias, C, (int)M, (int)N, (int)K, lda, ldc, strict, st};
    switch (dtype) {
    case QL_DTYPE_F16:
        return strict ? launch_16_any<f16, true>(p) : launch_16_any<f16, false>(p);
    case QL_DTYPE_BF16:
        return strict ? launch_16_any<__bf16, true>(p) : launch_16_any<__bf16, false>(p);
    case QL_DTYPE_F32:      // fp32 products are already exact-dequant: one kernel serves both modes
        if (M == 1) return launch_generic_mb<float, 1>(p);
        if (M == 2) return launch_generic_mb<float, 2>(p);
        return launch_generic_mb<float, 4>(p);
    default: return QL_ERR_BAD_DTYPE;
    }
}

}  // namespace ql

#ifdef QL_SPAN_PROBE
// developer entry points of the span-probe build (bench.py's kernel_span leg)
// in-stream reset (one memset node: start = end = 0 in every slot), so that the probed launch runs behind other work
extern "C" int qlinear_span_reset_async(void* stream) {
    void* p = nullptr;
    const int rc = (int)hipGetSymbolAddress(&p, HIP_SYMBOL(ql::ql_span_slots));
    if (rc) return rc;
    return (int)hipMemsetD32Async((hipDeviceptr_t)p, 0u, 4 * ql::kSpanWaves, (hipStream_t)stream);
}
// rate of the s_memrealtime counter in kHz (hipDeviceAttributeWallClockRate)
extern "C" int qlinear_span_clock_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
    return khz;
}
// host-synchronous: min start / max end over the slots written since the last reset; waves = number of slots seen
extern "C" int qlinear_span_read(unsigned long long* first_start, unsigned long long* last_end, int* waves) {
    static unsigned long long host[2 * ql::kSpanWaves];
    const int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ql::ql_span_slots), sizeof(host));
    unsigned long long lo = ~0ull, hi = 0ull;
    int n = 0;
    for (int i = 0; i < ql::kSpanWaves; ++i) {
        if (host[2 * i] == 0 || host[2 * i + 1] == 0) continue;
        lo = host[2 * i] < lo ? host[2 * i] : lo;
        hi = host[2 * i + 1] > hi ? host[2 * i + 1] : hi;
        ++n;
    }
    *first_start = lo;
    *last_end = hi;
    *waves = n;
    return rc;
}
#endif
