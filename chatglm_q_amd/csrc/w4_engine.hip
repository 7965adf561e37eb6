// The MLP of a one-row decode step as ONE persistent launch (gfx950 / MI355X) - VERDICT r2 item 1.
//
//   Out = round(w_out(round(silu(h) * gate)) + X),   (h | gate) = w_in(rmsnorm(X) * ln_weight)      chatglm_q/model.py:199-201,244-245
//
// replaces qlinear_w4g32_fwd_packed_fused(QL_PRO_ADDNORM | QL_EPI_SILU_GATE) + qlinear_w4g32_fwd_packed_residual, bit for bit.
//
// Structure (MI355X_MICROARCH.md, rows engine-vs-launches / ldsdma-fill / prefetch-credit; cdna_hip_programming.md 5.6):
//   * one workgroup per CU (gridDim = number of CUs), 8 waves: wave 0 is the LOADER, waves 1..7 are CONSUMERS;
//   * the loader streams this CU's share of BOTH projections' packed weights (part 1 of the derived layout, as it lies)
//     into an LDS ring of 13 x 9 KB slots with `global_load_lds_dwordx4 ... nt` (LDS-DMA: no VGPR round trip) and never
//     waits for an activation: while the CU waits for the (1, hidden) row between the projections, the ring fills with
//     the second projection's weights (the prefetch credit);
//   * a TASK = one (column quad, K slice) of the two-launch GEMV = what ONE wave of w4_packed_gemv_16_kernel computes:
//     the same lane -> group mapping, the same tile order, the same v_dot2c sequence, the same DPP reduction, so every
//     partial sum has the bits it has there; K slices of a quad are combined through LDS in slice order;
//   * the row between the projections goes through 8-byte {data, tag} granules written with ONE agent-scope (sc1) store
//     each and swept by the consumer waves (relaxed agent-scope loads, s_sleep between sweeps) into the LDS image of the
//     second projection's activation row - no flag, no fence (Guideline 16, R2).  tag = launch epoch, a word in the
//     workspace that the last workgroup to finish increments: no memset node between launches, graph-replay safe.
// Every spin is bounded; a wave that gives up sets the workspace's error word (results of that launch are garbage).
#include <stdlib.h>

#include "launch.h"
#include "w4_splice.h"

namespace ql {

constexpr int kEngWaves = 8;                     // 1 loader + 7 consumers
constexpr int kEngConsumers = kEngWaves - 1;
constexpr int kEngThreads = 64 * kEngWaves;
constexpr int kEngSlotBytes = 9216;              // 4 columns x 128 groups x 16 B + 1 KB of scales
constexpr int kEngSlots = 13;
constexpr int kEngDepth = 7;                     // tasks in flight per loader: 9 loads each, 63 = the vmcnt ceiling
constexpr int kEngLoadsPerTask = 9;
constexpr int kEngMaxGroupsPerTask = 128;
constexpr int kEngMaxQuadsB = 16;                // second-projection column quads per workgroup (K-slice combine slots)
constexpr unsigned kEngSpinLds = 1u << 22;       // bounds of the spins (each iteration sleeps): ~tens of ms
constexpr unsigned kEngSpinGlobal = 1u << 18;

// workspace words (u32): [0] epoch, [1] finished-workgroup count, [2] error code, [16 ...) granules (u64, 64-byte aligned)
constexpr int kEngWsHeaderBytes = 64;

#ifdef QL_ENGINE_TRACE
constexpr int kEngTraceWords = 16;               // per workgroup: s_memrealtime stamps (developer build)
#define ENG_STAMP(i) do { if (lane == 0 && p.trace) p.trace[(int)blockIdx.x * kEngTraceWords + (i)] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ENG_STAMP(i) do { } while (0)
#endif

template <typename T>
struct EngineArgs {
    const T* x;                 // (1, Ka) hidden state: RMSNorm input AND the residual the output is added to
    const T* ln_weight;
    float eps;
    const u32x4* Wa; const T* Sa; const T* bias_a; int Na, Ka, KSa;   // gate-interleaved first projection, KSa K slices per quad
    const u32x4* Wb; const T* Sb; const T* bias_b; int Nb, Kb, KSb;   // second projection, Kb == Na / 2
    T* out;
    unsigned* ws;               // header + granules
    unsigned long long* trace;  // developer builds only
};

struct EngTask {
    int phase;                  // 0: first projection, 1: second
    int quad, ks;               // column quad, K slice
    int g0, ng;                 // first group of the slice, groups in it (<= 128)
    int qlocal;                 // index of the quad among this workgroup's quads of the phase
};

struct EngPlan {                // this workgroup's share: contiguous quad ranges of both projections
    int qa0, qa1, qb0, qb1;     // [qa0, qa1) quads of the first projection, [qb0, qb1) of the second
    int nA, nB;                 // tasks per phase
    int Ga, Gb, gsa, gsb;       // groups per row / per K slice
};

template <typename T>
__device__ __forceinline__ EngPlan eng_plan(const EngineArgs<T>& p) {
    EngPlan pl;
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    const int qa = (p.Na + 3) >> 2, qb = (p.Nb + 3) >> 2;
    pl.qa0 = (int)((long long)qa * b / nb);
    pl.qa1 = (int)((long long)qa * (b + 1) / nb);
    pl.qb0 = (int)((long long)qb * b / nb);
    pl.qb1 = (int)((long long)qb * (b + 1) / nb);
    pl.nA = (pl.qa1 - pl.qa0) * p.KSa;
    pl.nB = (pl.qb1 - pl.qb0) * p.KSb;
    pl.Ga = p.Ka >> 5;
    pl.Gb = p.Kb >> 5;
    pl.gsa = (pl.Ga + p.KSa - 1) / p.KSa;
    pl.gsb = (pl.Gb + p.KSb - 1) / p.KSb;
    return pl;
}

// task i of the workgroup's stream: phase A quad-major (quad, slice), then phase B slice-minor as well
template <typename T>
__device__ __forceinline__ EngTask eng_task(const EngineArgs<T>& p, const EngPlan& pl, int i) {
    EngTask t;
    if (i < pl.nA) {
        t.phase = 0;
        t.qlocal = i / p.KSa;
        t.ks = i - t.qlocal * p.KSa;
        t.quad = pl.qa0 + t.qlocal;
        t.g0 = t.ks * pl.gsa;
        t.ng = min(pl.Ga, t.g0 + pl.gsa) - t.g0;
    } else {
        const int j = i - pl.nA;
        t.phase = 1;
        t.qlocal = j / p.KSb;
        t.ks = j - t.qlocal * p.KSb;
        t.quad = pl.qb0 + t.qlocal;
        t.g0 = t.ks * pl.gsb;
        t.ng = min(pl.Gb, t.g0 + pl.gsb) - t.g0;
    }
    if (t.ng < 0) t.ng = 0;
    return t;
}

// ---- LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to 1 KB of LDS at `lds_dst` (wave-uniform) -----------------
// M0 holds the LDS base of the transfer and is compiler-reserved: written and restored inside the statement
// (cdna_hip_programming.md 5.7).  The load is invisible to hipcc's vmcnt bookkeeping: the loader counts its own.
__device__ __forceinline__ void glds16_nt(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
// vmcnt(9 * d) for a run-time d in [0, kEngDepth): the immediate must be a constant
__device__ __forceinline__ void wait_tasks_in_flight(int d) {
    switch (d) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<9>(); break;
    case 2: wait_vmcnt<18>(); break;
    case 3: wait_vmcnt<27>(); break;
    case 4: wait_vmcnt<36>(); break;
    case 5: wait_vmcnt<45>(); break;
    default: wait_vmcnt<54>(); break;
    }
}

__device__ __forceinline__ void lds_fence() {       // the wave's LDS operations so far are done; no compiler motion across
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// spin until *flag == want (LDS word written by another wave of the workgroup); false after kEngSpinLds polls
__device__ __forceinline__ bool lds_wait_eq(volatile unsigned* flag, unsigned want) {
    bool hit = false;
    for (unsigned spins = 0; spins < kEngSpinLds; ++spins) {
        if (*flag == want) {
            hit = true;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");                  // nothing that follows is read before the flag was seen
    return hit;
}
__device__ __forceinline__ bool lds_wait_ge(volatile unsigned* flag, unsigned want) {
    for (unsigned spins = 0; spins < kEngSpinLds; ++spins) {
        if (*flag >= want) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// ---- the tile math of w4_packed_gemv_16_body (MB = 1), operands from LDS ------------------------------------------------------
// One tile = this lane's group (32 k) of the wave's 4 columns.  Same instruction sequence as compute_tile there: the sums
// carry the same bits.
template <typename T, bool STRICT>
__device__ __forceinline__ void eng_tile_math(const u32x4 (&w)[4], const u32x2 sv, const u32 (&av)[16], float (&acc)[4], u32 k_mask,
                                              u32 k_mask_odd, u32 k_magic) {
    typedef Splice<T> SP;
    if constexpr (STRICT && Act<T>::code == QL_DTYPE_BF16) {
        const float sc[4] = {SP::lo(sv[0]), SP::hi(sv[0]), SP::lo(sv[1]), SP::hi(sv[1])};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32 ww = w[c][j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float qe = u32_as_f32(((ww >> (4 * i)) & 0xFu) | 0x4B000000u) - 8388616.0f;
                    const float qo = u32_as_f32(((ww >> (16 + 4 * i)) & 0xFu) | 0x4B000000u) - 8388616.0f;
                    const u32 pr = pack2<__bf16>(qe * sc[c], qo * sc[c]);
                    acc[c] = SP::dot(pr, av[4 * j + i], acc[c]);
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
                const u32 ww = w[c][j], w8 = ww >> 8;
                const h2 w0 = (as_h2((ww & k_mask) | k_magic) - k1032) * s2[c];
                const h2 w1 = (as_h2((ww & k_mask_hi) | k_magic) * kInv16 + kM72) * s2[c];
                const h2 w2 = (as_h2((w8 & k_mask) | k_magic) - k1032) * s2[c];
                const h2 w3 = (as_h2((w8 & k_mask_hi) | k_magic) * kInv16 + kM72) * s2[c];
                float v = acc[c];
                v = __builtin_amdgcn_fdot2(w0, as_h2(av[4 * j + 0]), v, false);
                v = __builtin_amdgcn_fdot2(w1, as_h2(av[4 * j + 1]), v, false);
                v = __builtin_amdgcn_fdot2(w2, as_h2(av[4 * j + 2]), v, false);
                v = __builtin_amdgcn_fdot2(w3, as_h2(av[4 * j + 3]), v, false);
                acc[c] = v;
            }
    } else {
        float e0 = 0.f, o0 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            e0 = SP::dot(SP::kOnes, av[i], e0);
            o0 = SP::dot(SP::kOnes, av[i + 1], o0);
        }
        const float corr = SP::offset(e0, o0);
        const float sc[4] = {SP::lo(sv[0]), SP::hi(sv[0]), SP::lo(sv[1]), SP::hi(sv[1])};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float e = 0.f, o = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32 ww = w[c][j];
                const u32 wh = ww >> 8;
                const u32 x0 = (ww & k_mask) | k_magic;
                const u32 x1 = ((SP::kSplitChains ? ww : (ww >> 4)) & k_mask_odd) | k_magic;
                const u32 x2 = (wh & k_mask) | k_magic;
                const u32 x3 = ((SP::kSplitChains ? wh : (ww >> 12)) & k_mask_odd) | k_magic;
                e = SP::dot(x0, av[4 * j + 0], e);
                o = SP::dot(x1, av[4 * j + 1], o);
                e = SP::dot(x2, av[4 * j + 2], e);
                o = SP::dot(x3, av[4 * j + 3], o);
            }
            acc[c] = __builtin_fmaf(sc[c], SP::combine(e, o) - corr, acc[c]);
        }
    }
}

// LDS carve-up (dynamic region, every offset a multiple of 16):
//   [0, ring)                    kEngSlots x kEngSlotBytes
//   xrow   Ka * 2 bytes          normalised input row (chunk order of a_chunk_pos)
//   mid    Kb * 2 bytes          SiLU * gate row, gathered from the granules
//   parts  kEngMaxQuadsB x 4 x 4 floats   K-slice partial sums of the second projection (and of the first when KSa > 1)
//   flags  ready[13] freed[13] norm[4] cnt[...]
struct EngLds {
    unsigned ring, xrow, mid, parts, flags, total;
};
__host__ __device__ inline EngLds eng_lds(int Ka, int Kb) {
    EngLds L;
    L.ring = 0;
    L.xrow = kEngSlots * kEngSlotBytes;
    L.mid = L.xrow + (((unsigned)Ka * 2 + 15) & ~15u);
    L.parts = L.mid + (((unsigned)Kb * 2 + 15) & ~15u);
    L.flags = L.parts + kEngMaxQuadsB * 4 * 4 * 4 * 2;      // two phases
    L.total = L.flags + 512;
    return L;
}
// flag words (unsigned) inside the flags block
enum { F_READY = 0, F_FREED = 16, F_NORM = 32, F_NORMCNT = 36, F_XDONE = 37, F_GATHER = 38, F_ENDCNT = 39, F_QCNT = 48 /* 2 x kEngMaxQuadsB */,
       F_WORDS = 48 + 2 * kEngMaxQuadsB };
static_assert(F_WORDS * 4 <= 512, "flag block");

template <typename T, bool STRICT>
__global__ __launch_bounds__(kEngThreads, 2) void w4_mlp_engine_kernel(const EngineArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const EngPlan pl = eng_plan(p);
    const EngLds L = eng_lds(p.Ka, p.Kb);
    const int ntasks = pl.nA + pl.nB;
    volatile unsigned* flags = reinterpret_cast<volatile unsigned*>(smem + L.flags);
    const unsigned lds_base = (unsigned)(uintptr_t)smem;        // LDS byte address of the dynamic region (low 32 bits of the flat pointer)

    // ---- loader: issue the first tasks before anything else happens in the workgroup ----------------------------------------
    auto issue_task = [&](int i) {
        const EngTask t = eng_task(p, pl, i);
        const int slot = i % kEngSlots;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + L.ring + (unsigned)slot * kEngSlotBytes));
        const int G = t.phase ? pl.Gb : pl.Ga;
        const u32x4* W = t.phase ? p.Wb : p.Wa;
        const char* S = reinterpret_cast<const char*>(t.phase ? p.Sb : p.Sa);
        const int last = t.ng > 0 ? t.ng - 1 : 0;
        const int l0 = min(lane, last), l1 = min(lane + 64, last);
        const u32x4* wq = W + ((int64_t)t.quad * 4) * G + t.g0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            glds16_nt(wq + (int64_t)c * G + l0, dst + c * 2048);
            glds16_nt(wq + (int64_t)c * G + l1, dst + c * 2048 + 1024);
        }
        // scales: ng x 8 bytes from byte offset 8 * (quad * G + g0), fetched as 16-byte units from the aligned floor
        const int64_t sbyte = ((int64_t)t.quad * G + t.g0) * 8;
        const int64_t sfloor = sbyte & ~(int64_t)15;
        const int units = (int)((sbyte - sfloor + (int64_t)t.ng * 8 + 15) >> 4);
        glds16_nt(S + sfloor + 16 * (int64_t)min(lane, units > 0 ? units - 1 : 0), dst + 8192);
    };

    if (wave == 0) {
        const int first = min(ntasks, kEngDepth);
        for (int i = 0; i < first; ++i) issue_task(i);
    }
    // flags start at zero (LDS keeps whatever the previous workgroup left): ONE workgroup barrier, before any poll
    if (tid < F_WORDS) flags[tid] = 0u;
    lds_fence();
    __builtin_amdgcn_s_barrier();
    ENG_STAMP(0);

    if (wave == 0) {
        // ================================================= LOADER ================================================================
        bool ok = true;
        int marked = 0;                                      // tasks whose ready flag has been set
        auto mark_oldest = [&](int issued) {                 // blocks until task `marked` has landed, then publishes it
            wait_tasks_in_flight(issued - marked - 1);
            if (lane == 0) flags[F_READY + marked % kEngSlots] = (unsigned)(marked + 1);
            ++marked;
        };
        for (int i = min(ntasks, kEngDepth); i < ntasks; ++i) {
            // the oldest task in flight has landed once at most kEngDepth - 1 tasks' loads are outstanding
            mark_oldest(i);
            if (i >= kEngSlots) {
                // ring full: while the consumers are busy (or away gathering), publish what lands instead of just sleeping
                const unsigned need = (unsigned)(i - kEngSlots + 1);
                unsigned spins = 0;
                while (flags[F_FREED + i % kEngSlots] != need) {
                    if (marked < i) mark_oldest(i);
                    else __builtin_amdgcn_s_sleep(1);
                    if (++spins > kEngSpinLds) {
                        ok = false;
                        break;
                    }
                }
                asm volatile("" ::: "memory");
            }
            issue_task(i);
        }
        for (; marked < ntasks; ++marked) {                  // drain: ntasks - marked tasks still in flight
            wait_tasks_in_flight(ntasks - marked - 1);
            if (lane == 0) flags[F_READY + marked % kEngSlots] = (unsigned)(marked + 1);
        }
        if (!ok && lane == 0) __hip_atomic_store(p.ws + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ENG_STAMP(1);
        return;
    }

    // =================================================== CONSUMERS =================================================================
    typedef Splice<T> SP;
    const int cw = wave - 1;                                 // consumer index 0..6
    u32 k_mask, k_mask_odd, k_magic;
    asm volatile("s_mov_b32 %0, %1" : "=s"(k_mask) : "i"(SP::kMask));
    asm volatile("s_mov_b32 %0, %1" : "=s"(k_mask_odd) : "i"(SP::kMaskOdd));
    asm volatile("v_mov_b32 %0, %1" : "=v"(k_magic) : "i"(SP::kMagic));
    bool ok = true;

    // launch epoch (tag of this launch's granules): requested now, needed at the first publish
    const unsigned tag = __hip_atomic_load(p.ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;

    // ---- RMSNorm of the input row into LDS: the arithmetic of the PRO_NORM prologue of w4_packed_gemv_16_body, whose 256
    //      threads are played by consumer waves 0..3 (virtual thread v = 64 cw + lane)
    {
        const int cpr = p.Ka >> 3;                           // 16-byte chunks of the row
        const int ach = (cpr + 255) >> 8;                    // chunks per virtual thread (2, 4, 8 there; any count here)
        float* nred = reinterpret_cast<float*>(smem + L.flags) + F_NORM;
        if (cw < 4) {
            const int vt = cw * 64 + lane;
            float ss = 0.f;
            for (int i = 0; i < ach; ++i) {
                const int c = vt + i * 256;
                if (c < cpr) {
                    float hv[8];
                    unpack8<T>(*reinterpret_cast<const u32x4*>(p.x + c * 8), hv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(hv[e], hv[e], ss);
                }
            }
            ss = wave_sum(ss);
            if (lane == 0) {
                nred[cw] = ss;
                lds_fence();
                __hip_atomic_fetch_add(const_cast<unsigned*>(flags + F_NORMCNT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            ok &= lds_wait_eq(flags + F_NORMCNT, 4u);
            const float r = rsqrtf(((nred[0] + nred[1]) + (nred[2] + nred[3])) / (float)p.Ka + p.eps);
            for (int i = 0; i < ach; ++i) {
                const int c = vt + i * 256;
                if (c < cpr) {
                    float hv[8], wv[8];
                    unpack8<T>(*reinterpret_cast<const u32x4*>(p.x + c * 8), hv);
                    unpack8<T>(*reinterpret_cast<const u32x4*>(p.ln_weight + c * 8), wv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = Act<T>::round(hv[e] * r) * wv[e];
                    *reinterpret_cast<u32x4*>(smem + L.xrow + (size_t)a_chunk_pos(c >> 2, c & 3) * 16) = pack8<T>(hv);
                }
            }
            lds_fence();
            if (lane == 0) __hip_atomic_fetch_add(const_cast<unsigned*>(flags + F_XDONE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        ok &= lds_wait_eq(flags + F_XDONE, 4u);
    }
    if (cw == 0) ENG_STAMP(2);

    // epilogue operands of this workgroup's second-projection quads (residual = the input row, bias): lane q holds quad q's,
    // requested now so that no global round trip sits in the tail of the launch
    u32x2 resid_q = {0u, 0u}, bias_q = {0u, 0u};
    {
        const int nqb = pl.qb1 - pl.qb0;
        const int q = pl.qb0 + min(lane, nqb > 0 ? nqb - 1 : 0);
        if (nqb > 0) {
            resid_q = *reinterpret_cast<const u32x2*>(p.x + q * 4);
            if (p.bias_b) bias_q = *reinterpret_cast<const u32x2*>(p.bias_b + q * 4);
        }
    }
    unsigned long long* granules = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.ws) + kEngWsHeaderBytes);
    float* parts = reinterpret_cast<float*>(smem + L.parts);

    // one task: the sums of (quad, K slice) over its groups, then the epilogue or the K-slice combine
    auto run_task = [&](int i) {
        const EngTask t = eng_task(p, pl, i);
        const int slot = i % kEngSlots;
        ok &= lds_wait_eq(flags + F_READY + slot, (unsigned)(i + 1));
        const char* sl = smem + L.ring + (size_t)slot * kEngSlotBytes;
        const int G = t.phase ? pl.Gb : pl.Ga;
        const char* arow = smem + (t.phase ? L.mid : L.xrow);
        const int soff = (int)((((int64_t)t.quad * G + t.g0) * 8) & 8);          // the scale fetch started at the 16-byte floor
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const int iters = (t.ng + 63) >> 6;
        for (int it = 0; it < iters; ++it) {
            const int gi = it * 64 + lane;
            const bool valid = gi < t.ng;
            const int gic = valid ? gi : t.ng - 1;                                 // in-range LDS addresses for the idle lanes
            const int g = valid ? t.g0 + gi : G - 1;                               // the clamp of the two-launch kernel (gc)
            u32x4 w[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) w[c] = *reinterpret_cast<const u32x4*>(sl + c * 2048 + gic * 16);
            u32x2 sv = *reinterpret_cast<const u32x2*>(sl + 8192 + soff + gic * 8);
            if (!valid) sv = u32x2{0u, 0u};
            u32 av[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 x = *reinterpret_cast<const u32x4*>(arow + (size_t)a_chunk_pos(g, j) * 16);
                av[4 * j + 0] = x[0];
                av[4 * j + 1] = x[1];
                av[4 * j + 2] = x[2];
                av[4 * j + 3] = x[3];
            }
            eng_tile_math<T, STRICT>(w, sv, av, acc, k_mask, k_mask_odd, k_magic);
        }
        lds_fence();                                                               // every read of the slot has returned
        if (lane == 0) flags[F_FREED + slot] = (unsigned)(i + 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = wave_sum(acc[c]);

        const int KS = t.phase ? p.KSb : p.KSa;
        if (KS > 1) {
            // K slices of a quad meet in LDS; whoever arrives last adds them in slice order (the order of the two-launch kernel)
            float* pq = parts + ((t.phase * kEngMaxQuadsB + t.qlocal) * 4 + t.ks) * 4;
            bool last = false;
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) pq[c] = acc[c];
                lds_fence();
                const unsigned before = __hip_atomic_fetch_add(const_cast<unsigned*>(flags + F_QCNT + t.phase * kEngMaxQuadsB + t.qlocal), 1u,
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                last = before == (unsigned)KS - 1u;
            }
            if (!__builtin_amdgcn_readfirstlane((int)last)) return;
            if (lane == 0) {
                const float* q0 = parts + ((t.phase * kEngMaxQuadsB + t.qlocal) * 4) * 4;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = q0[c];
                    for (int s = 1; s < KS; ++s) v += q0[s * 4 + c];
                    acc[c] = v;
                }
            }
        }
        const int ql = __builtin_amdgcn_readfirstlane(t.qlocal) & 63;
        const u32 rq0 = (u32)__builtin_amdgcn_readlane((int)resid_q[0], ql), rq1 = (u32)__builtin_amdgcn_readlane((int)resid_q[1], ql);
        const u32 bq0 = (u32)__builtin_amdgcn_readlane((int)bias_q[0], ql), bq1 = (u32)__builtin_amdgcn_readlane((int)bias_q[1], ql);
        if (lane != 0) return;
        const int n0 = t.quad * 4;
        if (t.phase == 0) {
            // SiLU(h) * gate on the quad's (h0, h1, gate0, gate1) sums -> ONE granule {pair of outputs, tag}
            float y[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                y[c] = Act<T>::round(acc[c]);
                if (p.bias_a) y[c] = Act<T>::round(y[c] + Act<T>::load(p.bias_a + n0 + c));
            }
            const float o0 = Act<T>::round(Act<T>::round(y[0] / (1.0f + __expf(-y[0]))) * y[2]);
            const float o1 = Act<T>::round(Act<T>::round(y[1] / (1.0f + __expf(-y[1]))) * y[3]);
            __hip_atomic_store(granules + t.quad, ((unsigned long long)tag << 32) | pack2<T>(o0, o1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // out = round(y + x), y = rounded sum (+ bias, rounded): the quad as one 8-byte store (Nb % 4 == 0)
            float rq[4], bq[4], y[4];
            unpack2<T>(rq0, rq[0], rq[1]);
            unpack2<T>(rq1, rq[2], rq[3]);
            unpack2<T>(bq0, bq[0], bq[1]);
            unpack2<T>(bq1, bq[2], bq[3]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                y[c] = Act<T>::round(acc[c]);
                if (p.bias_b) y[c] = Act<T>::round(y[c] + bq[c]);
                y[c] = y[c] + rq[c];
            }
            *reinterpret_cast<u32x2*>(p.out + n0) = u32x2{pack2<T>(y[0], y[1]), pack2<T>(y[2], y[3])};
        }
    };

    // ---- phase A ----------------------------------------------------------------------------------------------------------------
    for (int i = cw; i < pl.nA; i += kEngConsumers) run_task(i);
    if (cw == 0) ENG_STAMP(3);

    // ---- gather the (1, Kb) row: granule g carries elements 2g, 2g + 1.  Consumer cw sweeps passes cw, cw + 7, ... of 1024
    //      granules (16 relaxed agent-scope 8-byte loads per lane in flight), until every tag of the pass is this launch's
    {
        const int ngran = p.Kb >> 1;
        const int passes = (ngran + 1023) >> 10;
        for (int ps = cw; ps < passes; ps += kEngConsumers) {
            unsigned long long v[16];
            unsigned spins = 0;
            for (;;) {
                bool good = true;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int gidx = ps * 1024 + k * 64 + lane;
                    v[k] = __hip_atomic_load(granules + min(gidx, ngran - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) good &= (unsigned)(v[k] >> 32) == tag;
                if (__all(good)) break;
                if (++spins > kEngSpinGlobal) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int gidx = ps * 1024 + k * 64 + lane;
                if (gidx < ngran) {
                    const int cc = gidx >> 2;                                       // 16-byte chunk of the row (8 elements = 4 granules)
                    *reinterpret_cast<u32*>(smem + L.mid + (size_t)a_chunk_pos(cc >> 2, cc & 3) * 16 + (gidx & 3) * 4) = (u32)v[k];
                }
            }
        }
        lds_fence();
        if (lane == 0) __hip_atomic_fetch_add(const_cast<unsigned*>(flags + F_GATHER), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ok &= lds_wait_eq(flags + F_GATHER, (unsigned)kEngConsumers);
    }
    if (cw == 0) ENG_STAMP(4);

    // ---- phase B ----------------------------------------------------------------------------------------------------------------
    for (int j = cw; j < pl.nB; j += kEngConsumers) run_task(pl.nA + j);
    if (cw == 0) ENG_STAMP(5);

    // ---- end of launch: the last consumer of the last workgroup bumps the epoch (device memory, so graph replays see it)
    if (!ok && lane == 0) __hip_atomic_store(p.ws + 2, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) {
        const unsigned before = __hip_atomic_fetch_add(const_cast<unsigned*>(flags + F_ENDCNT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (before == (unsigned)kEngConsumers - 1u) {
            const unsigned done = __hip_atomic_fetch_add(p.ws + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (done == gridDim.x - 1u) {
                __hip_atomic_store(p.ws + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.ws, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (cw == 0) ENG_STAMP(6);
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
static int eng_cu_count() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return n;
    }();
    return cus;
}

size_t w4_mlp_engine_workspace_bytes(int64_t N_in) { return (size_t)kEngWsHeaderBytes + (size_t)(N_in / 4) * 8 + 64; }

bool w4_mlp_engine_supported(int64_t Na, int64_t Ka, int64_t Nb, int64_t Kb) {
    if (Na <= 0 || Ka <= 0 || Nb <= 0 || Kb <= 0 || Ka % 32 || Kb % 32 || Na % 4 || Nb % 4 || Kb * 2 != Na || Nb != Ka) return false;
    const int cus = eng_cu_count();
    if (cus <= 0) return false;
    const int ksa = w4_gemv_ksplit((Na + 3) / 4, Ka / 32), ksb = w4_gemv_ksplit((Nb + 3) / 4, Kb / 32);
    const int64_t gsa = (Ka / 32 + ksa - 1) / ksa, gsb = (Kb / 32 + ksb - 1) / ksb;
    if (gsa > kEngMaxGroupsPerTask || gsb > kEngMaxGroupsPerTask || ksa > 4 || ksb > 4) return false;
    const int64_t qa = (Na + 3) / 4, qb = (Nb + 3) / 4;
    // the K-slice combine slots bound the quads per workgroup of a phase that splits K; lane q of a consumer holds the epilogue
    // operands of second-projection quad q
    if ((ksa > 1 && (qa + cus - 1) / cus > kEngMaxQuadsB) || (qb + cus - 1) / cus > kEngMaxQuadsB) return false;
    return eng_lds((int)Ka, (int)Kb).total <= 160 * 1024;
}

template <typename T, bool STRICT>
static int launch_engine(const EngineArgs<T>& p, hipStream_t st) {
    static bool attr_set = false;
    const EngLds L = eng_lds(p.Ka, p.Kb);
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&w4_mlp_engine_kernel<T, STRICT>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    w4_mlp_engine_kernel<T, STRICT><<<(unsigned)eng_cu_count(), kEngThreads, L.total, st>>>(p);
    return finish_launch();
}

template <typename T>
static int engine_t(bool strict, const void* x, const void* ln_weight, float eps, const void* packed_a, const void* bias_a, int64_t Na,
                    int64_t Ka, const void* packed_b, const void* bias_b, int64_t Nb, int64_t Kb, void* out, void* ws, void* trace,
                    hipStream_t st) {
    const int64_t Ga = Ka / 32, Gb = Kb / 32, NpA = (Na + 3) & ~(int64_t)3, NpB = (Nb + 3) & ~(int64_t)3;
    EngineArgs<T> p;
    p.x = (const T*)x; p.ln_weight = (const T*)ln_weight; p.eps = eps;
    p.Wa = (const u32x4*)packed_a; p.Sa = (const T*)((const char*)packed_a + NpA * Ga * 16); p.bias_a = (const T*)bias_a;
    p.Na = (int)Na; p.Ka = (int)Ka; p.KSa = w4_gemv_ksplit(NpA / 4, Ga);
    p.Wb = (const u32x4*)packed_b; p.Sb = (const T*)((const char*)packed_b + NpB * Gb * 16); p.bias_b = (const T*)bias_b;
    p.Nb = (int)Nb; p.Kb = (int)Kb; p.KSb = w4_gemv_ksplit(NpB / 4, Gb);
    p.out = (T*)out; p.ws = (unsigned*)ws; p.trace = (unsigned long long*)trace;
    return strict ? launch_engine<T, true>(p, st) : launch_engine<T, false>(p, st);
}

int w4_mlp_engine(int dtype, bool strict, const void* x, const void* ln_weight, float eps, const void* packed_a, const void* bias_a,
                  int64_t Na, int64_t Ka, const void* packed_b, const void* bias_b, int64_t Nb, int64_t Kb, void* out, void* ws,
                  void* trace, hipStream_t st) {
    if (!w4_mlp_engine_supported(Na, Ka, Nb, Kb)) return QL_ERR_UNSUPPORTED;
    if (dtype == QL_DTYPE_F16) return engine_t<f16>(strict, x, ln_weight, eps, packed_a, bias_a, Na, Ka, packed_b, bias_b, Nb, Kb, out, ws, trace, st);
    if (dtype == QL_DTYPE_BF16) return engine_t<__bf16>(strict, x, ln_weight, eps, packed_a, bias_a, Na, Ka, packed_b, bias_b, Nb, Kb, out, ws, trace, st);
    return QL_ERR_BAD_DTYPE;
}

}  // namespace ql
