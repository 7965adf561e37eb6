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

constexpr int kEngWaves = 8;                     // 2 loaders + 6 consumers
constexpr int kEngLoaders = 2;                   // a wave keeps at most 63 loads in flight (vmcnt): ONE loader's 63 KB against ~3.7 us of
                                                 // loaded HBM latency is 17 GB/s per CU (measured: 16); two keep the ring's 13 slots in flight
constexpr int kEngConsumers = kEngWaves - kEngLoaders;
constexpr int kEngThreads = 64 * kEngWaves;
constexpr int kEngSlotBytes = 9216;              // 4 columns x 128 groups x 16 B + 1 KB of scales
constexpr int kEngSlots = 13;
constexpr int kEngDepth = 7;                     // tasks in flight per loader wave: 9 loads each, 63 = the vmcnt ceiling
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

// ---- LDS-DMA ------------------------------------------------------------------------------------------------------------------------
// M0 holds the LDS base of a transfer and is compiler-reserved: written and restored inside the statement
// (cdna_hip_programming.md 5.7).  The loads are invisible to hipcc's vmcnt bookkeeping: the loader counts its own.
// One TASK in one statement: 9 transfers of 1 KB to consecutive KBs of the slot at `lds_dst`.  Sources: four column segments
// (wave-uniform 64-bit bases b0..b3 + per-lane byte offsets v0 / v1 for the lanes' first / second 64 groups) and the scale
// segment (bs + vs).  saddr form: the address arithmetic stays on the scalar unit; M0 saved once, advanced by 1 KB per
// transfer (s_add_u32 writes SCC: clobbered), restored once - ~30 instructions per task instead of ~300 with one
// statement and 64-bit VALU address arithmetic per transfer (the first version's loader was instruction-bound at ~1 us per task).
__device__ __forceinline__ void glds_task(unsigned lds_dst, unsigned v0, unsigned v1, unsigned vs, unsigned long long b0,
                                          unsigned long long b1, unsigned long long b2, unsigned long long b3, unsigned long long bs) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %7 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %7 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %8 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %8 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %9 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(v0), "v"(v1), "v"(vs), "s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(bs)
        : "memory", "scc");
}

// a wave-uniform 64-bit value the compiler may not recognise as such -> an SGPR pair
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
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

// Flag words live in LDS and are polled: the pointer type carries the LDS address space explicitly.  (A `volatile unsigned*`
// derived from the dynamic-LDS array stays a GENERIC pointer - LLVM's address-space inference does not rewrite volatile
// accesses - and every poll became a flat_load + s_waitcnt vmcnt(0): in the loader that drained the whole LDS-DMA queue
// per poll, 54 us per MLP.)
typedef __attribute__((address_space(3))) volatile unsigned lds_vu32;
typedef __attribute__((address_space(3))) unsigned lds_u32;
__device__ __forceinline__ unsigned lds_add(lds_vu32* p, unsigned v) {
    return __hip_atomic_fetch_add((lds_u32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// spin until *flag == want (LDS word written by another wave of the workgroup); false after kEngSpinLds polls
__device__ __forceinline__ bool lds_wait_eq(lds_vu32* flag, unsigned want) {
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
enum { F_READY = 0, F_FREED = 16, F_NORM = 32, F_NORMCNT = 36, F_XDONE = 37, F_GATHER = 38, F_ENDCNT = 39 /* + 40: trace */, F_NEXT = 41 /* + 42 */, F_QCNT = 48 /* 2 x kEngMaxQuadsB */,
       F_WORDS = 48 + 2 * kEngMaxQuadsB };
static_assert(F_WORDS * 4 <= 512, "flag block");

template <typename T, bool STRICT>
__global__ __launch_bounds__(kEngThreads, 2) void w4_mlp_engine_kernel(const EngineArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // This workgroup's share, as plain scalars (a struct selected by phase made hipcc build a lookup table in SCRATCH: a
    // scratch_load + s_waitcnt vmcnt(0) per task in the loader drained its whole LDS-DMA queue - 54 us per MLP).
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    const int qa_all = (p.Na + 3) >> 2, qb_all = (p.Nb + 3) >> 2;
    const int qa0 = (int)((long long)qa_all * b / nb), nqa = (int)((long long)qa_all * (b + 1) / nb) - qa0;
    const int qb0 = (int)((long long)qb_all * b / nb), nqb = (int)((long long)qb_all * (b + 1) / nb) - qb0;
    const int KSa = p.KSa, KSb = p.KSb;
    const int Ga = p.Ka >> 5, Gb = p.Kb >> 5;
    const int gsa = (Ga + KSa - 1) / KSa, gsb = (Gb + KSb - 1) / KSb;
    const int nA = nqa * KSa, nB = nqb * KSb, ntasks = nA + nB;
    const EngLds L = eng_lds(p.Ka, p.Kb);
    lds_vu32* flags = (lds_vu32*)(smem + L.flags);
    const unsigned lds_base = (unsigned)(uintptr_t)smem;        // LDS byte address of the dynamic region (low 32 bits of the flat pointer)

    // flags start at zero (LDS keeps whatever the previous workgroup left): written now, ONE workgroup barrier before any poll.
    // The loader reaches its barrier only after it has issued its first tasks.
    if (tid < F_WORDS) flags[tid] = 0u;
    lds_fence();

    if (wave < kEngLoaders) {
        // ================================================= LOADERS ===============================================================
        // loader w streams tasks w, w + 2, ... of the workgroup's list; issued / marked count ITS tasks, `gtask` all of them
        bool ok = true, synced = false;
        int issued = 0, marked = 0, slot = 0, gtask = 0;
        const int mine = (ntasks + kEngLoaders - 1 - wave) / kEngLoaders;
#ifdef QL_ENGINE_TRACE
        unsigned long long blocked = 0;
#endif
        const int first = min(mine, (kEngSlots - 1) / kEngLoaders);   // issued before the barrier: no slot is reused yet
        auto mark_oldest = [&]() {                           // blocks until this loader's oldest task has landed, then publishes it
            wait_tasks_in_flight(issued - marked - 1);
            const int g = marked * kEngLoaders + wave;       // its index in the workgroup's list
            if (lane == 0) flags[F_READY + g % kEngSlots] = (unsigned)(g + 1);
            ++marked;
        };
        auto step = [&](int quad, int ks, int G, int gs, unsigned long long W, unsigned long long S) {
            if (gtask % kEngLoaders != wave) {               // the other loader's task
                ++gtask;
                if (++slot == kEngSlots) slot = 0;
                return;
            }
            if (issued == first && !synced) {                // the first tasks are in flight: now join the workgroup's barrier
                __builtin_amdgcn_s_barrier();
                synced = true;
            }
            if (issued >= kEngDepth) {
                mark_oldest();                               // at most kEngDepth - 1 tasks stay in flight
            }
            {
                if (gtask >= kEngSlots) {
                    // ring full: while the consumers are busy (or away gathering), publish what lands instead of just sleeping
                    const unsigned need = (unsigned)(gtask - kEngSlots + 1);
                    unsigned spins = 0;
#ifdef QL_ENGINE_TRACE
                    const unsigned long long tb0 = __builtin_amdgcn_s_memrealtime();
#endif
                    while (flags[F_FREED + slot] != need) {
                        if (marked < issued) mark_oldest();
                        else __builtin_amdgcn_s_sleep(1);
                        if (++spins > kEngSpinLds) {
                            ok = false;
                            break;
                        }
                    }
                    asm volatile("" ::: "memory");
#ifdef QL_ENGINE_TRACE
                    blocked += __builtin_amdgcn_s_memrealtime() - tb0;
#endif
                }
            }
            const int g0 = ks * gs;
            const int ng = max(min(G, g0 + gs) - g0, 1);
            const unsigned long long b0 = W + ((unsigned long long)quad * 4ull * (unsigned)G + (unsigned)g0) * 16ull;
            // scales: ng x 8 bytes from byte offset 8 * (quad * G + g0), fetched as 16-byte units from the aligned floor
            const unsigned long long sbyte = ((unsigned long long)quad * (unsigned)G + (unsigned)g0) * 8ull;
            const unsigned long long sfloor = sbyte & ~15ull;
            const int units = (int)((sbyte - sfloor + (unsigned long long)ng * 8ull + 15ull) >> 4);
            const unsigned v0 = (unsigned)min(lane, ng - 1) * 16u, v1 = (unsigned)min(lane + 64, ng - 1) * 16u;
            const unsigned vs = (unsigned)min(lane, units - 1) * 16u;
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + L.ring + (unsigned)slot * kEngSlotBytes));
            const unsigned long long ub0 = uniform64(b0), ucol = uniform64((unsigned long long)G * 16ull);
            glds_task(dst, v0, v1, vs, ub0, ub0 + ucol, ub0 + 2ull * ucol, ub0 + 3ull * ucol, uniform64(S + sfloor));
            ++issued;
            ++gtask;
            if (++slot == kEngSlots) slot = 0;
        };
        for (int ql = 0; ql < nqa; ++ql)
            for (int ks = 0; ks < KSa; ++ks) step(qa0 + ql, ks, Ga, gsa, (unsigned long long)(uintptr_t)p.Wa, (unsigned long long)(uintptr_t)p.Sa);
        for (int ql = 0; ql < nqb; ++ql)
            for (int ks = 0; ks < KSb; ++ks) step(qb0 + ql, ks, Gb, gsb, (unsigned long long)(uintptr_t)p.Wb, (unsigned long long)(uintptr_t)p.Sb);
        if (!synced) __builtin_amdgcn_s_barrier();
        if (wave == 0) ENG_STAMP(0);
        while (marked < issued) mark_oldest();               // drain
        if (!ok && lane == 0) __hip_atomic_store(p.ws + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef QL_ENGINE_TRACE
        if (wave == 0) {
            ENG_STAMP(1);
            if (lane == 0 && p.trace) p.trace[(int)blockIdx.x * kEngTraceWords + 15] = blocked;
        }
#endif
        return;
    }

    // =================================================== CONSUMERS =================================================================
    typedef Splice<T> SP;
    const int cw = wave - kEngLoaders;                       // consumer index
    // consumers 0..3 play the 256 threads of the two-launch kernel's RMSNorm prologue: their chunks of the input row and of the
    // norm weight are requested now, in front of the barrier (the row was written by the previous launch: an L2 / MALL hit)
    constexpr int kMaxAch = 8;                               // K <= 16384
    const int cpr = p.Ka >> 3;                               // 16-byte chunks of the row
    const int ach = (cpr + 255) >> 8;                        // chunks per virtual thread
    u32x4 xin[kMaxAch], lnw[kMaxAch];
    if (cw < 4) {
        const int vt = cw * 64 + lane;
#pragma unroll
        for (int i = 0; i < kMaxAch; ++i) {
            const int c = vt + i * 256;
            if (i < ach && c < cpr) {
                xin[i] = *reinterpret_cast<const u32x4*>(p.x + c * 8);
                lnw[i] = *reinterpret_cast<const u32x4*>(p.ln_weight + c * 8);
            }
        }
    }
    // launch epoch (tag of this launch's granules): requested now, needed at the first publish
    const unsigned tag = __hip_atomic_load(p.ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    // epilogue operands of this workgroup's second-projection quads (residual = the input row, bias): lane q holds quad q's,
    // requested now so that no global round trip sits in the tail of the launch
    u32x2 resid_q = {0u, 0u}, bias_q = {0u, 0u};
    if (nqb > 0) {
        const int q = qb0 + min(lane, nqb - 1);
        resid_q = *reinterpret_cast<const u32x2*>(p.x + q * 4);
        if (p.bias_b) bias_q = *reinterpret_cast<const u32x2*>(p.bias_b + q * 4);
    }
    __builtin_amdgcn_s_barrier();
    if (cw == 0) ENG_STAMP(10);

    u32 k_mask, k_mask_odd, k_magic;
    asm volatile("s_mov_b32 %0, %1" : "=s"(k_mask) : "i"(SP::kMask));
    asm volatile("s_mov_b32 %0, %1" : "=s"(k_mask_odd) : "i"(SP::kMaskOdd));
    asm volatile("v_mov_b32 %0, %1" : "=v"(k_magic) : "i"(SP::kMagic));
    bool ok = true;

    // ---- RMSNorm of the input row into LDS: the arithmetic of the PRO_NORM prologue of w4_packed_gemv_16_body, whose 256
    //      threads are played by consumer waves 0..3 (virtual thread v = 64 cw + lane)
    {
        float* nred = reinterpret_cast<float*>(smem + L.flags) + F_NORM;
        if (cw < 4) {
            const int vt = cw * 64 + lane;
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < kMaxAch; ++i) {
                const int c = vt + i * 256;
                if (i < ach && c < cpr) {
                    float hv[8];
                    unpack8<T>(xin[i], hv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(hv[e], hv[e], ss);
                }
            }
            ss = wave_sum(ss);
            if (lane == 0) {
                nred[cw] = ss;
                lds_fence();
                lds_add(flags + F_NORMCNT, 1u);
            }
            ok &= lds_wait_eq(flags + F_NORMCNT, 4u);
            const float r = rsqrtf(((nred[0] + nred[1]) + (nred[2] + nred[3])) / (float)p.Ka + p.eps);
#pragma unroll
            for (int i = 0; i < kMaxAch; ++i) {
                const int c = vt + i * 256;
                if (i < ach && c < cpr) {
                    float hv[8], wv[8];
                    unpack8<T>(xin[i], hv);
                    unpack8<T>(lnw[i], wv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = Act<T>::round(hv[e] * r) * wv[e];
                    *reinterpret_cast<u32x4*>(smem + L.xrow + (size_t)a_chunk_pos(c >> 2, c & 3) * 16) = pack8<T>(hv);
                }
            }
            lds_fence();
            if (lane == 0) lds_add(flags + F_XDONE, 1u);
        }
        ok &= lds_wait_eq(flags + F_XDONE, 4u);
    }
    if (cw == 0) ENG_STAMP(2);

    unsigned long long* granules = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.ws) + kEngWsHeaderBytes);
    float* parts = reinterpret_cast<float*>(smem + L.parts);

#ifdef QL_ENGINE_TRACE
    unsigned long long waited[2] = {0, 0}, ntask[2] = {0, 0};   // consumer 0: time spent waiting for a landed slot, tasks run
#endif
    // one task (stream index i): the sums of (quad, K slice) over its groups, then the epilogue or the K-slice combine.
    // Every per-phase quantity arrives as a scalar argument (see the note on scratch above).
    auto run_task = [&](int i, const int phase, int quad, int qlocal, int ks, int G, int gs, int KS, unsigned arow_off) {
        const int slot = i % kEngSlots;
        const int g0 = ks * gs;
        const int ng = max(min(G, g0 + gs) - g0, 1);
#ifdef QL_ENGINE_TRACE
        const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
#endif
        ok &= lds_wait_eq(flags + F_READY + slot, (unsigned)(i + 1));
#ifdef QL_ENGINE_TRACE
        waited[phase] += __builtin_amdgcn_s_memrealtime() - tw0;
        ++ntask[phase];
#endif
        const char* sl = smem + L.ring + (size_t)slot * kEngSlotBytes;
        const char* arow = smem + arow_off;
        const int soff = (int)((((int64_t)quad * G + g0) * 8) & 8);               // the scale fetch started at the 16-byte floor
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const int iters = (ng + 63) >> 6;
        // The slot's bytes go to registers first (2 tiles x (4 x 16 B + 8 B) per lane) and the slot is handed back to the loader
        // BEFORE the math: a slot held for the ~2.5 us of a task's v_dot2c work left the loader 6 free slots of 13 and stalled
        // the stream (timeline in profiles/r03_mlp_engine.txt).
        u32x4 w[2][4];
        u32x2 sv[2];
        int gq[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int gi = it * 64 + lane;
            const bool valid = gi < ng;
            const int gic = valid ? gi : ng - 1;                                   // in-range LDS addresses for the idle lanes
            gq[it] = valid ? g0 + gi : G - 1;                                      // the clamp of the two-launch kernel (gc)
#pragma unroll
            for (int c = 0; c < 4; ++c) w[it][c] = *reinterpret_cast<const u32x4*>(sl + c * 2048 + gic * 16);
            sv[it] = *reinterpret_cast<const u32x2*>(sl + 8192 + soff + gic * 8);
            if (!valid) sv[it] = u32x2{0u, 0u};
        }
        lds_fence();                                                               // every read of the slot has returned
        if (lane == 0) flags[F_FREED + slot] = (unsigned)(i + 1);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if (it < iters) {
                u32 av[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 x = *reinterpret_cast<const u32x4*>(arow + (size_t)a_chunk_pos(gq[it], j) * 16);
                    av[4 * j + 0] = x[0];
                    av[4 * j + 1] = x[1];
                    av[4 * j + 2] = x[2];
                    av[4 * j + 3] = x[3];
                }
#ifdef QL_ENGINE_NOMATH        // timing ablation (results wrong): the consumers touch the operands and skip the dequant + dot work
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[c] += u32_as_f32((w[it][c][0] ^ w[it][c][1] ^ w[it][c][2] ^ w[it][c][3] ^ sv[it][0] ^ sv[it][1] ^ av[c]) & 0x3fffffffu);
#else
                eng_tile_math<T, STRICT>(w[it], sv[it], av, acc, k_mask, k_mask_odd, k_magic);
#endif
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = wave_sum(acc[c]);

        if (KS > 1) {
            // K slices of a quad meet in LDS; whoever arrives last adds them in slice order (the order of the two-launch kernel)
            float* q0 = parts + ((phase * kEngMaxQuadsB + qlocal) * 4) * 4;
            bool last = false;
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) q0[ks * 4 + c] = acc[c];
                lds_fence();
                const unsigned before = lds_add(flags + F_QCNT + phase * kEngMaxQuadsB + qlocal, 1u);
                last = before == (unsigned)KS - 1u;
            }
            if (!__builtin_amdgcn_readfirstlane((int)last)) return;
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = q0[c];
                    for (int s = 1; s < KS; ++s) v += q0[s * 4 + c];
                    acc[c] = v;
                }
            }
        }
        const int n0 = quad * 4;
        if (phase == 0) {
            if (lane != 0) return;
            // SiLU(h) * gate on the quad's (h0, h1, gate0, gate1) sums -> ONE granule {pair of outputs, tag}
            float y[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                y[c] = Act<T>::round(acc[c]);
                if (p.bias_a) y[c] = Act<T>::round(y[c] + Act<T>::load(p.bias_a + n0 + c));
            }
            const float o0 = Act<T>::round(Act<T>::round(y[0] / (1.0f + __expf(-y[0]))) * y[2]);
            const float o1 = Act<T>::round(Act<T>::round(y[1] / (1.0f + __expf(-y[1]))) * y[3]);
            __hip_atomic_store(granules + quad, ((unsigned long long)tag << 32) | pack2<T>(o0, o1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const int ql = __builtin_amdgcn_readfirstlane(qlocal) & 63;
            const u32 rq0 = (u32)__builtin_amdgcn_readlane((int)resid_q[0], ql), rq1 = (u32)__builtin_amdgcn_readlane((int)resid_q[1], ql);
            const u32 bq0 = (u32)__builtin_amdgcn_readlane((int)bias_q[0], ql), bq1 = (u32)__builtin_amdgcn_readlane((int)bias_q[1], ql);
            if (lane != 0) return;
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
    // tasks are CLAIMED (one LDS atomic each), not dealt: the consumer that shares its SIMD with the loader runs faster than
    // the pairs that share one, and a static deal made the slow pairs finish microseconds after the others
    auto claim = [&](int word) {
        unsigned i = 0;
        if (lane == 0) i = lds_add(flags + word, 1u);
        return __builtin_amdgcn_readfirstlane((int)i);
    };
    for (int i = claim(F_NEXT); i < nA; i = claim(F_NEXT)) {
        const int ql = i / KSa;
        run_task(i, 0, qa0 + ql, ql, i - ql * KSa, Ga, gsa, KSa, L.xrow);
    }
    if (cw == 0) ENG_STAMP(3);
#ifdef QL_ENGINE_TRACE
    if (lane == 0 && lds_add(flags + F_ENDCNT + 1, 1u) ==
                         (unsigned)kEngConsumers - 1u)
        ENG_STAMP(7);                                        // the LAST consumer of the workgroup finished phase A
    unsigned sweeps = 0;
#endif

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
#ifdef QL_ENGINE_TRACE
                ++sweeps;
#endif
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
#ifdef QL_ENGINE_TRACE
        if (cw == 0) {
            ENG_STAMP(8);                                    // consumer 0's own sweep complete
            if (lane == 0 && p.trace) p.trace[(int)blockIdx.x * kEngTraceWords + 9] = sweeps;
        }
#endif
        lds_fence();
        if (lane == 0) lds_add(flags + F_GATHER, 1u);
        ok &= lds_wait_eq(flags + F_GATHER, (unsigned)kEngConsumers);
    }
    if (cw == 0) ENG_STAMP(4);

    // ---- phase B ----------------------------------------------------------------------------------------------------------------
    for (int j = claim(F_NEXT + 1); j < nB; j = claim(F_NEXT + 1)) {
        const int ql = j / KSb;
        run_task(nA + j, 1, qb0 + ql, ql, j - ql * KSb, Gb, gsb, KSb, L.mid);
    }
    if (cw == 0) ENG_STAMP(5);

    // ---- end of launch: the last consumer of the last workgroup bumps the epoch (device memory, so graph replays see it)
    if (!ok && lane == 0) __hip_atomic_store(p.ws + 2, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) {
        const unsigned before = lds_add(flags + F_ENDCNT, 1u);
        if (before == (unsigned)kEngConsumers - 1u) {
            const unsigned done = __hip_atomic_fetch_add(p.ws + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (done == gridDim.x - 1u) {
                __hip_atomic_store(p.ws + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.ws, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (cw == 0) ENG_STAMP(6);
#ifdef QL_ENGINE_TRACE
    if (cw == 0 && lane == 0 && p.trace) {
        unsigned long long* tr = p.trace + (int)blockIdx.x * kEngTraceWords;
        tr[11] = waited[0]; tr[12] = waited[1]; tr[13] = ntask[0]; tr[14] = ntask[1];
    }
#endif
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
    if (cus <= 0 || Ka > 16384) return false;
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
