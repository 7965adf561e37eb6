// int4g32 GEMM for many activation rows (prefill / batched decode) on the derived layout, gfx950.
//
//   C[M,N] = A[M,K] . dequant(W)      fp16 / bf16 activations, fp32 accumulation on MFMA
//
// This IS a dense GEMM (the reference itself runs it on tl.dot, chatglm_q/int4/triton_ops.py:75), and it
// keeps the reference's arithmetic exactly: every weight is dequantised to (n - 8) * s ROUNDED to the
// activation dtype (triton_ops.py:72-73) in registers and fed to v_mfma_f32_32x32x16_{f16,bf16}, whose
// products are exact and whose accumulation is fp32.
//
// Decomposition: block = 4 waves side by side in N (128 columns) x BM = 32 MT rows.  A wave owns 32
// columns: lane (j = lane & 31, kb = lane >> 5) loads ONE 16-byte unit per 64-deep K step (from the TILE-MAJOR
// part of the derived layout, launch.h: the wave's 64 units of a step are 1 KB contiguous) - the 32
// nibbles of column j, group 2 kt + kb - and turns word s of it into the B fragment of MFMA sub-step s
// (the two half-waves supply the two groups of the step, so an MFMA's K = 16 is octet s of group 2kt
// and octet s of group 2kt+1; the A fragment uses the same assignment, any consistent K permutation
// gives the same sum).  The 13 VALU ops that build a B fragment are reused by MT MFMAs.
// The A tile (BM x 64 halves) is staged through LDS by all 4 waves (global -> registers one step ahead ->
// ds_write_b128 after the math), double buffered, one barrier per K step; 16-byte chunk c of row r lives
// at chunk position 8 r + (c ^ ((r >> 1) & 7)), which makes the 128-byte-pitch ds_read_b128 fragment
// reads bank-conflict free (16-lane service groups, 64-bank rows).
#include "launch.h"
#include "w4_dequant.h"
#include "w4_mma.h"

// Developer ablation switches (tools/microbench/gemm_ablate.hip compiles this file with -DQL_GEMM_ABLATE=bits to
// attribute loop time; always 0 in the library): 1 no dequant, 2 no steady-state weight/scale loads, 4 no MFMA,
// 8 no steady-state A loads / LDS stores, 16 no LDS fragment reads, 32 no barriers, 64 cached (not nt) weight loads.
#ifndef QL_GEMM_ABLATE
#define QL_GEMM_ABLATE 0
#endif

namespace ql {

// Build-time switches with their measurements (tools/ab, 8192 rows, TFLOP/s for qkv / o / w_in / w_out; product: 936 / 989 / 944 / 1070):
//   QL_GEMM_PIN=1    MFMA / dequant interleave pinned instruction by instruction (fragment reads, then MFMA + one dword of the
//                    next B fragment, sched_barrier after each pair): 903 / 962 / 920 / 1037 - the order hipcc finds from the
//                    sched_group_barrier hints is not what holds the kernel back
//   QL_GEMM_DEPTH    register ring depth (K steps of global loads in flight): 2: 919 / 989 / 942 / 1010, 4: 965 / 1025 / 989 / 1107
//                    against 957 / 1015 / 982 / 1089 on the same run (noise level)
//   QL_GEMM_WPE=2    amdgpu_waves_per_eu(2, 2): 888 / 944 / 924 / 1016 against 922 / 967 / 937 / 1037
#ifndef QL_GEMM_PIN
#define QL_GEMM_PIN 0
#endif
#ifndef QL_GEMM_DEPTH
#define QL_GEMM_DEPTH 3
#endif
#ifndef QL_GEMM_WPE
#define QL_GEMM_WPE 0
#endif
template <typename T, int MT, int NT, int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64)
#if QL_GEMM_WPE
__attribute__((amdgpu_waves_per_eu(QL_GEMM_WPE, QL_GEMM_WPE)))
#endif
void w4_packed_gemm_kernel(const T* __restrict__ A, const u32x4* __restrict__ Wt,
                                                                 const T* __restrict__ Sp, int M, int N, int K, int G,
                                                                 int64_t lda, int per, int nbx, const T* __restrict__ bias,
                                                                 T* __restrict__ C, int64_t ldc, float* __restrict__ part) {
    // (argument order: the leading 14 dwords - what the first loads need - are preloaded into SGPRs at wave launch)
    constexpr int BM = 32 * MT;                // rows per block
    constexpr int WN = 32 * NT;                // columns per wave
    constexpr int NTHR = NW * 64;
    constexpr int CH = (BM * 8 + NTHR - 1) / NTHR;   // 16-byte A chunks staged per thread per K step
    constexpr bool kAllStage = (BM * 8) % NTHR == 0;
    typedef Mma<T> MM;
    __shared__ __attribute__((aligned(16))) char smem[2][BM * 128];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kb = lane >> 5;
    // 1-D grid over (row tile, column tile), remapped so that each XCD (hardware block id mod 8) owns a contiguous
    // run of row-major tile indices: the column tiles that share an A row tile then hit ONE L2 instead of eight
    const TileXY tile = xcd_tile(blockIdx.x, gridDim.x, nbx);
    const int m0 = tile.y * BM;
    const int n_base = tile.x * (NW * WN) + wave * WN + j;       // + 32 t for the wave's t-th column tile
    const int ksteps = (G + 1) >> 1;           // 64 k (two groups) per step
    const int k0 = blockIdx.z * per;           // split-K: this block's K steps are [k0, k0 + nst)
    const int nst = ksteps - k0 < per ? ksteps - k0 : per;
    auto gstep = [&](int t) { return k0 + (t < nst ? t : nst - 1); };

    u32 k_mask_lo, k_mask_hi, k_magic;
    asm volatile("s_mov_b32 %0, 0x000F000F" : "=s"(k_mask_lo));
    asm volatile("s_mov_b32 %0, 0x00F000F0" : "=s"(k_mask_hi));
    asm volatile("v_mov_b32 %0, %1" : "=v"(k_magic) : "i"(MM::kMagic));

    // tile-major part of the derived layout (launch.h): the wave's units of a K step are 1 KB contiguous
    const int ctiles = (N + 31) >> 5;
    const u32x4* wcol[NT];
    const T* scol[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ct_raw = (tile.x * NW + wave) * NT + t;
        const int ct = ct_raw < ctiles ? ct_raw : ctiles - 1;   // clamped column tile: loads stay in bounds, stores are masked
        wcol[t] = Wt + (int64_t)ct * ksteps * 64 + lane;
        scol[t] = Sp + (int64_t)ct * ksteps * 64 + lane;
    }

    // A staging: thread -> CH chunks; chunk q: row q / 8, 16-byte column q % 8
    const T* a_src[CH];
    int a_dst[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        const int q = tid + u * NTHR;
        const int r = (q >> 3) % BM, c = q & 7;
        const int row = (m0 + r < M) ? (m0 + r) : (M - 1);
        a_src[u] = A + (int64_t)row * lda + c * 8;
        a_dst[u] = (r * 8 + (c ^ ((r >> 1) & 7))) * 16;
    }
    const int kmax = K - 8;                    // last in-bounds 8-half chunk start

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.f;

    // Register ring, DEPTH K-steps deep: the global loads (A chunks, this lane's weight units AND their scales) of
    // step kt + DEPTH are issued when step kt starts, so DEPTH - 1 steps of MFMA work cover their latency.  In
    // the steady-state loop every load is unconditional and real, so hipcc's vmcnt waits are exact (a load
    // under `if (more)` made it drain the queue every step; the per-step scale load sat on the critical path).
    typedef decltype(MM::scale_pair(scol[0], true)) scale_t;
    struct Stage {
        u32x4 a[CH];
        u32x4 w[NT];
        T s[NT];
    };
    Stage st[DEPTH];
    auto load_stage = [&](int kt, Stage& sg, bool steady = false) {
        if (!(steady && (QL_GEMM_ABLATE & 8))) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int c = (tid + u * NTHR) & 7;
                const int k = kt * 64 + c * 8;
                // a K tail (odd group count) reads a clamped chunk; its weights are zeroed at use
                sg.a[u] = *reinterpret_cast<const u32x4*>(a_src[u] + (k <= kmax ? kt * 64 : kmax - c * 8));
            }
        }
        if (steady && (QL_GEMM_ABLATE & 2)) return;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // plain (cacheable) loads: other row blocks re-read the same units
            sg.w[t] = wcol[t][(int64_t)kt * 64];
            sg.s[t] = scol[t][(int64_t)kt * 64];
        }
    };
    auto store_a = [&](int buf, const Stage& sg, bool steady = false) {
        if (steady && (QL_GEMM_ABLATE & 8)) return;
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (kAllStage || tid + u * NTHR < BM * 8) *reinterpret_cast<u32x4*>(smem[buf] + a_dst[u]) = sg.a[u];
    };
    auto block_sync = [&] {
        if (!(QL_GEMM_ABLATE & 32)) __syncthreads();
    };
    auto mma_step = [&](int buf, int kt, const u32x4 (&w_cur)[NT], const T (&s_raw)[NT]) {
        const int g = 2 * kt + kb;
        scale_t s[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const T s_eff = g < G ? s_raw[t] : (T)0.f;       // the missing half of an odd last step contributes 0
            s[t] = MM::scale_pair(&s_eff, true);
        }
        // A fragments are read ONE SUB-STEP AHEAD of the MFMAs that consume them: with the reads issued right in
        // front of each MFMA (what the straightforward loop compiled to) every MFMA waited out a full LDS round
        // trip (~100+ cycles against its own 32) and the matrix pipe idled 70 % of the time
        auto read_a = [&](int sub, u32x4 (&fr)[MT]) {
            const int c = kb * 4 + sub;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r = mt * 32 + j;
                if (QL_GEMM_ABLATE & 16) fr[mt] = u32x4{(u32)r, (u32)c, (u32)buf, 0x3c003c00u};
                else fr[mt] = *reinterpret_cast<const u32x4*>(smem[buf] + (r * 8 + (c ^ ((r >> 1) & 7))) * 16);
            }
        };
        auto deq = [&](u32 w, scale_t sc) {
            if (QL_GEMM_ABLATE & 1) return __builtin_bit_cast(typename MM::frag, u32x4{w, w ^ k_magic, w, w});
            return MM::dequant(w, k_mask_lo, k_mask_hi, k_magic, sc);
        };
        u32x4 fa[2][MT];
#if QL_GEMM_PIN
        if constexpr (NT == 1 && !QL_GEMM_ABLATE) {
            // exact instruction order, pinned with sched_barrier: the MT fragment reads of the next sub-step, then each MFMA
            // followed by its slice of the next B fragment's dequant (4 dwords over MT MFMAs)
            u32x4 fbr[2];
            read_a(0, fa[0]);
#pragma unroll
            for (int i = 0; i < 4; ++i) fbr[0][i] = MM::dequant_part(w_cur[0][0], i, k_mask_lo, k_mask_hi, k_magic, s[0]);
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                if (sub < 3) read_a(sub + 1, fa[(sub + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[mt][0] = MM::mma(__builtin_bit_cast(typename MM::frag, fa[sub & 1][mt]),
                                         __builtin_bit_cast(typename MM::frag, fbr[sub & 1]), acc[mt][0]);
                    if (sub < 3) {
                        constexpr int PER = MT >= 4 ? 1 : 4 / MT;
#pragma unroll
                        for (int i = mt * PER; i < (mt + 1) * PER && i < 4; ++i)
                            fbr[(sub + 1) & 1][i] = MM::dequant_part(w_cur[0][sub + 1], i, k_mask_lo, k_mask_hi, k_magic, s[0]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            return;
        }
#endif
        typename MM::frag fb[2][NT];
        read_a(0, fa[0]);
#pragma unroll
        for (int t = 0; t < NT; ++t) fb[0][t] = deq(w_cur[t][0], s[t]);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            // software pipeline inside the step: while the MFMAs of sub-step `sub` run, issue the A-fragment reads
            // and build the B fragments of sub-step sub + 1 (VALU slots under the MFMA shadows)
            if (sub < 3) {
                read_a(sub + 1, fa[(sub + 1) & 1]);
#pragma unroll
                for (int t = 0; t < NT; ++t) fb[(sub + 1) & 1][t] = deq(w_cur[t][sub + 1], s[t]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (QL_GEMM_ABLATE & 4) {
                        const u32x4 fbu = __builtin_bit_cast(u32x4, fb[sub & 1][t]);
                        acc[mt][t][sub] += u32_as_f32(fa[sub & 1][mt][0] ^ fa[sub & 1][mt][1] ^ fa[sub & 1][mt][2] ^
                                                      fa[sub & 1][mt][3] ^ fbu[0] ^ fbu[1] ^ fbu[2] ^ fbu[3]);
                    } else
                        acc[mt][t] = MM::mma(__builtin_bit_cast(typename MM::frag, fa[sub & 1][mt]), fb[sub & 1][t], acc[mt][t]);
                }
            // schedule shape per sub-step: the MT reads first, then each MFMA followed by a slice of the VALU work
            __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);            // DS reads of the next sub-step
#pragma unroll
            for (int q = 0; q < MT * NT; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, (16 + MT - 1) / MT, 0);   // its share of the dequant VALU
            }
        }
    };

#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_stage(gstep(d), st[d]);
    store_a(0, st[0]);
    __syncthreads();

    int kt = 0;
    for (; kt + DEPTH < nst; kt += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int buf = (kt + d) & 1;
            u32x4 w_cur[NT];
            T s_cur[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                w_cur[t] = st[d].w[t];
                s_cur[t] = st[d].s[t];
            }
            // slot d: its A chunks went to LDS last step, its weight units / scales were just copied out
            load_stage(gstep(kt + d + DEPTH), st[d], true);
            mma_step(buf, gstep(kt + d), w_cur, s_cur);
            store_a(buf ^ 1, st[(d + 1) % DEPTH], true);
            block_sync();
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        if (kt + d < nst) {
            const int buf = (kt + d) & 1;
            mma_step(buf, gstep(kt + d), st[d].w, st[d].s);
            if (kt + d + 1 < nst) store_a(buf ^ 1, st[(d + 1) % DEPTH]);
            __syncthreads();
        }
    }

    // C/D map of 32x32 MFMA: column = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    if (!part && (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
        // rounded tiles through the (now idle) A-tile LDS, 16-byte row chunks to global (ql_common.h)
        T* lds_wave = reinterpret_cast<T*>(smem[0]) + wave * 1024;
        static_assert(NW * 2048 <= 2 * BM * 128, "one 2 KB epilogue region per wave inside the A-tile buffers");
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                store_tile_32x32<T>(lds_wave, C, ldc, m0 + mt * 32, n_base - j + 32 * t, M, N, bias, lane,
                                    [&](int i) { return acc[mt][t][i]; });
        return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n_raw = n_base + 32 * t;
        if (n_raw >= N) continue;
        const T* bn = bias ? bias + n_raw : nullptr;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
                if (m >= M) continue;
                if (part) part[((int64_t)blockIdx.z * M + m) * N + n_raw] = acc[mt][t][i];   // fp32 slab; summed by splitk_reduce_kernel
                else store_out<T>(C + (int64_t)m * ldc + n_raw, acc[mt][t][i], bn);
            }
    }
}

template <typename T, int MT, int NT, int NW>
static int launch_gemm(const void* A, const void* tiled, const void* bias, void* C, int M, int N, int K, int64_t lda,
                       int64_t ldc, const GemmPlan& plan, float* ws, hipStream_t st) {
    const W4Layout L = w4_layout(N, K, sizeof(T));
    const int64_t G = L.G;
    const u32x4* Wt = (const u32x4*)tiled;                                 // tile-major part
    const T* Sp = (const T*)((const char*)tiled + (L.off_sm - L.off_wm));
    float* part = plan.ksplit > 1 ? ws : nullptr;
    constexpr int BN = NW * 32 * NT;
    const int nbx = (N + BN - 1) / BN, nby = (M + 32 * MT - 1) / (32 * MT);
    dim3 grid((unsigned)(nbx * nby), 1, (unsigned)plan.ksplit);
    w4_packed_gemm_kernel<T, MT, NT, NW, QL_GEMM_DEPTH><<<grid, NW * 64, 0, st>>>((const T*)A, Wt, Sp, M, N, K, (int)G, lda, plan.per,
        xcd_order(nbx, nby, (double)M * K * 2, (double)N * K * 0.5), (const T*)bias, (T*)C, ldc, part);
    const int rc = finish_launch(QL_K_W4_GEMM128);
    if (rc != 0 || !part) return rc;
    const int64_t total = (int64_t)M * N;
    splitk_reduce_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(part, (const T*)bias, (T*)C, M, N, ldc, plan.ksplit);
    return finish_launch(QL_K_SPLITK_REDUCE);
}


// Prefill-sized row counts go to the 256 x 256-tile kernel (w4_gemm256.hip), which pays in whole rounds of cu_count() workgroups.
// Since round 5 its persistent launch runs a last round that fills at most half the chip as half tiles by itself (K % 128 == 0, K >= 1024);
// what follows is the older rule for the shapes that launch does not take.
// When the last round would be less than 40 % full (qkv_proj at 8192 rows: 576 workgroups = 2.25 rounds at the price of 3), only
// the row tiles that fill the whole rounds go to it and the remaining rows to the 128-row-tile kernel as a second launch (same
// dequantised weights and fp32 sums; the two kernels add in different orders, so a row's last bit may depend on which one served
// it).  Returns the rows of the 256-tile launch: 0 = none, M = all, in between = peel.  QLINEAR_DISPATCH=nopeel: never peel.
int64_t w4_gemm256_rows(int64_t M, int64_t N, int64_t K, int64_t lda, const void* A, size_t esize) {
    if (!w4_gemm256_supported(M, N, K, lda, A, esize)) return 0;
    const int64_t nbx = (N + 255) / 256, nby = (M + 255) / 256, blocks = nbx * nby;
    if (w4_gemm256_tail_in_kernel(blocks, K)) return M;     // round 5: the persistent launch splits its own last round (half tiles, bit-equal)
    const int64_t cus = cu_count(), full = blocks / cus, tail = blocks - full * cus;
    const int64_t nby_main = full * cus / nbx, m_main = nby_main * 256;
    if ((dispatch_flags() & QL_D_NOPEEL) || full < 1 || tail == 0 || tail * 10 > cus * 4 || nby_main < 1 || m_main >= M) return M;
    return m_main;
}

template <typename T>
static int launch_gemm_any(const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N,
                           int64_t K, int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st, bool can_recurse = true) {
    if (can_recurse) {
        const int64_t m_main = w4_gemm256_rows(M, N, K, lda, A, sizeof(T));      // rows the 256 x 256-tile kernel takes (0: none)
        if (m_main >= M) return w4_gemm256(Act<T>::code, A, tiled, bias, C, M, N, K, lda, ldc, st);
        if (m_main > 0) {                                                          // "peel": the rest as a second launch below
            const int rc = w4_gemm256(Act<T>::code, A, tiled, bias, C, m_main, N, K, lda, ldc, st);
            if (rc != 0) return rc;
            return launch_gemm_any<T>((const T*)A + m_main * lda, tiled, bias, (T*)C + m_main * ldc, M - m_main, N, K, lda, ldc, ws, ws_bytes,
                                      st, false);
        }
    }
    const GemmPlan plan = gemm_plan(M, N, (K / 32 + 1) / 2, ws && ((uintptr_t)ws & 15) == 0 ? ws_bytes : 0);
    const int forced_nt = QL_TUNE("QLINEAR_GEMM_NT", 0);
    const int forced_nw = QL_TUNE("QLINEAR_GEMM_NW", 0);   // 4: two 4-wave blocks per CU (measurement)
    if (forced_nt == 2 && plan.mt == 4)       // 64 columns per wave: every A fragment feeds two MFMAs (experiment)
        return launch_gemm<T, 4, 2, 4>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, plan, (float*)ws, st);
    if (forced_nt == 2 && plan.mt == 2)
        return launch_gemm<T, 2, 2, 4>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, plan, (float*)ws, st);
    switch (plan.mt) {
    case 8:   // 256-row tiles, one wave per SIMD with the whole register file (QLINEAR_GEMM_MT=8: experiment)
        return launch_gemm<T, 8, 1, 4>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, plan, (float*)ws, st);
    case 4:
        // 8 waves (256 columns) per block halve the A-tile traffic per flop; worth it once that grid still fills the chip
        if (((N + 255) / 256) * ((M + 127) / 128) >= 256 && forced_nw != 4)
            return launch_gemm<T, 4, 1, 8>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, plan, (float*)ws, st);
        return launch_gemm<T, 4, 1, 4>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, plan, (float*)ws, st);
    case 2: return launch_gemm<T, 2, 1, 4>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, plan, (float*)ws, st);
    default: return launch_gemm<T, 1, 1, 4>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, plan, (float*)ws, st);
    }
}

size_t w4_packed_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) { return gemm_workspace_bytes(M, N, (K / 32 + 1) / 2); }

int w4_packed_gemm(int dtype, const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N,
                   int64_t K, int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_gemm_any<f16>(A, tiled, bias, C, M, N, K, lda, ldc, ws, ws_bytes, st);
    case QL_DTYPE_BF16: return launch_gemm_any<__bf16>(A, tiled, bias, C, M, N, K, lda, ldc, ws, ws_bytes, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

}  // namespace ql
