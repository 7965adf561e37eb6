// int4g32 GEMM for MANY activation rows (prefill: BASELINE config 5) on 256 x 256 output tiles, gfx950 - round 3.
//
//   C[M,N] = A[M,K] . dequant(W)      fp16 / bf16 activations, fp32 accumulation on v_mfma_f32_16x16x32_{f16,bf16} (round 3 - 4: 32x32x16)
//   (the reference's contraction, chatglm_q/int4/triton_ops.py:66-80: every weight dequantised to (n - 8) * s ROUNDED to the
//   activation dtype before the dot, :72-73)
//
// Why another kernel beside w4_gemm.hip (128 x 256 tiles, weights dequantised in registers straight into B fragments): there
// every weight is dequantised once per 128 rows and every MFMA needs a fresh 1 KB A fragment from LDS - 25 % of the SIMD cycles
// ran neither pipe and 19 % dequant VALU alone (profiles/r02_gemm_pmc.txt).  Here (MI355X guide, the 256 x 256
// template's geometry):
//   * block = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA tiles: 6 fragment reads feed 8 MFMAs (was 4 + a
//     13-instruction dequant for 4), 128 accumulator registers per lane, two waves per SIMD;
//   * the weights of a 64-deep K tile are dequantised ONCE per block: the block's 512 threads each take one 16-byte unit of the
//     tile-major layout (32 nibbles of one column and group), build its four 8-half B fragments and store them FRAGMENT-MAJOR
//     in LDS ([column tile][sub-step][lane][16 B]: every ds_write_b128 / ds_read_b128 is lane-linear, conflict-free) - half the
//     dequant work per flop of the 128-row kernel, and it is spread word by word behind the MFMAs of the tile before;
//   * the A tile (256 rows x 128 bytes) goes global -> LDS by LDS-DMA (no VGPR round trip, no ds_write pass), swizzled through
//     the SOURCE address (chunk c of row r at position 8 r + (c ^ ((r >> 1) & 7)): conflict-free 128-byte-pitch fragment reads);
//   * two LDS buffers per operand (128 KB), ONE block barrier per K tile (32 MFMAs per wave), every load hand-counted (vmq.h).
#include "launch.h"
#include "vmq.h"
#include "w4_dequant.h"
#include "w4_mma.h"

#ifndef QL_G256_PIN
#define QL_G256_PIN 1
#endif
#ifndef QL_G256_PRIO
#define QL_G256_PRIO 0
#endif
#ifndef QL_G256_SPREAD
#define QL_G256_SPREAD 1
#endif
#ifndef QL_G256_ABLATE
#define QL_G256_ABLATE 0
#endif
#ifndef QL_G256_ALT_PRIO
#define QL_G256_ALT_PRIO 0
#endif
#ifndef QL_G256_YOUNG_PRIO
#define QL_G256_YOUNG_PRIO 0
#endif

namespace ql {

// QL_G256_STAMPS (developer build, tools/ab/build_g256_variant.sh + tools/g256_timeline.py): lane 0 of waves 0 and 4 of every block sums
// the shader-clock cycles it spends in the two queue waits and the barrier of the K loop ([block][wave / 4][4] = W wait, A wait,
// barrier, whole loop).  s_memtime returns through lgkmcnt: the probes drain the LDS queue - read proportions, not absolutes.
#ifdef QL_G256_STAMPS
__device__ unsigned long long ql_g256_stamps[8192 * 2 * 4];
#define QL_G256_T() (unsigned long long)__builtin_amdgcn_s_memtime()
#define QL_G256_D() (QL_G256_STAMPS > 1 ? QL_G256_T() : 0ull)     // detailed probes (each drains the wave's LDS queue)
#endif

constexpr int kG256ABuf = 256 * 128;               // one A tile: 256 rows x 64 halves
constexpr int kG256BBuf = 8 * 4 * 64 * 16;         // one B tile: 8 column tiles x 4 sub-steps x 64 lanes x 16 bytes
constexpr int kG256Lds = 2 * kG256ABuf + 2 * kG256BBuf;

__device__ __forceinline__ void gload2(unsigned& dst, unsigned voff, unsigned long long base) {   // 16 bits, zero-extended
    asm volatile("s_nop 4\n\tglobal_load_ushort %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait_imm(i32x4& w, unsigned& s) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w), "+v"(s) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait_imm(i32x4& w, i32x4& w2) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w), "+v"(w2) : "n"(N) : "memory");
}

// int8 per-channel weights (W8 = true: the module layout's tile-major copy, qlinear_w8_tile): 8 bytes -> the 8 halves b * s ROUNDED
// to the activation dtype (chatglm_q/int8/triton_ops.py:62-73) in NATURAL k order - the A tile arrives by LDS-DMA and cannot be
// regrouped on the way like w8_gemm.hip's register-staged tile.  fp16: v_perm_b32 splices the bytes (b ^ 0x80) under 0x64 = 1152 + b
// pair by pair in k order (one instruction per pair, no shift), minus 1152 (exact), times s (ONE rounding).
template <typename T>
__device__ __forceinline__ u32x4 w8_dequant_natural(u32 w0, u32 w1, float s) {
    if constexpr (sizeof(T) == 2 && Act<T>::code == QL_DTYPE_F16) {
        const h2 k1152 = {(f16)1152.0f, (f16)1152.0f};
        const f16 sh = (f16)s;
        const h2 s2 = {sh, sh};
        const u32 t0 = w0 ^ 0x80808080u, t1 = w1 ^ 0x80808080u, k64 = 0x64646464u;
        const h2 e0 = (as_h2(__builtin_amdgcn_perm(k64, t0, 0x04010400u)) - k1152) * s2;      // (b0, b1)
        const h2 e1 = (as_h2(__builtin_amdgcn_perm(k64, t0, 0x04030402u)) - k1152) * s2;      // (b2, b3)
        const h2 e2 = (as_h2(__builtin_amdgcn_perm(k64, t1, 0x04010400u)) - k1152) * s2;
        const h2 e3 = (as_h2(__builtin_amdgcn_perm(k64, t1, 0x04030402u)) - k1152) * s2;
        return u32x4{as_u32(e0), as_u32(e1), as_u32(e2), as_u32(e3)};
    } else {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 w = i < 2 ? w0 : w1;
            const int sh = 16 * (i & 1);
            const float lo = (float)((int)(w << (24 - sh)) >> 24) * s;            // byte 2 i of the octet
            const float hi = (float)((int)(w << (16 - sh)) >> 24) * s;            // byte 2 i + 1
            const bf2 pr = {(__bf16)lo, (__bf16)hi};                              // one rounding each
            r[i] = __builtin_bit_cast(u32, pr);
        }
        return r;
    }
}

#ifdef QL_DEV_TUNING
// ---- round 3 / 4: the tile body on v_mfma_f32_32x32x16 - developer library only since round 5 (QLINEAR_G256_MI16=0: the A/B partner of the
// 16x16x32 body below, profiles/r05_g256_ab.txt, r05_w8_mi16_ab.txt); template parameters as for g256_tile_body16 -------------------------------
template <typename T, bool W8, bool GATE, int MT>
__device__ __forceinline__ void g256_tile_body(char* smem, const T* __restrict__ A, const u32x4* __restrict__ Wt, const T* __restrict__ Sp,
                                               int M, int N, int ksteps, int64_t lda, int tile_x, int m0,
                                               const T* __restrict__ bias, T* __restrict__ C, int64_t ldc,
                                               const T* __restrict__ resid, int64_t ldr) {
    static_assert(MT == 4 || MT == 2, "whole tile or half tile");
    typedef Mma<T> MM;
#ifdef QL_G256_STAMPS
    const unsigned long long t_block0 = QL_G256_T();
#endif
    int tid = threadIdx.x;
    if constexpr (MT != 4) asm volatile("" : "+v"(tid));   // behind the persistent kernel's K loops: an opaque thread id, so that nothing this body
                                                            // derives from it is computed early and kept in registers across those loops (hipcc did: spills)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int j = lane & 31, kb = lane >> 5;
    const int n0 = tile_x * 256;

    u32 k_mask_lo, k_mask_hi, k_magic;
    asm volatile("s_mov_b32 %0, 0x000F000F" : "=s"(k_mask_lo));
    asm volatile("s_mov_b32 %0, 0x00F000F0" : "=s"(k_mask_hi));
    asm volatile("v_mov_b32 %0, %1" : "=v"(k_magic) : "i"(MM::kMagic));

    // ---- this wave's share of the staging work -----------------------------------------------------------------------------------
    // weights: column tile `wave` of the block's 8, one 16-byte unit (+ its scale) per lane and K tile, 1 KB contiguous per wave
    const int ctiles = (N + 31) >> 5;
    const int ct_raw = tile_x * 8 + wave;
    const int ct = ct_raw < ctiles ? ct_raw : ctiles - 1;          // clamped: loads stay in bounds, stores are masked
    constexpr unsigned long long kWTile = W8 ? 2048ull : 1024ull;     // bytes of one column tile and K tile
    const unsigned long long w_base = sgpr64((unsigned long long)(uintptr_t)Wt + (unsigned long long)ct * (unsigned long long)ksteps * kWTile);
    const unsigned long long s_base = sgpr64((unsigned long long)(uintptr_t)Sp + (unsigned long long)ct * (unsigned long long)ksteps * (64ull * sizeof(T)));
    const float sc8 = W8 ? Act<T>::load(Sp + (32 * ct + j < N ? 32 * ct + j : N - 1)) : 0.f;     // the lane's output channel
    const unsigned w_voff = (unsigned)lane * 16u, s_voff = (unsigned)lane * (unsigned)sizeof(T);
    // activations: pieces MT wave .. MT wave + MT - 1 of the tile's 8 MT (1 KB = 8 rows each); lane -> (row, stored chunk position),
    // source chunk = position ^ swizzle(row)
    unsigned a_off[MT];
#pragma unroll
    for (int n = 0; n < MT; ++n) {
        const int q = 64 * (MT * wave + n) + lane, r = q >> 3, cp = q & 7;
        const int row = (m0 + r < M) ? (m0 + r) : (M - 1);
        a_off[n] = (unsigned)row * (unsigned)(lda * (int64_t)sizeof(T)) + (unsigned)((cp ^ ((r >> 1) & 7)) * 16);
    }
    const unsigned long long a_base = sgpr64((unsigned long long)(uintptr_t)A);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)smem);
    const unsigned a_dma = lds0 + (unsigned)(MT * wave) * 1024u;   // + buffer * kG256ABuf + n * 1024
    char* b_lds = smem + 2 * kG256ABuf;
    const int b_wr = ((wave * 4) * 64 + lane) * 16;                // + buffer * kG256BBuf + s * 1024
    // fragment read offsets (per sub-step s): A rows 32 MT wr + 32 mt + (lane & 31), chunk 4 kb + s; B column tiles 2 wc + nt
    int a_rd[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a_rd[s] = ((32 * MT * wr + j) * 8 + ((4 * kb + s) ^ ((j >> 1) & 7))) * 16;
    const int b_rd = ((2 * wc) * 4 * 64 + lane) * 16;              // + nt * 4096 + s * 1024

    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    i32x4 wq[2];                                       // packed weight unit of K tile (kt + 1), (kt + 2): two register sets
    std::conditional_t<W8, i32x4, unsigned> wsc[2];    // int4: the unit's scale; int8: the lane's second unit (half 1)
    auto issue_a = [&](int kt, int buf) {
#if QL_G256_ABLATE & 8                              // timing ablation: no A pieces in the loop (4 dummy loads keep the queue counts)
        if (kt > 1) {
            for (int n = 0; n < MT; ++n) asm volatile("s_nop 0" ::: "memory");
            return;
        }
#endif
        const int k = kt < ksteps ? kt : ksteps - 1;   // past the end: the last tile again (never read; keeps the queue counts fixed)
        const unsigned long long base = sgpr64(a_base + (unsigned long long)k * 128ull);
#pragma unroll
        for (int n = 0; n < MT; ++n) glds16(a_dma + (unsigned)(buf * kG256ABuf + n * 1024), a_off[n], base);
    };
    auto issue_w = [&](int kt, int set) {
        const int k = kt < ksteps ? kt : ksteps - 1;
        gload16(wq[set], w_voff, sgpr64(w_base + (unsigned long long)k * kWTile));
        if constexpr (W8) gload16(wsc[set], w_voff, sgpr64(w_base + (unsigned long long)k * kWTile + 1024ull));
        else gload2(wsc[set], s_voff, sgpr64(s_base + (unsigned long long)k * (64ull * sizeof(T))));
    };
    auto issue_w2 = [&](int k, int set) {              // second request of a K tile's weights: int8: half 1 of the lane's bytes; int4: the scale
        if constexpr (W8) gload16(wsc[set], w_voff, sgpr64(w_base + (unsigned long long)k * kWTile + 1024ull));
        else gload2(wsc[set], s_voff, sgpr64(s_base + (unsigned long long)k * (64ull * sizeof(T))));
    };
    typedef decltype(MM::scale_pair((const T*)nullptr, true)) scale_t;
    auto scale_of = [&](auto raw) {
        if constexpr (W8) return MM::scale_pair((const T*)nullptr, false);          // unused: the channel scale is sc8
        else {
            const uint16_t h = (uint16_t)raw;
            T sv;
            __builtin_memcpy(&sv, &h, 2);
            return MM::scale_pair(&sv, true);
        }
    };
    auto dequant_store = [&](int set, int buf, int s, scale_t sc) {     // word s of the unit -> B fragment (sub-step s) of the tile
#if QL_G256_ABLATE & 1                              // timing ablation (results wrong): no dequant arithmetic
        const u32x4 f = {(u32)wq[set][s], (u32)wq[set][s] ^ k_magic, (u32)wq[set][s], k_magic};
        (void)sc;
#else
        u32x4 f;
        if constexpr (W8) {                            // sub-step s = (half s >> 1, octet s & 1) of the lane's 32 bytes
            const i32x4& unit = (s >> 1) ? wsc[set] : wq[set];
            f = w8_dequant_natural<T>((u32)unit[2 * (s & 1)], (u32)unit[2 * (s & 1) + 1], sc8);
        } else {
            f = __builtin_bit_cast(u32x4, MM::dequant((u32)wq[set][s], k_mask_lo, k_mask_hi, k_magic, sc));
        }
#endif
#if QL_G256_ABLATE & 2                              // ... no B-fragment stores either
        if (f[0] == 0x12345678u && f[3] == 0x9abcdef0u)
#endif
        *reinterpret_cast<u32x4*>(b_lds + buf * kG256BBuf + b_wr + s * 1024) = f;
    };
    u32x4 fa[2][MT], fb[2][2];
    auto read_frags = [&](int buf, int s, u32x4 (&xa)[MT], u32x4 (&xb)[2]) {
#if QL_G256_ABLATE & 4                              // timing ablation: no fragment reads
        for (int nt = 0; nt < 2; ++nt) xb[nt] = u32x4{(u32)buf, (u32)s, 0x3c003c00u, (u32)nt};
        for (int mt = 0; mt < MT; ++mt) xa[mt] = u32x4{(u32)a_rd[s], (u32)buf, 0x3c003c00u, (u32)mt};
        return;
#endif
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) xb[nt] = *reinterpret_cast<const u32x4*>(b_lds + buf * kG256BBuf + b_rd + nt * 4096 + s * 1024);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xa[mt] = *reinterpret_cast<const u32x4*>(smem + buf * kG256ABuf + mt * 4096 + a_rd[s]);
    };

    // ---- prologue: A(0), W(0), W(1) requested; W(0) dequantised into B[0]; then the block every barrier is followed by ---------
    issue_a(0, 0);
    issue_w(0, 0);
    issue_w(1, 1);
    vm_wait_imm<2>(wq[0], wsc[0]);                     // A(0) and W(0) have landed (this wave's pieces)
    {
        const scale_t sc = scale_of(wsc[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) dequant_store(0, 0, s, sc);
    }
    __syncthreads();
    issue_a(1, 1);
    issue_w(2, 0);
    read_frags(0, 0, fa[0], fb[0]);

    // ---- K loop: one iteration = one 64-deep K tile = 32 MFMAs per wave, ONE block barrier - placed in front of the tile's LAST
    // sub-step, whose fragments are in registers by then: behind the barrier nobody reads the tile's buffers any more, so the
    // refill of them (A pieces of tile kt + 2, the unit of tile kt + 3) and the first fragment reads of tile kt + 1 go out at
    // once and their latency hides behind the 8 MFMAs of that last sub-step instead of idling the matrix pipe after every barrier
    // (first version, barrier at the end of the tile: 1 005 -> 1 048 TFLOP/s at 8192 x 4096 x 4096 with this skew).
    // Queue: behind each barrier MT A pieces, then unit + scale.  W(kt + 1) (requested behind the barrier of iteration kt - 2) has MT + 2
    // younger loads when iteration kt dequantises it; A(kt + 1) (behind the barrier of kt - 1) has 2 when iteration kt reaches its barrier.
    auto mma_sub = [&](int s) {
#if QL_G256_ALT_PRIO
        // the matrix pipe changes hands twice per K tile: group 0 is served first in sub-steps 3 and 0 (the two behind the barrier),
        // group 1 in sub-steps 1 and 2
        if (s == 3 || s == 1) {
            if ((s == 3) == (wave < 4)) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
#if QL_G256_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                acc[mt][nt] = MM::mma(__builtin_bit_cast(typename MM::frag, fa[s & 1][mt]),
                                      __builtin_bit_cast(typename MM::frag, fb[s & 1][nt]), acc[mt][nt]);
#if QL_G256_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
#if QL_G256_YOUNG_PRIO
    // the two waves of a SIMD (w, w + 4) share its matrix pipe; issue is arbitrated by priority, then AGE: at equal priority the
    // older wave ran ahead and parked ~1 000 cycles per K tile at the barrier while the younger one finished alone, its own
    // stalls uncovered (QL_G256_STAMPS=2).  One static raise for the younger half (MI355X_MICROARCH.md, two waves per SIMD, item 4).
    if (wave >= 4) __builtin_amdgcn_s_setprio(QL_G256_YOUNG_PRIO);
#endif
#ifdef QL_G256_STAMPS
    unsigned long long t_w = 0, t_a = 0, t_b = 0;
    const unsigned long long t_loop0 = QL_G256_T();
#endif
    auto k_tile = [&](int kt, auto curc) {
        constexpr int cur = decltype(curc)::value, nxt = cur ^ 1;
#ifdef QL_G256_STAMPS
        const unsigned long long t0 = QL_G256_D();
#endif
        vm_wait_imm<MT + 2>(wq[nxt], wsc[nxt]);        // W(kt + 1) has landed
#ifdef QL_G256_STAMPS
        t_w += QL_G256_D() - t0;
#endif
        const scale_t sc = scale_of(wsc[nxt]);
        static_for<3>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            read_frags(cur, s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
            mma_sub(s);
            dequant_store(nxt, nxt, s, sc);            // tile kt + 1's fragments 0..2 (3 with sub-step 2), behind these MFMAs
            if constexpr (s == 2) dequant_store(nxt, nxt, 3, sc);
            __builtin_amdgcn_sched_group_barrier(0x100, MT + 2, 0);        // the fragment reads of the next sub-step first
#pragma unroll
            for (int q = 0; q < 2 * MT; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // each MFMA with its share of the dequant VALU
                __builtin_amdgcn_sched_group_barrier(0x002, (s == 2 ? 4 : 2) * (4 / MT), 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x200, s == 2 ? 2 : 1, 0);
#if QL_G256_PIN
            __builtin_amdgcn_sched_barrier(0);         // sub-steps do not mix: left alone hipcc chains the MFMAs of one accumulator
#endif
        });
#ifdef QL_G256_STAMPS
        const unsigned long long t1 = QL_G256_D();
#endif
        vm_wait_imm<2>();                              // A(kt + 1) has landed
#ifdef QL_G256_STAMPS
        const unsigned long long t2 = QL_G256_D();
        t_a += t2 - t1;
#endif
        __syncthreads();                               // B(kt + 1), A(kt + 1) complete; fragments (kt, 3) are in registers
#ifdef QL_G256_STAMPS
        t_b += QL_G256_D() - t2;
#endif
#if QL_G256_SPREAD
        // behind the barrier both waves of a SIMD would issue their six requests (~100 issue cycles of M0 moves and wait states)
        // at the same time with the matrix pipe idle: the fragment reads go first, then one request behind each of the last
        // sub-step's MFMAs (whose fragments are in registers)
        read_frags(nxt, 0, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        {
            const int ka = kt + 2 < ksteps ? kt + 2 : ksteps - 1, kw = kt + 3 < ksteps ? kt + 3 : ksteps - 1;
            const unsigned long long abase_k = sgpr64(a_base + (unsigned long long)ka * 128ull);
            static_for<2 * MT>([&](auto qc) {
                constexpr int q = decltype(qc)::value, mt = q >> 1, nt = q & 1;
                acc[mt][nt] = MM::mma(__builtin_bit_cast(typename MM::frag, fa[1][mt]), __builtin_bit_cast(typename MM::frag, fb[1][nt]), acc[mt][nt]);
                if constexpr (q < MT) glds16(a_dma + (unsigned)(cur * kG256ABuf + q * 1024), a_off[q], abase_k);
                else if constexpr (q == MT) gload16(wq[nxt], w_voff, sgpr64(w_base + (unsigned long long)kw * kWTile));
                else if constexpr (q == MT + 1) issue_w2(kw, nxt);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
#else
        issue_a(kt + 2, cur);
        issue_w(kt + 3, nxt);                          // register set nxt held W(kt + 1): dequantised above
        read_frags(nxt, 0, fa[0], fb[0]);
        mma_sub(3);
#if QL_G256_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
#endif
    };
    int kt = 0;
    for (; kt + 1 < ksteps; kt += 2) {
        k_tile(kt, std::integral_constant<int, 0>{});
        k_tile(kt + 1, std::integral_constant<int, 1>{});
    }
    if (kt < ksteps) k_tile(kt, std::integral_constant<int, 0>{});
#ifdef QL_G256_STAMPS
    if (lane == 0 && (wave & 3) == 0 && blockIdx.x < 8192) {
        unsigned long long* o = ql_g256_stamps + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 4;
        o[0] = t_w; o[1] = t_a; o[2] = t_loop0 - t_block0; o[3] = QL_G256_T() - t_loop0;      // [2]: prologue (barrier time with STAMPS=2 is lost)
    }
    const unsigned long long t_loop1 = QL_G256_T();
#endif
    vm_wait_imm<0>(wq[0], wsc[0]);                     // the queue is empty before the registers / LDS are reused
    vm_wait_imm<0>(wq[1], wsc[1]);
    __syncthreads();                                   // ... and every wave is past its last fragment read

    // ---- epilogue: rounded 32 x 32 tiles through 2 KB of LDS per wave, 16-byte row chunks to global (ql_common.h) -------------
    const int mw = m0 + 32 * MT * wr, nw = n0 + 64 * wc;
    if (GATE || ((ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0)) {   // GATE: 8-byte chunks, alignment checked by the ABI
        T* lds_wave = reinterpret_cast<T*>(smem) + wave * 1024;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if constexpr (GATE)
                    store_tile_32x32_gated<T, 0, true>(lds_wave, C, ldc, mw + mt * 32, nw + 32 * nt, M, N, bias, lane, [&](int i) { return acc[mt][nt][i]; });
                else if (resid)                        // kernel-uniform: the residual stream is added in the row-chunk pass
                    store_tile_32x32_resid<T, 0, true>(lds_wave, C, ldc, resid, ldr, mw + mt * 32, nw + 32 * nt, M, N, bias, lane, [&](int i) { return acc[mt][nt][i]; });
                else
                    store_tile_32x32<T, 0, true>(lds_wave, C, ldc, mw + mt * 32, nw + 32 * nt, M, N, bias, lane, [&](int i) { return acc[mt][nt][i]; });
#ifdef QL_G256_STAMPS
        if (lane == 0 && (wave & 3) == 0 && blockIdx.x < 8192) ql_g256_stamps[((size_t)blockIdx.x * 2 + (wave >> 2)) * 4 + 1] = QL_G256_T() - t_loop1;
#endif
        return;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = nw + 32 * nt + j;
        if (n >= N) continue;
        const T* bn = bias ? bias + n : nullptr;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = mw + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
                if (m < M) store_out<T>(C + (int64_t)m * ldc + n, acc[mt][nt][i], bn);
            }
    }
}

template <typename T, bool W8 = false, bool GATE = false>
__global__ __launch_bounds__(512) void w4_gemm256_kernel(const T* __restrict__ A, const u32x4* __restrict__ Wt, const T* __restrict__ Sp,
                                                         int M, int N, int ksteps, int64_t lda, int nbx, int super_rows,
                                                         const T* __restrict__ bias, T* __restrict__ C, int64_t ldc,
                                                         const T* __restrict__ resid = nullptr, int64_t ldr = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // A[2] | B[2]; reused by the epilogue
    const TileXY tile = super_rows ? xcd_tile_super(blockIdx.x, gridDim.x, nbx, super_rows) : xcd_tile(blockIdx.x, gridDim.x, nbx);
    g256_tile_body<T, W8, GATE, 4>(smem, A, Wt, Sp, M, N, ksteps, lda, tile.x, tile.y * 256, bias, C, ldc, resid, ldr);
}

#endif   // QL_DEV_TUNING

// ---- round 5: the tile body on v_mfma_f32_16x16x32 (what the product runs, int4g32 and int8 per-channel weights) ---------------------------------
// W8 = false: int4g32, Wt / Sp = part 2 of the derived layout (units + scales, [column tile][K tile][lane]).
// W8 = true: int8 per channel, Wt = the tile-major copy ([column tile][K tile][half][lane][16 B]: two units per lane and K tile), Sp = S[n].
// GATE: the weight copy is gate-interleaved (a first MLP projection) and the epilogue applies SiLU * gate: C has N / 2 columns.
// MT = 4: the 256 x 256 block tile.  MT = 2: a HALF tile, 128 rows x 256 columns (8 waves as 2 x 4, wave tile 64 x 64) - the launch's last
// round: same K order per output element, so a half tile's outputs are bit-equal to the whole tile's.
// One output tile, rows m0 .. m0 + 64 MT - 1 x columns 256 tile_x ..: prologue, K loop, epilogue (block-wide; smem: kG256Lds bytes).
// Why: the GEMM sits at the board's power cap (profiles/r05_g256_power_cap.txt), MFMA-only loops sustain 12 - 17 % more on the 16x16 shapes than
// on 32x32 under that cap (1 KB of accumulators in and out per 16 K MACs instead of 4 KB per 32 K) and the vendor's dense kernel uses 16x16.
// Round 4 tried the shape in a ring of 32-deep stages only (a barrier per 32 k); here the SHIPPED loop keeps everything else: two LDS buffers per
// operand, one barrier per 64-deep K tile in front of its last quarter, the same LDS-DMA A image (its swizzle is conflict-free for the new
// fragment reads too), the same one-unit-per-thread dequant.  What changes:
//   * wave tile 128 x 64 = 8 x 4 tiles of 16 x 16 (128 accumulator registers as before); a K tile = 2 MFMA k-steps u (32 k = one int4 group);
//     lane (c = lane & 15, kq = lane >> 4) holds k = 32 u + 8 kq .. + 7: word kq of group u - the unit's words in natural order;
//   * B image: [16-column tile n16][step u][lane 16 kq + c]: the thread holding the unit of column j (of its wave's 32), group kb writes word s
//     to tile 2 wave + (j >> 4), step kb, lane 16 s + (j & 15) - 256-byte runs per 16 lanes, conflict-free;
//   * a K tile runs as four QUARTERS (u, h) of 16 MFMAs: m-tiles 4 h .. 4 h + 3 x the 4 n-tiles; per quarter 4 A fragment reads (+ the 4 B
//     fragments of the next step in front of h = 0) - 24 ds_read_b128 per K tile as before, 64 MFMAs of 16 cycles instead of 32 of 32.
// Output element K order differs from the 32x32 body (another contraction grouping): NOT bit-equal to it, same tolerance against the oracle.
// Measured (profiles/r05_g256_mi16_ab.txt, one tile per workgroup, 8192 rows, interleaved A/B, qkv / o / w_in / w_out TFLOP/s): 16x16x32 886 - 966 /
// 1 195 - 1 200 / 1 209 - 1 210 / 1 235 - 1 253 against 32x32x16 935 - 985 / 1 114 - 1 119 / 1 140 - 1 146 / 1 187 - 1 199: +7 / +6 / +4.5 % at the same 1 400 W.
// MT = 4: the 256 x 256 tile; MT = 2: a 128-row half tile (wave tile 64 x 64: the launch's last round) - same K order per output element: bit-equal
// to the whole tile.  (Quarter tiles - 64 rows, four units per left-over tile - were built and measured: qkv_proj 309 - 312 us against 300 - 320 with
// halves, profiles/r05_g256_ab.txt: no gain, removed.)
template <typename T> struct Mma16;
template <> struct Mma16<f16> {
    static __device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma16<__bf16> {
    static __device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

template <typename T, bool W8, bool GATE, int MT>
__device__ __forceinline__ void g256_tile_body16(char* smem, const T* __restrict__ A, const u32x4* __restrict__ Wt, const T* __restrict__ Sp,
                                                 int M, int N, int ksteps, int64_t lda, int tile_x, int m0,
                                                 const T* __restrict__ bias, T* __restrict__ C, int64_t ldc,
                                                 const T* __restrict__ resid, int64_t ldr) {
    static_assert(MT == 4 || MT == 2, "whole or half tile");
    typedef Mma<T> MM;
    typedef Mma16<T> M16;
    constexpr int MQ = MT;                             // 16-row tiles per quarter (a wave holds 2 MQ x 4 of them)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int j = lane & 31, kb = lane >> 5;           // staging: column j of the wave's column tile, group kb of the K tile
    const int c16 = lane & 15, kq = lane >> 4;         // MFMA: tile index, k quarter
    const int n0 = tile_x * 256;

    u32 k_mask_lo, k_mask_hi, k_magic;
    asm volatile("s_mov_b32 %0, 0x000F000F" : "=s"(k_mask_lo));
    asm volatile("s_mov_b32 %0, 0x00F000F0" : "=s"(k_mask_hi));
    asm volatile("v_mov_b32 %0, %1" : "=v"(k_magic) : "i"(MM::kMagic));

    const int ctiles = (N + 31) >> 5;
    const int ct_raw = tile_x * 8 + wave;
    const int ct = ct_raw < ctiles ? ct_raw : ctiles - 1;
    // W8 (int8 per channel, the tile-major copy of qlinear_w8_tile: two 16-byte units per lane and K tile = the channel's 32 consecutive k of group
    // kb, in natural order like the int4 unit's words): Sp = S[n], the lane's channel scale; the second unit takes the scale's place in the queue
    constexpr unsigned long long kWTile = W8 ? 2048ull : 1024ull;
    const unsigned long long w_base = sgpr64((unsigned long long)(uintptr_t)Wt + (unsigned long long)ct * (unsigned long long)ksteps * kWTile);
    const unsigned long long s_base = sgpr64((unsigned long long)(uintptr_t)Sp + (unsigned long long)ct * (unsigned long long)ksteps * (64ull * sizeof(T)));
    const float sc8 = W8 ? Act<T>::load(Sp + (32 * ct + j < N ? 32 * ct + j : N - 1)) : 0.f;
    const unsigned w_voff = (unsigned)lane * 16u, s_voff = (unsigned)lane * (unsigned)sizeof(T);
    unsigned a_off[MT];
#pragma unroll
    for (int n = 0; n < MT; ++n) {
        const int q = 64 * (MT * wave + n) + lane, r = q >> 3, cp = q & 7;
        const int row = (m0 + r < M) ? (m0 + r) : (M - 1);
        a_off[n] = (unsigned)row * (unsigned)(lda * (int64_t)sizeof(T)) + (unsigned)((cp ^ ((r >> 1) & 7)) * 16);
    }
    const unsigned long long a_base = sgpr64((unsigned long long)(uintptr_t)A);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)smem);
    const unsigned a_dma = lds0 + (unsigned)(MT * wave) * 1024u;
    char* b_lds = smem + 2 * kG256ABuf;
    const int b_wr = (((2 * wave + (j >> 4)) * 2 + kb) * 64 + (j & 15)) * 16;      // + buffer * kG256BBuf + s * 256 (word s -> lanes 16 s ..)
    int a_rd[2];                                       // per step u: row 32 MT wr + c16 (+ 16 per m-tile), chunk 4 u + kq
#pragma unroll
    for (int u = 0; u < 2; ++u) a_rd[u] = ((32 * MT * wr + c16) * 8 + ((4 * u + kq) ^ ((c16 >> 1) & 7))) * 16;
    const int b_rd = ((4 * wc) * 2 * 64 + lane) * 16;  // + nt * 2048 + u * 1024

    f32x4 acc[2 * MQ][4];
#pragma unroll
    for (int mt = 0; mt < 2 * MQ; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    i32x4 wq[2];
    std::conditional_t<W8, i32x4, unsigned> wsc[2];    // int4: the unit's scale; int8: the lane's second unit
    auto issue_a = [&](int kt, int buf) {
        const int k = kt < ksteps ? kt : ksteps - 1;
        const unsigned long long base = sgpr64(a_base + (unsigned long long)k * 128ull);
#pragma unroll
        for (int n = 0; n < MT; ++n) glds16(a_dma + (unsigned)(buf * kG256ABuf + n * 1024), a_off[n], base);
    };
    auto issue_w = [&](int kt, int set) {
        const int k = kt < ksteps ? kt : ksteps - 1;
        gload16(wq[set], w_voff, sgpr64(w_base + (unsigned long long)k * kWTile));
        if constexpr (W8) gload16(wsc[set], w_voff, sgpr64(w_base + (unsigned long long)k * kWTile + 1024ull));
        else gload2(wsc[set], s_voff, sgpr64(s_base + (unsigned long long)k * (64ull * sizeof(T))));
    };
    typedef decltype(MM::scale_pair((const T*)nullptr, true)) scale_t;
    auto scale_of = [&](auto raw) {
        if constexpr (W8) return MM::scale_pair((const T*)nullptr, false);          // unused: the channel scale is sc8
        else {
            const uint16_t h = (uint16_t)raw;
            T sv;
            __builtin_memcpy(&sv, &h, 2);
            return MM::scale_pair(&sv, true);
        }
    };
    auto dequant_store = [&](int set, int buf, int s, scale_t sc) {     // word s (8 k) of the lane's 32 -> lanes 16 s .. of the step's B image
        u32x4 f;
        if constexpr (W8) {
            const i32x4& unit = (s >> 1) ? wsc[set] : wq[set];
            f = w8_dequant_natural<T>((u32)unit[2 * (s & 1)], (u32)unit[2 * (s & 1) + 1], sc8);
        } else {
            f = __builtin_bit_cast(u32x4, MM::dequant((u32)wq[set][s], k_mask_lo, k_mask_hi, k_magic, sc));
        }
        *reinterpret_cast<u32x4*>(b_lds + buf * kG256BBuf + b_wr + s * 256) = f;
    };
    // fragments: A double-buffered per quarter; B ONE set per step, refilled in place - the second quarter of a step runs its MFMAs n-tile by
    // n-tile and requests the next step's fragment of an n-tile right behind that n-tile's last MFMA (16 registers instead of 32: with two sets
    // the persistent kernel spilled)
    u32x4 fa[2][MQ], fb[4];
    auto read_a = [&](int buf, int qd, u32x4 (&xa)[MQ]) {
        const int u = qd >> 1, h = qd & 1;
#pragma unroll
        for (int i = 0; i < MQ; ++i) xa[i] = *reinterpret_cast<const u32x4*>(smem + buf * kG256ABuf + (MQ * h + i) * 2048 + a_rd[u]);
    };
    auto read_b1 = [&](int buf, int u, int nt) {
        return *reinterpret_cast<const u32x4*>(b_lds + buf * kG256BBuf + b_rd + nt * 2048 + u * 1024);
    };

    issue_a(0, 0);
    issue_w(0, 0);
    issue_w(1, 1);
    vm_wait_imm<2>(wq[0], wsc[0]);
    {
        const scale_t sc = scale_of(wsc[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) dequant_store(0, 0, s, sc);
    }
    __syncthreads();
    issue_a(1, 1);
    issue_w(2, 0);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) fb[nt] = read_b1(0, 0, nt);
    read_a(0, 0, fa[0]);

    auto k_tile = [&](int kt, auto curc) {
        constexpr int cur = decltype(curc)::value, nxt = cur ^ 1;
        vm_wait_imm<MT + 2>(wq[nxt], wsc[nxt]);        // W(kt + 1) has landed
        const scale_t sc = scale_of(wsc[nxt]);
        // quarters (u, 0): m-tile by m-tile on the step's B fragments; quarters (u, 1): n-tile by n-tile, each n-tile's fragment of the NEXT
        // step requested behind its last MFMA
        static_for<2>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            read_a(cur, 2 * u + 1, fa[1]);
#pragma unroll
            for (int i = 0; i < MQ; ++i)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[i][nt] = M16::mma(fa[0][i], fb[nt], acc[i][nt]);
            dequant_store(nxt, nxt, 2 * u, sc);
            if constexpr (u == 1) dequant_store(nxt, nxt, 3, sc);
            __builtin_amdgcn_sched_group_barrier(0x100, MQ, 0);
#pragma unroll
            for (int q = 0; q < 4 * MQ; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (u == 1 ? 2 : 1) * (4 / MQ), 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x200, u == 1 ? 2 : 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (u == 0) {
                read_a(cur, 2, fa[0]);
                static_for<4>([&](auto ntc) {
                    constexpr int nt = decltype(ntc)::value;
#pragma unroll
                    for (int i = 0; i < MQ; ++i) acc[MQ + i][nt] = M16::mma(fa[1][i], fb[nt], acc[MQ + i][nt]);
                    fb[nt] = read_b1(cur, 1, nt);
                    if constexpr (nt == 1) dequant_store(nxt, nxt, 1, sc);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        });
        vm_wait_imm<2>();                              // A(kt + 1) has landed
        __syncthreads();
        read_a(nxt, 0, fa[0]);
        __builtin_amdgcn_sched_barrier(0);
        {
            const int ka = kt + 2 < ksteps ? kt + 2 : ksteps - 1, kw = kt + 3 < ksteps ? kt + 3 : ksteps - 1;
            const unsigned long long abase_k = sgpr64(a_base + (unsigned long long)ka * 128ull);
            static_for<4>([&](auto ntc) {              // quarter (1, 1): n-tile by n-tile; behind each: the next K tile's B fragment, requests
                constexpr int nt = decltype(ntc)::value;
#pragma unroll
                for (int i = 0; i < MQ; ++i) acc[MQ + i][nt] = M16::mma(fa[1][i], fb[nt], acc[MQ + i][nt]);
                fb[nt] = read_b1(nxt, 0, nt);
                if constexpr (MT == 4) {
                    if constexpr (nt < 2) {
                        glds16(a_dma + (unsigned)(cur * kG256ABuf + (2 * nt) * 1024), a_off[2 * nt], abase_k);
                        glds16(a_dma + (unsigned)(cur * kG256ABuf + (2 * nt + 1) * 1024), a_off[2 * nt + 1], abase_k);
                    }
                } else {
                    if constexpr (nt < 2) glds16(a_dma + (unsigned)(cur * kG256ABuf + nt * 1024), a_off[nt], abase_k);
                }
                if constexpr (nt == 2) gload16(wq[nxt], w_voff, sgpr64(w_base + (unsigned long long)kw * kWTile));
                if constexpr (nt == 3) {
                    if constexpr (W8) gload16(wsc[nxt], w_voff, sgpr64(w_base + (unsigned long long)kw * kWTile + 1024ull));
                    else gload2(wsc[nxt], s_voff, sgpr64(s_base + (unsigned long long)kw * (64ull * sizeof(T))));
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    };
    int kt = 0;
    for (; kt + 1 < ksteps; kt += 2) {
        k_tile(kt, std::integral_constant<int, 0>{});
        k_tile(kt + 1, std::integral_constant<int, 1>{});
    }
    if (kt < ksteps) k_tile(kt, std::integral_constant<int, 0>{});
    vm_wait_imm<0>(wq[0], wsc[0]);
    vm_wait_imm<0>(wq[1], wsc[1]);
    __syncthreads();

    // ---- epilogue: 32 x 32 blocks of 2 x 2 tiles through the wave's 2 KB of LDS (ql_common.h, LAYOUT 1), 16-byte row chunks ----
    const int mw = m0 + 32 * MT * wr, nw = n0 + 64 * wc;
    if (!GATE && !((ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0)) {
        // rows of C that are not 16-byte aligned (ldc % 8 != 0): element by element from the accumulator layout - tile (mt, nt) register r is
        // row 16 mt + 4 kq + r, column 16 nt + c16 (no residual here: that entry point requires aligned rows)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = nw + 16 * nt + c16;
            if (n >= N) continue;
            const T* bn = bias ? bias + n : nullptr;
#pragma unroll
            for (int mt = 0; mt < 2 * MQ; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mw + 16 * mt + 4 * kq + r;
                    if (m < M) store_out<T>(C + (int64_t)m * ldc + n, acc[mt][nt][r], bn);
                }
        }
        return;
    }
    T* lds_wave = reinterpret_cast<T*>(smem) + wave * 1024;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < MT; ++mb) {
            auto val = [&](int i) { return acc[2 * mb + (i >> 3)][2 * nb + ((i >> 2) & 1)][i & 3]; };
            if constexpr (GATE)
                store_tile_32x32_gated<T, 1, true>(lds_wave, C, ldc, mw + mb * 32, nw + 32 * nb, M, N, bias, lane, val);
            else if (resid)
                store_tile_32x32_resid<T, 1, true>(lds_wave, C, ldc, resid, ldr, mw + mb * 32, nw + 32 * nb, M, N, bias, lane, val);
            else
                store_tile_32x32<T, 1, true>(lds_wave, C, ldc, mw + mb * 32, nw + 32 * nb, M, N, bias, lane, val);
        }
}

// The launch (int4g32 and int8 weights).  Workgroups 0 .. whole - 1 compute the tiles of that id in the XCD-aware order (`total` tiles in all).  The tiles
// whole .. total - 1 - what is left over after the whole rounds of one workgroup per CU - run as HALF tiles on the workgroups behind:
// 128 rows x 256 columns each, the same K order per output element (bit-equal to a whole tile), so that the last round keeps twice as
// many CUs busy for half a tile time instead of a ragged round (or a second launch on the 128-row-tile kernel: the older peel).  Unit
// u = id - whole: tile whole + 8 (u / 16) + u % 8 - its id keeps the workgroup's XCD (id % 8, whole % 8 == 0) in the tile order - half
// (u / 8) % 2.  The hardware hands workgroups out in id order as CUs free up: the half tiles are the launch's last round by themselves.
// History (LABNOTES 4d): a PERSISTENT form on the 32x32x16 body (one workgroup per CU walking its tiles as ONE stream of K tiles: no
// prologue after the first tile, epilogue stores draining behind the next tile's MFMAs) measured +3 - 4 % on o_proj / w_in at the board's
// power cap; the 16x16x32 body is worth +5 - 7 % by itself and needs 16 more fragment registers - around it the stream's two-tile state
// no longer fitted 256 registers (hipcc spilled accumulators and registers with loads in flight), and what was left of the persistent
// launch without the stream - the half-tile last round - needs no persistence.
template <typename T, bool W8 = false, bool GATE = false>
__global__ __launch_bounds__(512) void w4_gemm256x16_kernel(const T* __restrict__ A, const u32x4* __restrict__ Wt, const T* __restrict__ Sp,
                                                            int M, int N, int ksteps, int64_t lda, int nbx, int super_rows, int total, int whole,
                                                            const T* __restrict__ bias, T* __restrict__ C, int64_t ldc,
                                                            const T* __restrict__ resid = nullptr, int64_t ldr = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < whole) {
        const TileXY tile = super_rows ? xcd_tile_super(blockIdx.x, (unsigned)total, nbx, super_rows) : xcd_tile(blockIdx.x, (unsigned)total, nbx);
        g256_tile_body16<T, W8, GATE, 4>(smem, A, Wt, Sp, M, N, ksteps, lda, tile.x, tile.y * 256, bias, C, ldc, resid, ldr);
        return;
    }
    const unsigned u = blockIdx.x - (unsigned)whole;
    const unsigned t = (unsigned)whole + (((u >> 4) << 3) | (u & 7u));
    if (t >= (unsigned)total) return;
    const TileXY tile = super_rows ? xcd_tile_super(t, (unsigned)total, nbx, super_rows) : xcd_tile(t, (unsigned)total, nbx);
    g256_tile_body16<T, W8, GATE, 2>(smem, A, Wt, Sp, M, N, ksteps, lda, tile.x, tile.y * 256 + 128 * (int)((u >> 3) & 1u), bias, C, ldc, resid, ldr);
}

// ---- round 4 EXPERIMENT (developer library only, -DQL_DEV_TUNING; QLINEAR_G256_RING=1 selects it): the structure that took the int8 x
// int8 kernel from 1.81 to 2.07 POP/s (w8a8_gemm256.hip) applied to the weight-only GEMM.  Measured at 8192 rows, qkv / o / w_in / w_out,
// TFLOP/s (tools/gemm_yardstick.py, profiles/r04_gemm_power.txt): kernel above 987 / 1 145 / 1 147 / 1 185; ring, 8 waves, jobs between the
// MFMAs 900 / 1 117 / 1 103 / 1 171; ring, 4 waves 814 / 1 015 / 1 057 / 1 100; ring, 8 waves, ping-pong phases 818 / 976 / 1 036 / 1 094;
// younger-wave priority 1 / 2: within the run-to-run spread.  Four loop structures, two MFMA shapes, one result: this GEMM is bound by
// the power budget at the energy per flop its dequant + LDS round trip cost, not by how its instructions are arranged.  Not shipped.
#ifdef QL_DEV_TUNING
// The same GEMM as a RING OF FOUR 32-deep K stages on v_mfma_f32_16x16x32_{f16,bf16}, one wave per SIMD -----------------------
// Why (profiles/r04_gemm_power.txt): these GEMMs are POWER bound (the loop above idles ~20 % of its cycles at ~1.4 GHz), MFMA-only loops
// on random operands sustain 1.92 - 1.99 PFLOP/s with the 16x16x32 shape against 1.71 - 1.78 with 32x32x16 under the same cap (a 32x32 MFMA
// moves 4 KB of accumulators in and out per 32 K flops, a 16x16 one 1 KB per 16 K), the vendor's dense f16 kernel on this chip is
// MT256x256x64 / MI16x16 / 4 waves - and the int8 x int8 kernel restructured this way went 1.81 -> 2.07 POP/s (w8a8_gemm256.hip).
//   * 4 waves as 2 x 2, wave tile 128 x 128 = 8 x 8 tiles of 16 x 16, 256 accumulator registers pinned to the AGPRs;
//   * stage = 32 k = ONE int4 group = one MFMA k-step: A 256 rows x 64 B by LDS-DMA (swizzle in the source address, chunk c of row r at
//     position c ^ (2 (r >> 3 & 1)): conflict-free for the four non-contiguous 16-lane groups of a ds_read_b128), B 256 columns x 32 k dequantised ONCE per block - each of the 256 threads owns one unit (column,
//     group) per stage, builds its four 8-half fragments dword by dword BETWEEN the MFMAs and stores them fragment-major
//     ([16-column tile][lane 16 q + c][16 B]); 4 stages x (16 + 16) KB = 128 KB of LDS;
//   * step t: barrier (A(t + 1), B(t + 1) visible; everybody's fragments of stage t are in registers), 64 MFMAs on stage t with, in
//     between: the 16 fragment reads of stage t + 1, the dequant of stage t + 2 (its unit was requested two steps ago), the requests
//     of stage t + 4 (2 weight loads into register set t & 3, 4 A pieces into the buffer stage t just left).  vmcnt(10) at the barrier;
//   * the WEIGHT fragment is the MFMA's first operand: D[i][j] has i = output column, so a lane holds FOUR CONSECUTIVE COLUMNS of one
//     row per tile and the epilogue stores 8-byte row pieces straight from registers (SiLU * gate: the lane's four columns are one
//     (h, h, gate, gate) quad -> one 4-byte store; residual: one 8-byte load) - no LDS transposition.
#ifdef QL_R4_STAMPS                                  // developer build (tools/r4_timeline.py): wave 0 of every block stamps its phases
__device__ unsigned long long ql_r4_stamps[16384 * 12];
#define QL_R4_STAMP(i) do { if (stamp_on) { st_rt[i] = __builtin_amdgcn_s_memrealtime(); st_ck[i] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define QL_R4_STAMP(i) do { } while (0)
#endif
constexpr int kR4Stage = 32768;                    // A 16 KB | B 16 KB
constexpr int kR4Lds = 4 * kR4Stage;
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ void mfma16_acc(f32x4v& acc, const u32x4& w, const u32x4& a) {
    // accumulators pinned to the accumulation registers ("+a": left to itself hipcc kept part of the 64 tiles in VGPRs and moved
    // them around every step, with spills); operands come from ds_read_b128 (hipcc places the lgkmcnt waits for "v" inputs itself)
    if constexpr (Act<T>::code == QL_DTYPE_F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}

__device__ __forceinline__ void gload8(u32x2& dst, unsigned voff, unsigned long long base) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait_imm(u32x2& w, unsigned& s) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w), "+v"(s) : "n"(N) : "memory");
}

// NW = 4: 2 x 2 waves, wave tile 128 x 128 (one wave per SIMD).  NW = 8: 2 x 4 waves, wave tile 128 x 64, two waves per SIMD - from ONE
// wave a 16x16 MFMA issues every 20 cycles instead of 16 (tools/microbench/mfma_i8_rate.hip) and the dequant VALU has nothing to hide
// behind; two waves alternate on the pipe at the full rate.  With 8 waves a thread dequantises HALF a unit (fragments 2 h, 2 h + 1,
// h = wave >> 2) and the 128 + 128 register budget holds the next stage's A fragments one row behind the current one's.
// PP (8 waves): PING-PONG.  The two waves of a SIMD (w, w + 4: the wave tile rows 0..127 / 128..255) run half a step apart - a step is
// [barrier, M: the 32 MFMAs back to back] [barrier, J: fragment reads of the next stage, dequant, requests], and group 1 passes one
// extra barrier first, so while one wave of a SIMD issues its MFMAs (one per 16 cycles: half the pipe) the other one's LDS / VALU /
// vector-memory instructions go to the other units.  Measured reason (tools/r4_timeline.py): with both waves in the same phase and
// their jobs spread between their MFMAs the older wave of each SIMD ran ahead and waited 450 - 530 of every 1 480 - 1 570 cycles at the
// step's barrier while the younger one finished alone.  M ends before J starts, so ONE set of fragment registers serves both.
template <typename T, bool W8, bool GATE, int NW, bool PP>
__global__ __launch_bounds__(NW * 64) void w4_gemm256_r4_kernel(const T* __restrict__ A, const u32x4* __restrict__ Wt, const T* __restrict__ Sp,
                                                                int M, int N, int steps, int wsteps, int64_t lda, int nbx, int super_rows,
                                                                const T* __restrict__ bias, T* __restrict__ C, int64_t ldc,
                                                                const T* __restrict__ resid = nullptr, int64_t ldr = 0) {
    static_assert(NW == 4 || NW == 8, "4 waves (2 x 2) or 8 waves (2 x 4)");
    static_assert(!PP || NW == 8, "ping-pong needs two waves per SIMD");
    typedef Mma<T> MM;
    constexpr int NWN = NW / 2;                        // waves side by side in N
    constexpr int NT = 16 / NWN;                       // 16-column tiles per wave
    constexpr int NAL = 16 / NW;                       // A pieces a wave stages per step
    constexpr int NFR = 16 / NW;                       // fragments (of the unit's 4) a thread dequantises: 4 or 2
    constexpr int NWL = (W8 && NW == 8) ? 1 : 2;       // weight loads per step and thread
    constexpr int S = 8 * NT;                          // MFMAs per wave and step
    extern __shared__ __attribute__((aligned(16))) char smem[];   // stage[4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef QL_R4_STAMPS
    unsigned long long st_rt[4], st_ck[4];
    const bool stamp_on = wave == 0;
    QL_R4_STAMP(0);
#endif
    const int wr = wave / NWN, wc = wave % NWN;
    const int c16 = lane & 15, kq = lane >> 4;
    const TileXY tile = super_rows ? xcd_tile_super(blockIdx.x, gridDim.x, nbx, super_rows) : xcd_tile(blockIdx.x, gridDim.x, nbx);
    const int m0 = tile.y * 256, n0 = tile.x * 256;

    u32 k_mask_lo, k_mask_hi, k_magic;
    asm volatile("s_mov_b32 %0, 0x000F000F" : "=s"(k_mask_lo));
    asm volatile("s_mov_b32 %0, 0x00F000F0" : "=s"(k_mask_hi));
    asm volatile("v_mov_b32 %0, %1" : "=v"(k_magic) : "i"(MM::kMagic));

    // ---- staging share: A pieces NAL wave .. (16 rows x 64 B each); weights: unit of column 64 (wave & 3) + lane (column tile
    // 2 (wave & 3) + (lane >> 5) of the block's 8, j = lane & 31), group t; with 8 waves: its half h = wave >> 2 ----------------------------
    unsigned a_off[NAL];
#pragma unroll
    for (int n = 0; n < NAL; ++n) {
        const int r = 16 * (NAL * wave + n) + (lane >> 2), cp = lane & 3;
        const int row = (m0 + r < M) ? (m0 + r) : (M - 1);
        a_off[n] = (unsigned)row * (unsigned)(lda * (int64_t)sizeof(T)) + (unsigned)((cp ^ (((r >> 3) & 1) << 1)) * 16);
    }
    const unsigned long long a_base = sgpr64((unsigned long long)(uintptr_t)A);
    const int ctiles = (N + 31) >> 5;
    const int wu = wave & 3, hh = NW == 8 ? wave >> 2 : 0;
    const int ct0_raw = tile.x * 8 + 2 * wu;
    const int ct0 = ct0_raw < ctiles ? ct0_raw : ctiles - 1;       // clamped: loads stay in bounds, stores are masked
    const int ct1 = ct0_raw + 1 < ctiles ? ct0_raw + 1 : ctiles - 1;
    const int jw = lane & 31, ctl = lane >> 5;
    constexpr unsigned long long kWStep = W8 ? 2048ull : 1024ull;     // bytes of one column tile and 64-deep K step of the tile-major copy
    const unsigned long long w_base = sgpr64((unsigned long long)(uintptr_t)Wt + (unsigned long long)ct0 * (unsigned long long)wsteps * kWStep +
                                             (unsigned long long)hh * (W8 ? 1024ull : 8ull));
    const unsigned long long s_base = sgpr64((unsigned long long)(uintptr_t)Sp + (unsigned long long)ct0 * (unsigned long long)wsteps * (64ull * sizeof(T)));
    const unsigned ct_d = ctl ? (unsigned)(ct1 - ct0) : 0u;           // 0 or 1 column tiles past ct0
    const unsigned w_voff = ct_d * (unsigned)wsteps * (unsigned)kWStep + (unsigned)jw * 16u;
    const unsigned s_voff = (ct_d * (unsigned)wsteps * 64u + (unsigned)jw) * (unsigned)sizeof(T);
    const float sc8 = W8 ? Act<T>::load(Sp + (32 * (ctl ? ct1 : ct0) + jw < N ? 32 * (ctl ? ct1 : ct0) + jw : N - 1)) : 0.f;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)smem);
    const unsigned a_dma = lds0 + (unsigned)(NAL * wave) * 1024u;                // + stage * kR4Stage + n * 1024
    // MFMA tile nt of the block (16 per block) does NOT cover 16 consecutive columns: its index i is column 32 (nt >> 1) + 8 (i >> 2) +
    // 4 (nt & 1) + (i & 3), so that the lane holding D rows 4 q .. 4 q + 3 of tiles 2 p and 2 p + 1 holds EIGHT CONSECUTIVE columns
    // 32 p + 8 q + 0..7 of its row: one 16-byte store, 64 contiguous bytes per row and instruction.  The B image is [tile][16 q + (i ^
    // 8 (nt & 1))]: fragment reads are lane-linear up to that flip, which keeps the two tiles a 16-column run writes to on different banks.
    // This thread's column 32 pc + jw -> tile 2 pc + ((jw >> 2) & 1), index 4 (jw >> 3) + (jw & 3).
    const int pcw = 2 * wu + ctl, tb = (jw >> 2) & 1;
    char* b_wr = smem + 16384 + ((2 * pcw + tb) * 64 + ((4 * (jw >> 3) + (jw & 3)) ^ (8 * tb))) * 16 + (NFR == 2 ? hh * 512 : 0);   // + stage * kR4Stage + i * 256
    const int a_rd0 = (128 * wr + c16) * 64 + ((kq ^ (((c16 >> 3) & 1) << 1)) * 16);    // + mt * 1024
    const int b_rd0 = 16384 + (NT * wc) * 1024 + (16 * kq + c16) * 16, b_rd1 = 16384 + (NT * wc) * 1024 + (16 * kq + (c16 ^ 8)) * 16;   // even / odd tiles, + nt * 1024
    const char* a_rd[2] = {smem + a_rd0, smem + a_rd0 + 2 * kR4Stage};
    const char* b_rd[2][2] = {{smem + b_rd0, smem + b_rd0 + 2 * kR4Stage}, {smem + b_rd1, smem + b_rd1 + 2 * kR4Stage}};

    f32x4v acc[8][NT];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // the thread's share of the unit of stage s in register set s & 3.  4 waves: int4 the 16-byte unit + its scale, int8 bytes 0..15 +
    // bytes 16..31.  8 waves: int4 words 2 h, 2 h + 1 + the scale, int8 bytes 16 h .. 16 h + 15.
    std::conditional_t<(NFR == 2 && !W8), u32x2, i32x4> wq[4];
    std::conditional_t<W8, i32x4, unsigned> wsc[4];
    auto issue_a = [&](int t, int buf, int n) {
        const int k = t < steps ? t : steps - 1;       // past the end: the last stage again (never read; keeps the queue counts fixed)
        glds16(a_dma + (unsigned)(buf * kR4Stage + n * 1024), a_off[n], sgpr64(a_base + (unsigned long long)k * 64ull));
    };
    auto issue_w = [&](int t, int set, int part) {     // stage t = half (t & 1) of 64-deep step t >> 1 of the tile-major copy
        const int k = t < steps ? t : steps - 1;
        if constexpr (W8) {                            // [step][half h][lane 32 kb + j]: bytes 16 h + 0..15 of the stage's 32
            const unsigned long long b = sgpr64(w_base + (unsigned long long)(k >> 1) * 2048ull + (unsigned long long)(k & 1) * 512ull + (unsigned long long)part * 1024ull);
            if (part == 0) gload16(wq[set], w_voff, b);
            else gload16(wsc[set], w_voff, b);
        } else {
            const unsigned long long b = sgpr64(w_base + (unsigned long long)(k >> 1) * 1024ull + (unsigned long long)(k & 1) * 512ull);
            if (part == 0) {
                if constexpr (NFR == 2) gload8(wq[set], w_voff, b);
                else gload16(wq[set], w_voff, b);
            } else gload2(wsc[set], s_voff, sgpr64(s_base + (unsigned long long)((k >> 1) * 64 + (k & 1) * 32) * sizeof(T)));
        }
    };
    typedef decltype(MM::scale_pair((const T*)nullptr, true)) scale_t;
    auto scale_of = [&](auto raw) {
        if constexpr (W8) return MM::scale_pair((const T*)nullptr, false);          // unused: the channel scale is sc8
        else {
            const uint16_t h = (uint16_t)raw;
            T sv;
            __builtin_memcpy(&sv, &h, 2);
            return MM::scale_pair(&sv, true);
        }
    };
    // dword d (0..3) of the thread's i-th fragment (fragment NFR hh + i of the unit) from register set `set`
    auto dequant_dword = [&](int set, int i, int d, scale_t sc) -> u32 {
        if constexpr (W8) {
            const i32x4& unit = (NFR == 4 && (i >> 1)) ? wsc[set] : wq[set];
            const u32x4 f = w8_dequant_natural<T>((u32)unit[2 * (i & 1)], (u32)unit[2 * (i & 1) + 1], sc8);   // hipcc keeps the dword asked for
            return f[d];
        } else {
            return MM::dequant_part((u32)wq[set][i], d, k_mask_lo, k_mask_hi, k_magic, sc);
        }
    };
    auto wait_w = [&](auto nc, int set) {
        constexpr int n = decltype(nc)::value;
        if constexpr (NWL == 1) vm_wait_imm<n>(wq[set]);
        else vm_wait_imm<n>(wq[set], wsc[set]);
    };
    constexpr int NFB = PP ? 1 : 2;                    // fragment register sets
    u32x4 fa[NFB][8], fb[NFB][NT];
    auto read_a = [&](int buf, int mt, u32x4& x) { x = *reinterpret_cast<const u32x4*>(a_rd[buf >> 1] + (buf & 1) * kR4Stage + mt * 1024); };
    auto read_b = [&](int buf, int nt, u32x4& x) { x = *reinterpret_cast<const u32x4*>(b_rd[nt & 1][buf >> 1] + (buf & 1) * kR4Stage + nt * 1024); };

    // ---- prologue.  Queue order (the steady state's): W(0) W(1) | A(0) A(1) | W(2) A(2) | W(3) A(3) --------------------------------
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int part = 0; part < NWL; ++part) issue_w(st, st, part);
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int n = 0; n < NAL; ++n) issue_a(st, st, n);
#pragma unroll
    for (int st = 2; st < 4; ++st) {
#pragma unroll
        for (int part = 0; part < NWL; ++part) issue_w(st, st, part);
#pragma unroll
        for (int n = 0; n < NAL; ++n) issue_a(st, st, n);
    }
    wait_w(std::integral_constant<int, 4 * NAL + 2 * NWL>{}, 0);
    wait_w(std::integral_constant<int, 4 * NAL + 2 * NWL>{}, 1);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const scale_t sc = scale_of(wsc[st]);
#pragma unroll
        for (int i = 0; i < NFR; ++i) {
            u32x4 f;
#pragma unroll
            for (int d = 0; d < 4; ++d) f[d] = dequant_dword(st, i, d, sc);
            *reinterpret_cast<u32x4*>(b_wr + st * kR4Stage + i * 256) = f;
        }
    }
    if constexpr (PP) vm_wait_imm<2 * (NAL + NWL)>(); // A(0) and A(1) have landed (the ping-pong steps' rule: a stage's pieces are
    else vm_wait_imm<3 * NAL + 2 * NWL>();             // waited for one step before their first readers); otherwise A(0) alone
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NT; ++i) read_b(0, i, fb[0][i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) read_a(0, i, fa[0][i]);

    if constexpr (PP) {
        // ---- ping-pong steps.  Barrier events: group 0's k-th call meets group 1's k-th call, and group 1 made one call more at the
        // start, so between two events one group is in M(t) and the other in J(t - 1) / J(t).  What the barriers order:
        //   * J(t) reads stage t + 1: its B image was written in everybody's J(t - 1), its A pieces requested in J(t - 3) and waited
        //     for at the END of J(t - 1) (vmcnt(2 (NAL + NWL))) - both at least one event before either group's J(t);
        //   * J(t) refills buffer t & 3 (A pieces of stage t + 4): its last readers were the J(t - 1)s, one event earlier at least;
        //   * J(t) dequantises stage t + 2 (requested in J(t - 2): vmcnt(2 NAL + NWL) at its start) into buffer (t + 2) & 3.
        QL_R4_STAMP(1);
        if (wr == 1) __builtin_amdgcn_s_barrier();
        auto step_pp = [&](int t, auto bufc) {
            constexpr int BUF = decltype(bufc)::value, NXT = (BUF + 1) & 3, DQ = (BUF + 2) & 3;
            __syncthreads();
            __builtin_amdgcn_s_setprio(1);
            static_for<S>([&](auto qc) {
                constexpr int q = decltype(qc)::value, mt = q / NT, nt = q % NT;
                mfma16_acc<T>(acc[mt][nt], fb[0][nt], fa[0][mt]);
            });
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            wait_w(std::integral_constant<int, 2 * NAL + NWL>{}, DQ);       // W(t + 2) has landed
#pragma unroll
            for (int i = 0; i < NT; ++i) read_b(NXT, i, fb[0][i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) read_a(NXT, i, fa[0][i]);
            const scale_t sc = scale_of(wsc[DQ]);
#pragma unroll
            for (int i = 0; i < NFR; ++i) {
                u32x4 f;
#pragma unroll
                for (int d = 0; d < 4; ++d) f[d] = dequant_dword(DQ, i, d, sc);
                *reinterpret_cast<u32x4*>(b_wr + DQ * kR4Stage + i * 256) = f;
            }
#pragma unroll
            for (int part = 0; part < NWL; ++part) issue_w(t + 4, BUF, part);
#pragma unroll
            for (int n = 0; n < NAL; ++n) issue_a(t + 4, BUF, n);
            vm_wait_imm<2 * (NAL + NWL)>();            // this wave's pieces of A(t + 2) have landed
        };
        int t = 0;
        for (; t + 4 <= steps; t += 4) {
            step_pp(t, std::integral_constant<int, 0>{});
            step_pp(t + 1, std::integral_constant<int, 1>{});
            step_pp(t + 2, std::integral_constant<int, 2>{});
            step_pp(t + 3, std::integral_constant<int, 3>{});
        }
        if (t < steps) step_pp(t, std::integral_constant<int, 0>{});
        if (t + 1 < steps) step_pp(t + 1, std::integral_constant<int, 1>{});
        if (t + 2 < steps) step_pp(t + 2, std::integral_constant<int, 2>{});
        if (wr == 0) __builtin_amdgcn_s_barrier();     // the call group 1 made first
    } else {
    // ---- step t (buffer / register set t & 3 = BUF): row mt of the wave tile = NT MFMAs; the next stage's A fragment of row mt is read
    // behind the row's first MFMA (one row of slack: 9 live A fragments), its B fragments in the first rows, the dequant dwords of stage
    // t + 2 one per second MFMA, the requests of stage t + 4 in the second half ---------------------------------------------------------------
#ifdef QL_R4_STAMPS
    unsigned long long t_vm = 0, t_bar = 0;
#endif
    auto step = [&](int t, auto bufc) {
        constexpr int BUF = decltype(bufc)::value, NXT = (BUF + 1) & 3, DQ = (BUF + 2) & 3, cur = BUF & 1, nxt = cur ^ 1;
#if defined(QL_R4_STAMPS) && QL_R4_STAMPS > 1
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
        wait_w(std::integral_constant<int, 2 * NAL + NWL>{}, DQ);   // W(t + 2) and A(t + 1) have landed (this wave's pieces)
#if defined(QL_R4_STAMPS) && QL_R4_STAMPS > 1
        const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#if defined(QL_R4_STAMPS) && QL_R4_STAMPS > 1
        const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
        t_vm += ts1 - ts0;
        t_bar += ts2 - ts1;
#endif
        const scale_t sc = scale_of(wsc[DQ]);
        u32x4 f;
        static_for<S>([&](auto qc) {
            constexpr int q = decltype(qc)::value, mt = q / NT, nt = q % NT;
            mfma16_acc<T>(acc[mt][nt], fb[cur][nt], fa[cur][mt]);
            // at most one job behind an MFMA, jobs of a kind every fourth MFMA (one wave issues in order: whatever stands between two
            // of its MFMAs beyond ~3 instructions delays the second one)
            if constexpr (nt == 0) read_a(NXT, mt, fa[nxt][mt]);                                   // q = NT mt     (q % 4 == 0)
            if constexpr ((q & 7) == 2 && (q >> 3) < NT) read_b(NXT, q >> 3, fb[nxt][q >> 3]);      // q = 8 nt + 2  (q % 4 == 2)
            if constexpr ((q & 3) == 1 && (q >> 2) < 4 * NFR) {    // one dword of a fragment at a time; fragment stored when complete
                constexpr int p = q >> 2, i = p >> 2, d = p & 3;
                f[d] = dequant_dword(DQ, i, d, sc);
                if constexpr (d == 3) *reinterpret_cast<u32x4*>(b_wr + DQ * kR4Stage + i * 256) = f;
            }
            // queue order inside a step: the weight loads, THEN the A pieces (the wait counts above are written for that order)
            if constexpr (q >= S / 2 && (q & 3) == 3 && (q - S / 2) / 4 < NWL) issue_w(t + 4, BUF, (q - S / 2) / 4);
            if constexpr (q >= S / 2 && (q & 3) == 3 && (q - S / 2) / 4 >= NWL && (q - S / 2) / 4 < NWL + NAL) issue_a(t + 4, BUF, (q - S / 2) / 4 - NWL);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    static_assert(S / 4 >= 4 * NFR && S / 8 >= NWL + NAL && S / 8 >= NT, "every job has its slot");
    QL_R4_STAMP(1);
#ifdef QL_R4_YOUNG_PRIO
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(QL_R4_YOUNG_PRIO);   // experiment: the younger wave of each SIMD served first
#endif
    int t = 0;
    for (; t + 4 <= steps; t += 4) {
        step(t, std::integral_constant<int, 0>{});
        step(t + 1, std::integral_constant<int, 1>{});
        step(t + 2, std::integral_constant<int, 2>{});
        step(t + 3, std::integral_constant<int, 3>{});
    }
    if (t < steps) step(t, std::integral_constant<int, 0>{});
    if (t + 1 < steps) step(t + 1, std::integral_constant<int, 1>{});
    if (t + 2 < steps) step(t + 2, std::integral_constant<int, 2>{});
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) wait_w(std::integral_constant<int, 0>{}, st);   // the queue is empty before the registers go out of scope
    QL_R4_STAMP(2);

    // ---- epilogue: 8 consecutive columns of one row per lane and tile pair -> one 16-byte store ---------------------------------------------
    const int mw = m0 + 128 * wr, nw = n0 + 16 * NT * wc;
    const bool wide = GATE || ((ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
                               (!resid || ((ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(resid) & 15) == 0)));   // GATE: 8-byte stores, checked by the ABI
    static_for<NT / 2>([&](auto ppc) {
        constexpr int pp = decltype(ppc)::value;
        const int nb = nw + 32 * pp + 8 * kq;          // first of the lane's 8 columns
        float bs[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) bs[r] = (bias && nb + r < N) ? Act<T>::load(bias + nb + r) : 0.f;
        static_for<8>([&](auto mtc) {
            constexpr int mt = decltype(mtc)::value;
            const int m = mw + 16 * mt + c16;
            float y[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                y[r] = Act<T>::round(r < 4 ? acc[mt][2 * pp][r & 3] : acc[mt][2 * pp + 1][r & 3]);
                if (bias) y[r] = Act<T>::round(y[r] + bs[r]);
            }
            if (m >= M || nb >= N) return;
            if constexpr (GATE) {                      // two (h0, h1, gate0, gate1) quads -> out[nb / 2 + 0..3] = round(round(silu(h)) * gate)
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float hv = y[4 * (i >> 1) + (i & 1)], gv = y[4 * (i >> 1) + 2 + (i & 1)];
                    o[i] = Act<T>::round(hv / (1.0f + __expf(-hv))) * gv;
                }
                T* dst = C + (int64_t)m * ldc + (nb >> 1);
                if (nb + 8 <= N) *reinterpret_cast<u32x2*>(dst) = u32x2{pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])};
                else *reinterpret_cast<u32*>(dst) = pack2<T>(o[0], o[1]);                    // N % 8 == 4: the last quad alone
            } else {
                T* dst = C + (int64_t)m * ldc + nb;
                if (wide && nb + 8 <= N) {
                    if (resid) {
                        float r[8];
                        unpack8<T>(*reinterpret_cast<const u32x4*>(resid + (int64_t)m * ldr + nb), r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) y[e] += r[e];
                    }
                    *reinterpret_cast<u32x4*>(dst) = pack8<T>(y);
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        if (nb + r < N) Act<T>::store(dst + r, resid ? y[r] + Act<T>::load(resid + (int64_t)m * ldr + nb + r) : y[r]);
                }
            }
        });
    });
#ifdef QL_R4_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left
    QL_R4_STAMP(3);
    if (stamp_on && lane == 0 && blockIdx.x < 16384) {
        unsigned long long* o = ql_r4_stamps + (size_t)blockIdx.x * 12;
        o[10] = t_vm; o[11] = t_bar;
        for (int i = 0; i < 4; ++i) { o[i] = st_rt[i]; o[4 + i] = st_ck[i]; }
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[8] = hw; o[9] = xcc;
    }
#endif
}

#ifndef QL_G256_NW
#define QL_G256_NW 8
#endif
template <typename T, bool W8, bool GATE, int NW, bool PP>
static void launch_r4_nw(unsigned grid, hipStream_t st, const T* A, const u32x4* Wt, const T* Sp, int M, int N, int steps, int wsteps, int64_t lda,
                         int nbx, int sy, const T* bias, T* C, int64_t ldc, const T* resid, int64_t ldr) {
    static bool attr4 = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&w4_gemm256_r4_kernel<T, W8, GATE, NW, PP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kR4Lds) == hipSuccess;
    }();
    (void)attr4;
    w4_gemm256_r4_kernel<T, W8, GATE, NW, PP><<<grid, NW * 64, kR4Lds, st>>>(A, Wt, Sp, M, N, steps, wsteps, lda, nbx, sy, bias, C, ldc, resid, ldr);
}
template <typename T, bool W8, bool GATE>
static void launch_r4(unsigned grid, hipStream_t st, const T* A, const u32x4* Wt, const T* Sp, int M, int N, int steps, int wsteps, int64_t lda,
                      int nbx, int sy, const T* bias, T* C, int64_t ldc, const T* resid, int64_t ldr) {
#ifndef QL_R4_PP
#define QL_R4_PP 0
#endif
#ifdef QL_DEV_TUNING
    const int nw = QL_TUNE("QLINEAR_G256_NW", 8), pp = QL_TUNE("QLINEAR_G256_PP", QL_R4_PP);
    if (nw == 4) return launch_r4_nw<T, W8, GATE, 4, false>(grid, st, A, Wt, Sp, M, N, steps, wsteps, lda, nbx, sy, bias, C, ldc, resid, ldr);
    if ((pp != 0) != (QL_R4_PP != 0))
        return launch_r4_nw<T, W8, GATE, 8, QL_R4_PP == 0>(grid, st, A, Wt, Sp, M, N, steps, wsteps, lda, nbx, sy, bias, C, ldc, resid, ldr);
#endif
    launch_r4_nw<T, W8, GATE, 8, QL_R4_PP != 0>(grid, st, A, Wt, Sp, M, N, steps, wsteps, lda, nbx, sy, bias, C, ldc, resid, ldr);
}

#ifndef QL_G256_RING
#define QL_G256_RING 0
#endif

#endif  // QL_DEV_TUNING

// whether `rem` left-over tiles of a round of `grid` workgroups run as half tiles: every unit 16 (t / 8) + (t % 8) + 8 half must fit a round
static inline bool g256_tail_halves(int rem, int grid) { return rem > 0 && 16 * ((rem + 7) / 8) <= grid; }

// the 16x16x32 launch of both weight formats: whole tiles, then the left-over tiles as half tiles (w4_gemm256x16_kernel)
template <typename T, bool W8, bool GATE>
static int launch_x16(const T* A, const u32x4* Wt, const T* Sp, int M, int N, int ksteps, int64_t lda, int nbx, int nby, int order, int sy,
                      const T* bias, T* C, int64_t ldc, const T* resid, int64_t ldr, hipStream_t st) {
    const int total = nbx * nby, cus = cu_count();
    const int grid_p = QL_TUNE("QLINEAR_G256_PGRID", 0) > 0 ? QL_TUNE("QLINEAR_G256_PGRID", 0) : (cus & ~7);   // workgroups per round (developer build: tests)
    const int rem = total > grid_p && grid_p >= 16 ? total % grid_p : 0;
    const bool halves = QL_TUNE("QLINEAR_G256_TAIL", 1) && !(dispatch_flags() & QL_D_NOHALF) && g256_tail_halves(rem, grid_p);
    const int whole = halves ? total - rem : total, units = halves ? 16 * ((rem + 7) / 8) : 0;
    static bool attr16 = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&w4_gemm256x16_kernel<T, W8, GATE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kG256Lds) == hipSuccess;
    }();
    (void)attr16;
    w4_gemm256x16_kernel<T, W8, GATE><<<(unsigned)(whole + units), 512, kG256Lds, st>>>(A, Wt, Sp, M, N, ksteps, lda, order, sy, total, whole, bias, C, ldc,
                                                                                         resid, ldr);
    return finish_launch(W8 ? QL_K_W8_GEMM256 : QL_K_W4_GEMM256);
}

template <typename T, bool GATE = false>
static int launch_gemm256(const void* A, const void* tiled, const void* bias, void* C, int M, int N, int K, int64_t lda, int64_t ldc,
                          hipStream_t st, const void* resid = nullptr, int64_t ldr = 0) {
    const W4Layout L = w4_layout(N, K, sizeof(T));
    const u32x4* Wt = (const u32x4*)tiled;
    const T* Sp = (const T*)((const char*)tiled + (L.off_sm - L.off_wm));
    const int nbx = (N + 255) / 256, nby = (M + 255) / 256;
    const bool no_super = QL_TUNE("QLINEAR_GEMM_SUPER", 1) == 0;
    const int sy = QL_TUNE("QLINEAR_GEMM_SY", 4);      // 32 blocks in flight per XCD: 8 columns x 4 rows share 12 operand panels
    // grouped order (ql_common.h: xcd_tile_super) for column counts that are a multiple of 8 or wide (w_in: 107 column tiles, int8 x int8
    // 1.80 -> 2.05 POP/s, int4g32 +1.5 %); 18 column tiles (qkv_proj) measured better in whole rows (tools/ab/run_sy_sweep.sh: 1.78 vs 1.68 POP/s)
    const bool super = !no_super && nby >= 2 && (nbx % 8 == 0 || nbx >= 32);
#ifdef QL_DEV_TUNING
    if (QL_TUNE("QLINEAR_G256_RING", QL_G256_RING)) {
        launch_r4<T, false, GATE>((unsigned)(nbx * nby), st, (const T*)A, Wt, Sp, M, N, K / 32, (int)L.ksteps, lda,
                                  super ? nbx : xcd_order(nbx, nby, (double)M * K * 2, (double)N * K * 0.5), super ? sy : 0, (const T*)bias, (T*)C, ldc,
                                  (const T*)resid, ldr);
        return finish_launch(QL_K_W4_GEMM256);
    }
#endif
    // the 16x16x32 body, the tiles left over after the whole rounds as half tiles behind the whole ones (w4_gemm256x16_kernel); the developer
    // build's QLINEAR_G256_MI16=0 selects round 4's 32x32x16 body (A/B partner, profiles/r05_g256_ab.txt)
    if (QL_TUNE("QLINEAR_G256_MI16", 1))
        return launch_x16<T, false, GATE>((const T*)A, Wt, Sp, M, N, (int)L.ksteps, lda, nbx, nby,
                                          super ? nbx : xcd_order(nbx, nby, (double)M * K * 2, (double)N * K * 0.5), super ? sy : 0, (const T*)bias, (T*)C,
                                          ldc, (const T*)resid, ldr, st);
#ifdef QL_DEV_TUNING                                 // QLINEAR_G256_MI16=0: round 4's 32x32x16 body (developer library only)
    static bool attr_set = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&w4_gemm256_kernel<T, false, GATE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kG256Lds) == hipSuccess;
    }();
    (void)attr_set;
    w4_gemm256_kernel<T, false, GATE><<<(unsigned)(nbx * nby), 512, kG256Lds, st>>>(
        (const T*)A, Wt, Sp, M, N, (int)L.ksteps, lda, super ? nbx : xcd_order(nbx, nby, (double)M * K * 2, (double)N * K * 0.5),
        super ? sy : 0, (const T*)bias, (T*)C, ldc, (const T*)resid, ldr);
    return finish_launch(QL_K_W4_GEMM256);
#else
    return QL_ERR_UNSUPPORTED;                         // unreachable: the product's QLINEAR_G256_MI16 is the constant 1
#endif
}

template <typename T, bool GATE = false>
static int launch_gemm256_w8(const void* A, const int8_t* Wm, const void* S, const void* bias, void* C, int M, int N, int K, int64_t lda,
                             int64_t ldc, hipStream_t st, const void* resid = nullptr, int64_t ldr = 0) {
    const int nbx = (N + 255) / 256, nby = (M + 255) / 256;
    const bool no_super = QL_TUNE("QLINEAR_GEMM_SUPER", 1) == 0;
    const int sy = QL_TUNE("QLINEAR_GEMM_SY", 4);
    // grouped order (ql_common.h: xcd_tile_super) for column counts that are a multiple of 8 or wide (w_in: 107 column tiles, int8 x int8
    // 1.80 -> 2.05 POP/s, int4g32 +1.5 %); 18 column tiles (qkv_proj) measured better in whole rows (tools/ab/run_sy_sweep.sh: 1.78 vs 1.68 POP/s)
    const bool super = !no_super && nby >= 2 && (nbx % 8 == 0 || nbx >= 32);
#ifdef QL_DEV_TUNING
    if (QL_TUNE("QLINEAR_G256_RING", QL_G256_RING)) {
        launch_r4<T, true, GATE>((unsigned)(nbx * nby), st, (const T*)A, (const u32x4*)Wm, (const T*)S, M, N, K / 32, K / 64, lda,
                                 super ? nbx : xcd_order(nbx, nby, (double)M * K * 2, (double)N * K), super ? sy : 0, (const T*)bias, (T*)C, ldc,
                                 (const T*)resid, ldr);
        return finish_launch(QL_K_W8_GEMM256);
    }
#endif
    if (QL_TUNE("QLINEAR_G256_MI16", 1))
        return launch_x16<T, true, GATE>((const T*)A, (const u32x4*)Wm, (const T*)S, M, N, K / 64, lda, nbx, nby,
                                         super ? nbx : xcd_order(nbx, nby, (double)M * K * 2, (double)N * K), super ? sy : 0, (const T*)bias, (T*)C, ldc,
                                         (const T*)resid, ldr, st);
#ifdef QL_DEV_TUNING
    static bool attr_set = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&w4_gemm256_kernel<T, true, GATE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kG256Lds) == hipSuccess;
    }();
    (void)attr_set;
    w4_gemm256_kernel<T, true, GATE><<<(unsigned)(nbx * nby), 512, kG256Lds, st>>>(
        (const T*)A, (const u32x4*)Wm, (const T*)S, M, N, K / 64, lda, super ? nbx : xcd_order(nbx, nby, (double)M * K * 2, (double)N * K),
        super ? sy : 0, (const T*)bias, (T*)C, ldc, (const T*)resid, ldr);
    return finish_launch(QL_K_W8_GEMM256);
#else
    return QL_ERR_UNSUPPORTED;                         // unreachable: the product's QLINEAR_G256_MI16 is the constant 1
#endif
}

// int8 weight-only with the epilogues of the int4 kernel: SiLU * gate on a gate-interleaved copy (C (M, N / 2)), residual add
int w8_gemm256_gated(int dtype, const void* A, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                     int64_t lda, int64_t ldc, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_gemm256_w8<f16, true>(A, Wm, S, bias, C, (int)M, (int)N, (int)K, lda, ldc, st);
    case QL_DTYPE_BF16: return launch_gemm256_w8<__bf16, true>(A, Wm, S, bias, C, (int)M, (int)N, (int)K, lda, ldc, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}
int w8_gemm256_residual(int dtype, const void* A, const int8_t* Wm, const void* S, const void* bias, const void* resid, void* C, int64_t M,
                        int64_t N, int64_t K, int64_t lda, int64_t ldc, int64_t ldr, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_gemm256_w8<f16>(A, Wm, S, bias, C, (int)M, (int)N, (int)K, lda, ldc, st, resid, ldr);
    case QL_DTYPE_BF16: return launch_gemm256_w8<__bf16>(A, Wm, S, bias, C, (int)M, (int)N, (int)K, lda, ldc, st, resid, ldr);
    default: return QL_ERR_BAD_DTYPE;
    }
}

// int8 per-channel weights (tile-major copy) through the same kernel: C = A . (W^T * s[n]) with the reference's per-weight rounding
int w8_gemm256(int dtype, const void* A, const int8_t* Wm, const void* S, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
               int64_t lda, int64_t ldc, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_gemm256_w8<f16>(A, Wm, S, bias, C, (int)M, (int)N, (int)K, lda, ldc, st);
    case QL_DTYPE_BF16: return launch_gemm256_w8<__bf16>(A, Wm, S, bias, C, (int)M, (int)N, (int)K, lda, ldc, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

// rows / shapes the 256 x 256 kernel takes (the launcher in w4_gemm.hip asks): whole 64-deep K tiles, 16-byte aligned rows,
// 32-bit byte offsets into A, and enough 256-row tiles to fill the chip
bool w4_gemm256_supported(int64_t M, int64_t N, int64_t K, int64_t lda, const void* A, size_t esize) {
    if ((dispatch_flags() & QL_D_NO256) || esize != 2 || K % 64 != 0 || K < 128 || (lda * (int64_t)esize) % 16 != 0 || ((uintptr_t)A & 15) != 0) return false;
    if (M * lda * (int64_t)esize >= ((int64_t)1 << 31)) return false;
    const int64_t blocks = ((N + 255) / 256) * ((M + 255) / 256);
    const int min_blocks = QL_TUNE("QLINEAR_GEMM_256_MIN_BLOCKS", 0);   // tuning sweeps (developer build)
    if (min_blocks > 0) return M >= 256 && blocks >= min_blocks;
    // one block per CU at a time (128 KB of LDS): the grid pays in whole rounds of cu_count() (256) blocks.  Measured against the 128-row-tile
    // kernel (tools/prefill_gemm_ab.py, M = 1024 .. 8192 x the four layer shapes): ahead from one full round on when the last
    // round is at least ~70 % full (256 blocks +11 %, 428 +13 %, 576 +3.5 %, 856 +3 %), behind below that (288 blocks -9 %, 128 -25 %)
    const int64_t cus = cu_count(), rounds = (blocks + cus - 1) / cus;
    return blocks >= cus && blocks * 10 >= rounds * cus * 7;
}

// int4g32: whether the launch runs the tiles left over after the whole rounds as half tiles of its own (then nothing is peeled off to the
// 128-row-tile kernel: w4_gemm.hip, w4_gemm256_rows)
bool w4_gemm256_tail_in_kernel(int64_t blocks, int64_t K) {
    const int grid = cu_count() & ~7;
    (void)K;
    return QL_TUNE("QLINEAR_G256_MI16", 1) && !(dispatch_flags() & QL_D_NOHALF) && QL_TUNE("QLINEAR_G256_TAIL", 1) &&
           QL_TUNE("QLINEAR_G256_PGRID", 0) == 0 && grid >= 16 && blocks > grid && g256_tail_halves((int)(blocks % grid), grid);
}

// what the kernel itself needs (the dispatch heuristic above is a speed choice on top of this)
bool w4_gemm256_can_run(int64_t M, int64_t N, int64_t K, int64_t lda, const void* A, size_t esize) {
    return esize == 2 && M > 0 && N > 0 && K % 64 == 0 && K >= 128 && (lda * (int64_t)esize) % 16 == 0 && ((uintptr_t)A & 15) == 0 &&
           M * lda * (int64_t)esize < ((int64_t)1 << 31);
}

int w4_gemm256(int dtype, const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
               int64_t ldc, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_gemm256<f16>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, st);
    case QL_DTYPE_BF16: return launch_gemm256<__bf16>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

// C = round(round(A . dequant(W) (+ bias)) + resid): the residual add of the block in the GEMM's epilogue (16-byte aligned C / resid rows)
int w4_gemm256_residual(int dtype, const void* A, const void* tiled, const void* bias, const void* resid, void* C, int64_t M, int64_t N,
                        int64_t K, int64_t lda, int64_t ldc, int64_t ldr, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_gemm256<f16>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, st, resid, ldr);
    case QL_DTYPE_BF16: return launch_gemm256<__bf16>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, st, resid, ldr);
    default: return QL_ERR_BAD_DTYPE;
    }
}

// gate-interleaved copy of a first MLP projection, SiLU * gate in the epilogue: C is (M, N / 2)
int w4_gemm256_gated(int dtype, const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                     int64_t ldc, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_gemm256<f16, true>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, st);
    case QL_DTYPE_BF16: return launch_gemm256<__bf16, true>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

}  // namespace ql

#if defined(QL_R4_STAMPS) && defined(QL_DEV_TUNING)
extern "C" int qlinear_r4_stamps_read(unsigned long long* out, int blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ql::ql_r4_stamps), sizeof(unsigned long long) * 12 * blocks);
}
#endif
#ifdef QL_G256_STAMPS
extern "C" int qlinear_g256_stamps_read(unsigned long long* out, int blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ql::ql_g256_stamps), sizeof(unsigned long long) * 8 * blocks);
}
#endif
