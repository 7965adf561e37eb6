// int4g32 forward for 3 .. 16 activation rows (batched decode) on PART 1 of the derived layout, gfx950 - round 5.
//
// What it replaces at these row counts: w4_fewrow_kernel (w4_fewrow.hip: 32 columns x a K slice per workgroup on part 2, K also split over
// workgroups into fp32 slabs + splitk_reduce_kernel for every matrix narrower than ~8 k columns).  A batch-8 decode step spent 20 % of its
// time in those reduce launches and 59 % in kernels running at half the one-row GEMV's byte rate (tools/ab/run_batched_decode_profile.sh).
// Here the GEMV's execution shape is kept - many small independent waves, every weight byte requested in the wave's first instructions,
// no cross-workgroup reduction - and the arithmetic goes to v_mfma_f32_16x16x32:
//   * workgroup = 16 NT output columns x ALL of K, KW waves; wave w walks the 128-k blocks w, w + KW, ... (four int4 groups each);
//   * lane (n = lane & 15, kq = lane >> 4) requests the 16-byte unit of column n, group 4 blk + kq - straight from part 1 (Wt[n][g]: 64
//     contiguous bytes per column and request, the 8 waves' blocks contiguous behind each other) - and its scale;
//   * the block's four MFMAs s = 0 .. 3 contract k = 32 (4 blk + kq) + 8 s .. + 7: the B operand is word s of the lane's unit, dequantised in
//     registers with the reference's rounding ((n - 8) * s rounded to the activation dtype, chatglm_q/int4/triton_ops.py:72-73); the A
//     operand is the activation rows' 16 bytes at that k (rows m = lane & 15, clamped to M - 1), read straight from global / L1 - the four
//     reads of a block touch the same lines;
//   * requests of a block go out A first, then the unit: they return in order, so block i's arithmetic runs when ITS unit has landed while
//     the later units are still in flight (D blocks per wave in flight);
//   * the waves' partial tiles are summed through LDS in a fixed order by 256 NT threads, which also run the epilogue (bias; SiLU * gate
//     on gate-interleaved columns, chatglm_q/model.py:200-201) - one launch, deterministic.
// No part 2 needed: a decode-only session (low-footprint mode) serves batched steps from the GEMV's copy.
#include "launch.h"
#include "w4_mma.h"

namespace ql {

typedef float r16_f32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mma16x16;
template <> struct Mma16x16<f16> {
    static __device__ __forceinline__ r16_f32x4 mma(f16x8 a, f16x8 b, r16_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mma16x16<__bf16> {
    static __device__ __forceinline__ r16_f32x4 mma(bf16x8 a, bf16x8 b, r16_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

template <typename T, int NT, int KW, bool GATE, int D, int MT = 1>
__global__ __launch_bounds__(KW * 64) void w4_rows16_kernel(const T* __restrict__ A, const u32x4* __restrict__ Wt, const T* __restrict__ Sp,
                                                                   int M, int N, int Npad, int G, int lda, const T* __restrict__ bias,
                                                                   T* __restrict__ C, int64_t ldc) {
    typedef Mma<T> MM;
    typedef typename MM::frag frag;
    // D: blocks of a wave in flight (21 registers per block with one column tile, 26 with two)
    // MT: 16-row tiles (1: up to 16 rows, 2: up to 32 - every weight fragment then feeds two MFMAs)
    __shared__ float red[KW][MT][NT][4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kq = lane >> 4;
    const int nblk = (G + 3) >> 2;                             // 128-k blocks
    const int mine = wave < nblk ? (nblk - wave + KW - 1) / KW : 0;   // blocks of this wave: wave, wave + KW, ...

    u32 k_mask_lo, k_mask_hi, k_magic;
    asm volatile("s_mov_b32 %0, 0x000F000F" : "=s"(k_mask_lo));
    asm volatile("s_mov_b32 %0, 0x00F000F0" : "=s"(k_mask_hi));
    asm volatile("v_mov_b32 %0, %1" : "=v"(k_magic) : "i"(MM::kMagic));

    // per column tile: the lane's column (clamped: loads stay inside part 1, stores are masked), its unit row and scale row
    const u32x4* wcol[NT];
    const T* scol[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n_raw = ((int)blockIdx.x * NT + t) * 16 + n16;
        const int n = n_raw < Npad ? n_raw : Npad - 1;
        wcol[t] = Wt + (int64_t)n * G;
        scol[t] = Sp + ((int64_t)(n >> 2) * G) * 4 + (n & 3);
    }
    const T* arow[MT];                                         // A operand: row m = 16 mt + (lane & 15)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) arow[mt] = A + (int64_t)(16 * mt + n16 < M ? 16 * mt + n16 : M - 1) * lda;

    struct Stage {
        u32x4 a[MT][4];
        u32x4 w[NT];
        T s[NT];
    };
    auto load_stage = [&](int i, Stage& sg) {                  // the wave's i-th block (clamped past the end: requested, never used)
        const int blk = wave + KW * (i < mine ? i : (mine > 0 ? mine - 1 : 0));
        const int g_raw = 4 * blk + kq, g = g_raw < G ? g_raw : G - 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const T* ap = arow[mt] + 32 * g;
#pragma unroll
            for (int s = 0; s < 4; ++s) sg.a[mt][s] = u32x4{0u, 0u, 0u, 0u};
            if (16 * mt + n16 < M) {                           // rows past M: lanes switched off (their output rows are never stored)
#pragma unroll
                for (int s = 0; s < 4; ++s) sg.a[mt][s] = *reinterpret_cast<const u32x4*>(ap + 8 * s);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            sg.w[t] = __builtin_nontemporal_load(wcol[t] + g);   // streamed once
            sg.s[t] = scol[t][(int64_t)g * 4];
        }
    };
    r16_f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = r16_f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int i, const Stage& sg) {
        const int g_raw = 4 * (wave + KW * i) + kq;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const T s_eff = g_raw < G ? sg.s[t] : (T)0.f;      // the missing groups of a ragged last block contribute 0
            const auto sc = MM::scale_pair(&s_eff, true);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const frag b = MM::dequant(sg.w[t][s], k_mask_lo, k_mask_hi, k_magic, sc);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][t] = Mma16x16<T>::mma(__builtin_bit_cast(frag, sg.a[mt][s]), b, acc[mt][t]);
            }
        }
    };

    // bias of the epilogue thread's column(s): requested up front, not in the tail (output e = tid + j KW 64: column e & 15, row 16 mt + ((e >> 4)
    // & 15) of tile e >> 8 = mt NT + t)
    constexpr int NOUT = 256 * MT * NT;
    constexpr int EPT = (NOUT + KW * 64 - 1) / (KW * 64);     // outputs per thread
    float bias_v[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int e = tid + j * KW * 64;
        const int n = ((int)blockIdx.x * NT + ((e >> 8) % NT)) * 16 + (e & 15);
        bias_v[j] = Act<T>::load((bias ? bias : Sp) + (bias && e < NOUT && n < N ? n : 0));
    }
    if (mine > 0) {
        Stage st[D];
#pragma unroll
        for (int d = 0; d < D; ++d) load_stage(d, st[d]);
        int i = 0;
        for (; i + D < mine; i += D) {                         // steady state: compute block i + d, request block i + d + D into its registers
#pragma unroll
            for (int d = 0; d < D; ++d) {
                Stage cur = st[d];
                load_stage(i + d + D, st[d]);
                compute(i + d, cur);
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (i + d < mine) compute(i + d, st[d]);
    }

    // sum the KW partial tiles in wave order: thread e < 256 NT takes output (row = (e >> 4) & 15, column e & 15) of tile e >> 8
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][mt][t][r][lane] = acc[mt][t][r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int e = tid + j * KW * 64;
        if (e >= NOUT) break;                                  // (wave-uniform: KW 64 and NOUT are multiples of 64)
        const int mt = (e >> 8) / NT, t = (e >> 8) % NT, r16 = (e >> 4) & 15, col = e & 15, row = 16 * mt + r16;
        const int src = (r16 >> 2) * 16 + col;                 // the lane that holds (row, col): rows 4 q .. 4 q + 3 of column `col`
        float sum = red[0][mt][t][r16 & 3][src];
#pragma unroll
        for (int w = 1; w < KW; ++w) sum += red[w][mt][t][r16 & 3][src];
        const int n = ((int)blockIdx.x * NT + t) * 16 + col;
        if constexpr (GATE) {
            // SiLU * gate on gate-interleaved columns: columns 4 p, 4 p + 1 hold h, 4 p + 2, 4 p + 3 the gates of the same output pair; C gets
            // N / 2 columns (N % 4 == 0).  y = rounded sum (+ bias, rounded); out = round(round(silu(y_h)) * y_gate)
            float y = Act<T>::round(sum);
            if (bias) y = Act<T>::round(y + bias_v[j]);
            const float yg = __shfl_down(y, 2);                // every lane takes part
            if (row < M && n < N && (col & 2) == 0)
                Act<T>::store(C + (int64_t)row * ldc + ((n >> 2) << 1) + (col & 1), Act<T>::round(Act<T>::round(y / (1.0f + __expf(-y))) * yg));
        } else {
            if (row < M && n < N) {                            // store_out's sequence with the preloaded bias
                float y = Act<T>::round(sum);
                if (bias) y = y + bias_v[j];
                Act<T>::store(C + (int64_t)row * ldc + n, y);
            }
        }
    }
}

// ---- WIDE matrices (a first MLP projection: 27 392 columns): the activation rows in LDS, no K split ---------------------------------------------
// With one workgroup per 16 / 32 columns every workgroup re-reads the M x K activation rows (224 MB of L2 -> L1 traffic at 16 rows against 63 MB
// of weights, and the lanes' 16-byte pieces of 16 different rows cost four address-unit passes per request).  Here a workgroup of 8 waves stages
// the rows ONCE into LDS (row pitch 2 K + 16 bytes: the 16 rows of a fragment read fall into 16 different 16-byte bank groups) and every wave
// then owns 16 columns over ALL of K: no partial tiles, no reduction, the epilogue runs from the accumulator registers (column n of rows 4 q ..
// 4 q + 3 per lane; the SiLU * gate partner column n + 2 is two lanes away).  A block's requests are the unit and its scale only (5 registers):
// eight blocks per wave in flight.  Measured level with the few-row kernel (see rows16_cfg below): its point is the layout, not the time.
template <typename T, bool GATE, int D>
__global__ __launch_bounds__(512) void w4_rows16w_kernel(const T* __restrict__ A, const u32x4* __restrict__ Wt, const T* __restrict__ Sp, int M, int N,
                                                         int Npad, int G, int lda, const T* __restrict__ bias, T* __restrict__ C, int64_t ldc) {
    typedef Mma<T> MM;
    typedef typename MM::frag frag;
    extern __shared__ __attribute__((aligned(16))) char a_lds[];   // M rows x (2 K + 16) bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kq = lane >> 4;
    const int K = G * 32, pitch = 2 * K + 16;
    const int nblk = (G + 3) >> 2;

    u32 k_mask_lo, k_mask_hi, k_magic;
    asm volatile("s_mov_b32 %0, 0x000F000F" : "=s"(k_mask_lo));
    asm volatile("s_mov_b32 %0, 0x00F000F0" : "=s"(k_mask_hi));
    asm volatile("v_mov_b32 %0, %1" : "=v"(k_magic) : "i"(MM::kMagic));

    // (1) the activation rows: 16-byte chunks, consecutive lanes consecutive chunks of a row - requested first (they come from L2 / the
    // memory-side cache, are needed first, and requests return in order), then (2) the first D blocks of this wave's 16 columns, and only then
    // the rows go to LDS: the weight stream is under way while they arrive
    const int cpr = K / 8, chunks = M * cpr;                  // chunks per row, in all
    constexpr int SR = 16;                                     // staging requests per thread and pass (16 rows x 4096: one pass)
    int m0 = tid / cpr, k0 = tid - m0 * cpr;                   // chunk tid: row, 16-byte column (stepped by 512 below: no more divisions)
    auto stage_request = [&](u32x4 (&stg)[SR], int (&dst)[SR], int c0, int& m, int& k8) {
#pragma unroll
        for (int u = 0; u < SR; ++u) {
            const bool ok = c0 + tid + 512 * u < chunks;
            const int mm = ok ? m : M - 1, kk = ok ? k8 : 0;
            stg[u] = *reinterpret_cast<const u32x4*>(A + (int64_t)mm * lda + 8 * kk);
            dst[u] = ok ? mm * pitch + 16 * kk : -1;
            k8 += 512;
            while (k8 >= cpr) k8 -= cpr, ++m;
        }
    };
    auto stage_store = [&](const u32x4 (&stg)[SR], const int (&dst)[SR]) {
#pragma unroll
        for (int u = 0; u < SR; ++u)
            if (dst[u] >= 0) *reinterpret_cast<u32x4*>(a_lds + dst[u]) = stg[u];
    };
    u32x4 stg[SR];
    int dst[SR];
    stage_request(stg, dst, 0, m0, k0);
    const int n_raw = ((int)blockIdx.x * 8 + wave) * 16 + n16;
    const int n = n_raw < Npad ? n_raw : Npad - 1;
    const u32x4* wcol = Wt + (int64_t)n * G;
    const T* scol = Sp + ((int64_t)(n >> 2) * G) * 4 + (n & 3);
    const float bias_v = Act<T>::load((bias ? bias : Sp) + (bias && n_raw < N ? n_raw : 0));
    struct Stage {
        u32x4 w;
        T s;
    };
    auto load_stage = [&](int i, Stage& sg) {
        const int blk = i < nblk ? i : nblk - 1;
        const int g_raw = 4 * blk + kq, g = g_raw < G ? g_raw : G - 1;
        sg.w = __builtin_nontemporal_load(wcol + g);
        sg.s = scol[(int64_t)g * 4];
    };
    Stage st[D];
#pragma unroll
    for (int d = 0; d < D; ++d) load_stage(d, st[d]);
    stage_store(stg, dst);
    for (int c0 = 512 * SR; c0 < chunks; c0 += 512 * SR) {     // (more than 64 K activation values: further passes behind the first units)
        stage_request(stg, dst, c0, m0, k0);
        stage_store(stg, dst);
    }
    // the rows are in LDS; NOT __syncthreads(): its fence would wait for the units in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const char* arow = a_lds + (n16 < M ? n16 : M - 1) * pitch + 64 * kq;   // A operand: row lane & 15, + 256 blk + 16 s
    r16_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int i, const Stage& sg) {
        const int g_raw = 4 * i + kq;
        const T s_eff = g_raw < G ? sg.s : (T)0.f;
        const auto sc = MM::scale_pair(&s_eff, true);
        const char* ap = arow + 256 * i - (g_raw < G ? 0 : 64 * (g_raw - (G - 1)));   // (a ragged last block: stay inside the row)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            acc = Mma16x16<T>::mma(__builtin_bit_cast(frag, *reinterpret_cast<const u32x4*>(ap + 16 * s)),
                                   MM::dequant(sg.w[s], k_mask_lo, k_mask_hi, k_magic, sc), acc);
    };
    int i = 0;
    for (; i + D < nblk; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const Stage cur = st[d];
            load_stage(i + d + D, st[d]);
            compute(i + d, cur);
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (i + d < nblk) compute(i + d, st[d]);

    // epilogue from registers: lane (n16, q) holds rows 4 q .. 4 q + 3 of its column
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * kq + r;
        float y = Act<T>::round(acc[r]);
        if constexpr (GATE) {
            if (bias) y = Act<T>::round(y + bias_v);
            const float yg = __shfl_down(y, 2);                // the gate of the same output pair: column n + 2
            if (row < M && n_raw < N && (n16 & 2) == 0)
                Act<T>::store(C + (int64_t)row * ldc + ((n_raw >> 2) << 1) + (n16 & 1), Act<T>::round(Act<T>::round(y / (1.0f + __expf(-y))) * yg));
        } else {
            if (bias) y = y + bias_v;
            if (row < M && n_raw < N) Act<T>::store(C + (int64_t)row * ldc + n_raw, y);
        }
    }
}

// column tiles per workgroup x waves per workgroup x blocks in flight, or nt = 0: not served.  Measured per layer shape against the few-row
// kernel + its reduce launch (tools/rows16_sweep.py, profiles/r05_rows16_sweep.txt; us at 5 / 8 / 16 rows):
//   N <= 16 x CUs, K <= 8192 (o_proj 4096 -> 4096: one workgroup per CU, one round)  1 x 8 x 4: 5.6 / 6.1 / 7.2  against 8.4 / 8.4 / 9.3
//   N <= 32 x CUs, K <= 8192 (qkv_proj 4096 -> 4608)                                 2 x 4 x 3: 7.9 / 8.5 / 9.8  against 9.0 / 9.3 / 10.4
//   N <= 16 x CUs, K  > 8192 (w_out 13696 -> 4096), up to 8 rows                     1 x 8 x 2: 13.9 / 14.7      against 14.8 / 15.2 (16 rows: 19.7 / 15.8)
//   wider (w_in 4096 -> 27392): every workgroup re-reads the activation rows for its 16 or 32 columns - 23.7 / 26.3 / 32.0 against 20.0 / 20.3 / 22.3;
//   with the rows in LDS (w4_rows16w_kernel, nt = -1; rows x (2 K + 16) bytes must fit 160 KB): 21.5 / 21.0 / 21.5 against 21.3 / 21.5 / 22.6 - level
//   with the few-row kernel (both are ~8 us of dequant VALU beside a 10 us stream that do not overlap: profiles/r05_w_in_fewrows_pmc.txt); what it
//   buys is that part 2 is not needed for these row counts
struct Rows16Cfg {
    int nt, kw, d;
};
static Rows16Cfg rows16_cfg(int64_t M, int64_t N, int64_t K) {
    const int64_t cus = cu_count();
    Rows16Cfg c{0, 8, 4};
    if (N <= 16 * cus && K <= 8192) c = {1, 8, 4};
    else if (N <= 32 * cus && K <= 8192) c = {2, 4, 3};
    else if (N <= 16 * cus && M <= 8) c = {1, 8, 2};
    else if (N > 32 * cus && M * (2 * K + 16) <= 160 * 1024 && QL_TUNE("QLINEAR_ROWS16_WIDE", 1)) c = {-1, 8, 8};   // wide: the rows in LDS (w4_rows16w_kernel)
    if (QL_TUNE("QLINEAR_ROWS16_NT", 0)) c.nt = QL_TUNE("QLINEAR_ROWS16_NT", 0), c.d = c.nt == 1 ? 4 : 3;
    if (QL_TUNE("QLINEAR_ROWS16_KW", 0)) c.kw = QL_TUNE("QLINEAR_ROWS16_KW", 0);
    if (QL_TUNE("QLINEAR_ROWS16_D", 0)) c.d = QL_TUNE("QLINEAR_ROWS16_D", 0);
    return c;
}

// 3 .. 16 rows of a 16-bit dtype with 16-byte aligned activation rows, shapes as above; QLINEAR_DISPATCH=norows16: the few-row kernel on part 2
bool w4_rows16_serves(int dtype, int64_t M, int64_t N, int64_t K, int64_t lda) {
    const int lo = QL_TUNE("QLINEAR_ROWS16_MIN", 3), hi = QL_TUNE("QLINEAR_ROWS16_MAX", 16);      // (developer build: 32 = the two-tile form, measured below)
    return !(dispatch_flags() & QL_D_NOROWS16) && (dtype == QL_DTYPE_F16 || dtype == QL_DTYPE_BF16) && M >= lo && M <= hi && N > 0 && K >= 128 &&
           K % 32 == 0 && lda % 8 == 0 && lda <= 0x7fffffff && rows16_cfg(M, N, K).nt != 0;
}

template <typename T, int NT, int KW, bool GATE, int D, int MT = 1>
static int launch_rows16_cfg(const void* A, const void* packed, const void* bias, void* C, int M, int N, int K, int64_t lda, int64_t ldc, hipStream_t st) {
    const W4Layout L = w4_layout(N, K, sizeof(T));
    const u32x4* Wt = (const u32x4*)packed;
    const T* Sp = (const T*)((const char*)packed + L.off_sp);
    w4_rows16_kernel<T, NT, KW, GATE, D, MT><<<(unsigned)((N + 16 * NT - 1) / (16 * NT)), KW * 64, 0, st>>>((const T*)A, Wt, Sp, M, N, (int)L.Npad, (int)L.G,
                                                                                                 (int)lda, (const T*)bias, (T*)C, ldc);
    return finish_launch(QL_K_W4_ROWS16);
}

template <typename T, bool GATE>
static int launch_rows16(const void* A, const void* packed, const void* bias, void* C, int M, int N, int K, int64_t lda, int64_t ldc, hipStream_t st) {
    const Rows16Cfg c = rows16_cfg(M, N, K);
    if (c.nt < 0) {
        const W4Layout L = w4_layout(N, K, sizeof(T));
        const int lds = M * (2 * K + 16);
        static bool attr = [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(&w4_rows16w_kernel<T, GATE, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024) == hipSuccess;
        }();
        (void)attr;
        w4_rows16w_kernel<T, GATE, 8><<<(unsigned)((N + 127) / 128), 512, (size_t)lds, st>>>(
            (const T*)A, (const u32x4*)packed, (const T*)((const char*)packed + L.off_sp), M, N, (int)L.Npad, (int)L.G, (int)lda, (const T*)bias, (T*)C,
            ldc);
        return finish_launch(QL_K_W4_ROWS16);
    }
#define QL_R16(NT_, KW_, D_) \
    if (c.nt == NT_ && c.kw == KW_ && c.d == D_) return launch_rows16_cfg<T, NT_, KW_, GATE, D_>(A, packed, bias, C, M, N, K, lda, ldc, st);
#ifdef QL_DEV_TUNING
    // two 16-row tiles (QLINEAR_ROWS16_MAX=32; tools/rows16_mt2_check.py): correct (1e-5 of the few-row kernel), and slower than it - o_proj at
    // 17 / 24 / 32 rows 11.4 / 13.1 / 15.0 us against 10.1 / 10.3 / 11.1, qkv_proj 11.7 / 12.1 / 12.8 against 10.8 / 11.4 / 12.6: the activation rows every
    // workgroup reads grow with M, the few-row kernel's staged tiles do not.  Not in the product.
    if (M > 16) {
        if (c.nt == 1 && c.kw == 8) return launch_rows16_cfg<T, 1, 8, GATE, 3, 2>(A, packed, bias, C, M, N, K, lda, ldc, st);
        if (c.nt == 2 && c.kw == 4) return launch_rows16_cfg<T, 2, 4, GATE, 2, 2>(A, packed, bias, C, M, N, K, lda, ldc, st);
        return QL_ERR_UNSUPPORTED;
    }
#else
    if (M > 16) return QL_ERR_UNSUPPORTED;
#endif
    QL_R16(1, 8, 4) QL_R16(2, 4, 3) QL_R16(1, 8, 2)
#ifdef QL_DEV_TUNING
    QL_R16(1, 8, 3) QL_R16(1, 4, 4) QL_R16(1, 2, 4) QL_R16(2, 8, 3) QL_R16(2, 2, 3)
#endif
#undef QL_R16
    return QL_ERR_UNSUPPORTED;
}

int w4_rows16(int dtype, const void* A, const void* packed, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc,
              hipStream_t st, bool gate) {
    if (gate && N % 4 != 0) return QL_ERR_BAD_SHAPE;
    if (dtype == QL_DTYPE_F16)
        return gate ? launch_rows16<f16, true>(A, packed, bias, C, (int)M, (int)N, (int)K, lda, ldc, st)
                    : launch_rows16<f16, false>(A, packed, bias, C, (int)M, (int)N, (int)K, lda, ldc, st);
    if (dtype == QL_DTYPE_BF16)
        return gate ? launch_rows16<__bf16, true>(A, packed, bias, C, (int)M, (int)N, (int)K, lda, ldc, st)
                    : launch_rows16<__bf16, false>(A, packed, bias, C, (int)M, (int)N, (int)K, lda, ldc, st);
    return QL_ERR_BAD_DTYPE;
}

}  // namespace ql
