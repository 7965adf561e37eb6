// int4g32 GEMM for FEW activation rows (5 <= M <= 32: batched decode, short prompts) on the derived layout, gfx950.
//
// The regime is bound by the weight stream like the GEMV, but wants MFMA for the arithmetic.  The tiled GEMM
// (w4_gemm.hip, one 32-row tile + split-K over workgroups) serves it at less than half the GEMV's byte rate: PMC and
// ablation show its dequant-VALU, LDS and MFMA phases adding up instead of overlapping, because the few resident
// waves of a CU march in lockstep between block barriers.  This kernel copies the GEMV's execution shape instead:
// many small INDEPENDENT waves.
//   block = KW waves that share 32 output columns and split K between them (no barrier until the end);
//   a wave walks its K slice in 64-deep steps exactly like a wave of w4_packed_gemm_kernel<T, 1, ...>: lane (j, kb)
//   takes the 16-byte unit of column j, group 2 kt + kb (tile-major part of the derived layout: 1 KB contiguous
//   per wave and step, consecutive steps consecutive), dequantises it (reference rounding) into the B fragments
//   of 4 MFMA sub-steps; the A tile (up to 32 rows x 64 halves) is staged by THE WAVE ITSELF into a private 4 KB
//   LDS region (swizzled, double buffered) - LDS operations of one wave execute in order, so no barrier is needed;
//   the KW partial accumulators are summed through LDS by wave 0 after one __syncthreads().
// With N >= ~8 k columns that is already thousands of waves; for narrower matrices K is additionally split over
// blockIdx.y into fp32 slabs (splitk_reduce_kernel sums them).
#include "launch.h"
#include "w4_mma.h"

namespace ql {

template <typename T, int KW, int NT, int MT>
__global__ __launch_bounds__(KW * 64) void w4_fewrow_kernel(const T* __restrict__ A, const u32x4* __restrict__ Wm,
                                                            const T* __restrict__ Sm, int M, int N, int K, int G, int64_t lda,
                                                            int per, const T* __restrict__ bias, T* __restrict__ C,
                                                            int64_t ldc, float* __restrict__ part, int gate) {
    // (argument order: the leading 14 dwords - what the first loads need - are preloaded into SGPRs at wave launch)
    typedef Mma<T> MM;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // KW x 2 x 4 KB A tiles; reused for the reduction
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kb = lane >> 5;
    const int n_base = blockIdx.x * (32 * NT) + j;            // + 32 t for the wave's t-th column tile
    const int ksteps = (G + 1) >> 1;
    // this wave's K steps: [k0, k0 + nst)
    const int slice = blockIdx.y * KW + wave;
    const int k0 = slice * per;
    const int nst = k0 >= ksteps ? 0 : (ksteps - k0 < per ? ksteps - k0 : per);

    u32 k_mask_lo, k_mask_hi, k_magic;
    asm volatile("s_mov_b32 %0, 0x000F000F" : "=s"(k_mask_lo));
    asm volatile("s_mov_b32 %0, 0x00F000F0" : "=s"(k_mask_hi));
    asm volatile("v_mov_b32 %0, %1" : "=v"(k_magic) : "i"(MM::kMagic));

    // tile-major part of the derived layout (launch.h): this wave's units of a step are 1 KB contiguous
    const int ctiles = (N + 31) >> 5;
    const u32x4* wtile[NT];
    const T* stile[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ct = blockIdx.x * NT + t < ctiles ? blockIdx.x * NT + t : ctiles - 1;
        wtile[t] = Wm + (int64_t)ct * ksteps * 64 + lane;
        stile[t] = Sm + (int64_t)ct * ksteps * 64 + lane;
    }
    constexpr int ATILE = 4096 * MT;                          // one A tile: 32 MT rows x 128 bytes
    char* abuf = smem + wave * (2 * ATILE);                   // two tiles, private to the wave

    // A staging by the wave: chunk q = lane + 64 u -> row q >> 3 (8 u .. 8 u + 7), 16-byte column q & 7
    const T* a_src[4 * MT];
    int a_dst[4 * MT];
#pragma unroll
    for (int u = 0; u < 4 * MT; ++u) {
        const int q = lane + 64 * u, r = q >> 3, c = q & 7;
        a_src[u] = A + (int64_t)(r < M ? r : M - 1) * lda + c * 8;
        a_dst[u] = (r * 8 + (c ^ ((r >> 1) & 7))) * 16;
    }
    const int c_mine = lane & 7;
    const int kmax = K - 8;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.f;

    typedef decltype(MM::scale_pair(stile[0], true)) scale_t;
    struct Stage {
        u32x4 a[4 * MT];
        u32x4 w[NT];
        T s[NT];
    };
    auto load_stage = [&](int kt, Stage& sg) {
        const int k = kt * 64 + c_mine * 8;
#pragma unroll
        for (int u = 0; u < 4 * MT; ++u)
            if (8 * u < M)                                     // rows past M are never loaded (block-uniform test)
                sg.a[u] = *reinterpret_cast<const u32x4*>(a_src[u] + (k <= kmax ? kt * 64 : kmax - c_mine * 8));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            sg.w[t] = __builtin_nontemporal_load(wtile[t] + (int64_t)kt * 64);   // streamed once, whole lines
            sg.s[t] = stile[t][(int64_t)kt * 64];
        }
    };
    auto store_a = [&](int buf, const Stage& sg) {
#pragma unroll
        for (int u = 0; u < 4 * MT; ++u)
            if (8 * u < M) *reinterpret_cast<u32x4*>(abuf + buf * ATILE + a_dst[u]) = sg.a[u];
    };
    auto mma_step = [&](int buf, int kt, const u32x4 (&w)[NT], const T (&s_raw)[NT]) {
        const int g = 2 * kt + kb;
        scale_t s[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const T s_eff = g < G ? s_raw[t] : (T)0.f;       // the missing half of an odd last step contributes 0
            s[t] = MM::scale_pair(&s_eff, true);
        }
        const char* sa = abuf + buf * ATILE;
        u32x4 fa[2][MT];
        typename MM::frag fb[2][NT];
        auto read_a = [&](int sub, u32x4 (&fr)[MT]) {
            const int c = kb * 4 + sub;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r = mt * 32 + j;
                fr[mt] = *reinterpret_cast<const u32x4*>(sa + (r * 8 + (c ^ ((r >> 1) & 7))) * 16);
            }
        };
        read_a(0, fa[0]);
#pragma unroll
        for (int t = 0; t < NT; ++t) fb[0][t] = MM::dequant(w[t][0], k_mask_lo, k_mask_hi, k_magic, s[t]);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            if (sub < 3) {
                read_a(sub + 1, fa[(sub + 1) & 1]);
#pragma unroll
                for (int t = 0; t < NT; ++t) fb[(sub + 1) & 1][t] = MM::dequant(w[t][sub + 1], k_mask_lo, k_mask_hi, k_magic, s[t]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[mt][t] = MM::mma(__builtin_bit_cast(typename MM::frag, fa[sub & 1][mt]), fb[sub & 1][t], acc[mt][t]);
        }
    };

    // bias of the lane's column(s): requested behind the first weight loads (its pointer is a late kernel argument) and
    // not in the epilogue, where it would be a global round trip in the wave's tail
    float bias_t[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias_t[t] = 0.f;
    auto load_bias = [&]() {                                  // unconditional (a load under `if (bias)` drains the queue)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n_base + 32 * t;
            bias_t[t] = Act<T>::load((bias ? bias : Sm) + (bias && n < N ? n : 0));
        }
    };
    if (nst > 0) {
        // two steps in flight; loads unconditional (a step past the end re-reads the last one: never stored)
        Stage st0, st1;
        load_stage(k0, st0);
        load_stage(k0 + (nst > 1 ? 1 : 0), st1);
        load_bias();
        int t = 0;
        for (; t + 2 < nst; t += 2) {
            store_a(0, st0);
            {
                u32x4 w[NT];
                T sc[NT];
#pragma unroll
                for (int q = 0; q < NT; ++q) { w[q] = st0.w[q]; sc[q] = st0.s[q]; }
                load_stage(k0 + t + 2, st0);
                mma_step(0, k0 + t, w, sc);
            }
            store_a(1, st1);
            {
                u32x4 w[NT];
                T sc[NT];
#pragma unroll
                for (int q = 0; q < NT; ++q) { w[q] = st1.w[q]; sc[q] = st1.s[q]; }
                load_stage(k0 + (t + 3 < nst ? t + 3 : nst - 1), st1);
                mma_step(1, k0 + t + 1, w, sc);
            }
        }
        store_a(0, st0);
        mma_step(0, k0 + t, st0.w, st0.s);
        if (t + 1 < nst) {
            store_a(1, st1);
            mma_step(1, k0 + t + 1, st1.w, st1.s);
        }
    }

    if (nst == 0) load_bias();                                // (a wave without K steps still writes its tile when it is wave 0)

    // sum the KW slices: every wave parks its accumulator in LDS, wave 0 adds them up and writes the 32 x 32 tile
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if (wave > 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) red[((((wave - 1) * MT + mt) * NT + t) * 16 + i) * 64 + lane] = acc[mt][t][i];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < KW; ++w)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][t][i] += red[((((w - 1) * MT + mt) * NT + t) * 16 + i) * 64 + lane];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n_raw = n_base + 32 * t;
        if (gate) {
            // SiLU * gate epilogue (chatglm_q/model.py:200-201) on gate-interleaved columns: lanes j % 4 = 0, 1 hold
            // h, lanes 2, 3 the gate of the same output pair; C gets N / 2 columns.  N % 32 == 0, no K slabs.
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float y = Act<T>::round(acc[mt][t][i]);
                    if (bias) y = Act<T>::round(y + bias_t[t]);
                    const float yg = __shfl_down(y, 2);       // every lane takes part
                    const int m = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
                    if (m < M && (j & 2) == 0)
                        Act<T>::store(C + (int64_t)m * ldc + ((n_raw >> 2) << 1) + (j & 1),
                                      Act<T>::round(Act<T>::round(y / (1.0f + __expf(-y))) * yg));
                }
            continue;
        }
        if (n_raw >= N) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
                if (m >= M) continue;
                if (part) {
                    part[((int64_t)blockIdx.y * M + m) * N + n_raw] = acc[mt][t][i];
                } else {                                      // store_out's sequence with the preloaded bias
                    float y = Act<T>::round(acc[mt][t][i]);
                    if (bias) y = y + bias_t[t];
                    Act<T>::store(C + (int64_t)m * ldc + n_raw, y);
                }
            }
    }
}

// column tiles per wave: the kernel template supports 2 (wide matrices), but measured against the tiled split-K GEMM
// this kernel only wins for narrow ones (see w4_fewrow_supported), where 1 gives more waves
static int fewrow_nt(int64_t) { return 1; }

struct FewRowPlan {
    int ksplit, per;
};
constexpr int kFewRowWaves = 4;
// global K split: aim at >= ~2048 waves (8 per CU), at least 2 K steps per wave
static int fewrow_waves() { return QL_TUNE("QLINEAR_FEWROW_KW", kFewRowWaves); }   // developer build: 4 / 8 / 16 waves per block
static FewRowPlan fewrow_plan(int64_t M, int64_t N, int64_t K, size_t ws_bytes) {
    const int kFewRowWaves = fewrow_waves();
    const int forced = QL_TUNE("QLINEAR_FEWROW_KSPLIT", 0);
    const int nt = fewrow_nt(N);
    const int64_t ksteps = (K / 32 + 1) / 2, nb = (N + 32 * nt - 1) / (32 * nt);
    int64_t ks = forced > 0 ? forced : (2048 + nb * kFewRowWaves - 1) / (nb * kFewRowWaves);
    if (ks > ksteps / (2 * kFewRowWaves)) ks = ksteps / (2 * kFewRowWaves);
    while (ks > 1 && (size_t)(ks * M * N) * sizeof(float) > ws_bytes) --ks;
    if (ks < 1) ks = 1;
    const int64_t per = (ksteps + ks * kFewRowWaves - 1) / (ks * kFewRowWaves);
    return {(int)ks, (int)per};
}

// Measured against w4_packed_gemm (one 32-row tile + split-K over workgroups, fragments fetched per column), ChatGLM2-6B
// shapes, fp16, us at 8 rows: 4096->4608 9.5 vs 12.8, 4096->4096 8.7 vs 11.6, 4096->27392 21.0 vs 27.9,
// 13696->4096 14.3 vs 20.6 (the GEMV at ONE row: 5.3 / 4.7 / 14.1 / 10.2).  QLINEAR_DISPATCH=nofewrow disables it.
bool w4_fewrow_supported(int64_t M, int64_t N, int64_t K) {
    return !(dispatch_flags() & QL_D_NOFEWROW) && M <= 32 && K >= 512;       // two row tiles (33..64 rows) measured no better than the tiled GEMM
}

size_t w4_fewrow_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    const FewRowPlan p = fewrow_plan(M, N, K, (size_t)-1);
    return p.ksplit > 1 ? (size_t)(p.ksplit * M * N) * sizeof(float) : 0;
}

template <typename T, int NT, int MT, int KW = kFewRowWaves>
static int launch_fewrow_nt(const void* A, const void* tiled, const void* bias, void* C, int M, int N, int K, int64_t lda,
                            int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st, bool gate) {
    const FewRowPlan plan = fewrow_plan(M, N, K, ws && ((uintptr_t)ws & 15) == 0 ? ws_bytes : 0);
    const W4Layout L = w4_layout(N, K, sizeof(T));
    const int64_t G = L.G;
    const u32x4* Wt = (const u32x4*)tiled;
    const T* Sp = (const T*)((const char*)tiled + (L.off_sm - L.off_wm));
    float* part = plan.ksplit > 1 ? (float*)ws : nullptr;
    if (gate && (part || N % 32 != 0)) return QL_ERR_UNSUPPORTED;   // the gate epilogue lives in this kernel only
    dim3 grid((unsigned)((N + 32 * NT - 1) / (32 * NT)), (unsigned)plan.ksplit);
    constexpr size_t lds = (size_t)KW * 8192 * MT;
    static_assert((KW - 1) * MT * NT * 16 * 64 * 4 <= KW * 8192 * MT, "reduction scratch fits in the A buffers");
    if constexpr (lds > 65536) {
        static bool attr = [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(&w4_fewrow_kernel<T, KW, NT, MT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds) == hipSuccess;
        }();
        (void)attr;
    }
    w4_fewrow_kernel<T, KW, NT, MT><<<grid, KW * 64, lds, st>>>((const T*)A, Wt, Sp, M, N, K, (int)G, lda, plan.per,
                                                            (const T*)bias, (T*)C, ldc, part, gate ? 1 : 0);
    const int rc = finish_launch(QL_K_W4_FEWROW);
    if (rc != 0 || !part) return rc;
    const int64_t total = (int64_t)M * N;
    splitk_reduce_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(part, (const T*)bias, (T*)C, M, N, ldc, plan.ksplit);
    return finish_launch(QL_K_SPLITK_REDUCE);
}

template <typename T>
static int launch_fewrow(const void* A, const void* tiled, const void* bias, void* C, int M, int N, int K, int64_t lda,
                         int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st, bool gate) {
#ifdef QL_DEV_TUNING
    if (fewrow_waves() == 16) return launch_fewrow_nt<T, 1, 1, 16>(A, tiled, bias, C, M, N, K, lda, ldc, ws, ws_bytes, st, gate);
    if (fewrow_waves() == 8) return launch_fewrow_nt<T, 1, 1, 8>(A, tiled, bias, C, M, N, K, lda, ldc, ws, ws_bytes, st, gate);
#endif
    return launch_fewrow_nt<T, 1, 1>(A, tiled, bias, C, M, N, K, lda, ldc, ws, ws_bytes, st, gate);
}

int w4_fewrow(int dtype, const void* A, const void* tiled, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
              int64_t lda, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st, bool gate) {
    switch (dtype) {
    case QL_DTYPE_F16: return launch_fewrow<f16>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, ws, ws_bytes, st, gate);
    case QL_DTYPE_BF16: return launch_fewrow<__bf16>(A, tiled, bias, C, (int)M, (int)N, (int)K, lda, ldc, ws, ws_bytes, st, gate);
    default: return QL_ERR_BAD_DTYPE;
    }
}

}  // namespace ql
