// Micro-benchmark harness (developer tool, not part of the product): times kernel variants on
// rotating buffers with a captured hipGraph, so sub-3-us kernels are not host-bound.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 probe.hip -L../../chatglm_q_amd/csrc -lqlinear_hip -o probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <functional>
#include "../../include/qlinear_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// pure streaming read: each thread reads LOADS x 16 B (coalesced 1 KiB per wave instruction), XORs, one store per wave
template <int LOADS, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ src, uint32_t* __restrict__ out, size_t n16) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u32x4* p = src + wave * (size_t)LOADS * 64 + lane;
    u32x4 v[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        size_t idx = (size_t)(p - src) + (size_t)i * 64;
        if (idx >= n16) idx = lane;
        v[i] = NT ? __builtin_nontemporal_load(src + idx) : src[idx];
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) x ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x ^= __shfl_xor(x, off, 64);
    if (lane == 0) out[wave] = x;
}

// strided read: each wave reads NCH chunks of 1 KiB, chunk c at byte offset wave_base + c * stride16*16;
// wave bases tile the buffer so that all bytes are read exactly once when stride == NCH-th of a block
template <int NCH>
__global__ __launch_bounds__(256) void stride_read_kernel(const u32x4* __restrict__ src, uint32_t* __restrict__ out,
                                                          size_t n16, int stride16, int chunks_per_stream) {
    // stream layout: NCH streams of `chunks_per_stream` KiB each, streams `stride16` apart, forming one block
    // of NCH * stride16 units; wave w handles chunk (w % chunks_per_stream) of block (w / chunks_per_stream)
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t block = wave / chunks_per_stream, chunk = wave % chunks_per_stream;
    const size_t base = block * (size_t)NCH * stride16 + chunk * 64 + lane;
    u32x4 v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        size_t idx = base + (size_t)c * stride16;
        if (idx >= n16) idx = lane;
        v[c] = __builtin_nontemporal_load(src + idx);
    }
    uint32_t x = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) x ^= v[c][0] ^ v[c][1] ^ v[c][2] ^ v[c][3];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x ^= __shfl_xor(x, off, 64);
    if (lane == 0) out[wave] = x;
}

// the one-row GEMV's load pattern without any math: block = 4/KS column quads x KS K-slices, a wave issues ITERS tiles
// of (4 columns x 1 KiB + 512 B of scales); STAGE: also stage K halves of x into LDS first (as the GEMV does)
template <int KS, bool STAGE, bool SCALES>
__global__ __launch_bounds__(256) void gemv_pattern_kernel(const u32x4* __restrict__ Wt, const uint2* __restrict__ Sp,
                                                           const u32x4* __restrict__ X, uint32_t* __restrict__ out, int G,
                                                           int quads, int xchunks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = wave % KS, QW = 4 / KS;
    int t = blockIdx.x * QW + wave / KS;
    if (t >= quads) t = 0;
    const int gs = (G + KS - 1) / KS, g0 = ks * gs, g1 = min(G, g0 + gs);
    const int iters = (g1 - g0 + 63) >> 6;
    uint32_t x = 0;
    if (STAGE) {
        for (int c = tid; c < xchunks; c += 256) reinterpret_cast<u32x4*>(smem)[c] = X[c];
    }
    const u32x4* wb = Wt + (size_t)t * 4 * G;
    const uint2* sb = Sp + (size_t)t * G;
    for (int it = 0; it < iters; it += 2) {
        u32x4 v[8];
        uint2 sc[2] = {{0, 0}, {0, 0}};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int g = g0 + (it + u) * 64 + lane;
            if (g >= g1 || it + u >= iters) g = G - 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[u * 4 + c] = __builtin_nontemporal_load(wb + (size_t)c * G + g);
            if (SCALES) sc[u] = sb[g];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) x ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
        x ^= sc[0].x ^ sc[0].y ^ sc[1].x ^ sc[1].y;
    }
    if (STAGE) {
        __syncthreads();
        x ^= reinterpret_cast<uint32_t*>(smem)[tid];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x ^= __shfl_xor(x, off, 64);
    if (lane == 0) out[blockIdx.x * 4 + wave] = x;
}

// Hand-off feasibility: workgroups 0..P-1 are "producers" (a dependent chain of `spin` s_sleep rounds standing in for the
// attention, then 8 KB of output, release fence, flag = tag); the others are GEMV-pattern consumers that request their
// weights at once, wait for the P flags, acquire, read the 8 KB and finish.  tag must differ between launches.
__global__ __launch_bounds__(256) void handoff_kernel(const u32x4* __restrict__ Wt, const uint2* __restrict__ Sp, u32x4* xbuf,
                                                      uint32_t* flags, const uint32_t* __restrict__ tagp, uint32_t* __restrict__ out,
                                                      int G, int quads, int P, int spin, int delay) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tag = tagp[0];
    if ((int)blockIdx.x < P) {
        for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
        for (int c = tid; c < 512; c += 256) xbuf[c] = u32x4{tag, (uint32_t)c, 1u, 2u};
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + blockIdx.x, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int blk = (int)blockIdx.x - P;
    int t = blk * 4 + wave;
    if (t >= quads) t = 0;
    for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(8);
    const u32x4* wb = Wt + (size_t)t * 4 * G;
    const uint2* sb = Sp + (size_t)t * G;
    u32x4 v[8];
    uint2 sc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int g = u * 64 + lane;
        if (g >= G) g = G - 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[u * 4 + c] = __builtin_nontemporal_load(wb + (size_t)c * G + g);
        sc[u] = sb[g];
    }
    // wait for the producers (one lane polls, bounded)
    if (tid == 0) {
        for (int p = 0; p < P; ++p) {
            int guard = 0;
            while (__hip_atomic_load(flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag && ++guard < (1 << 20))
                __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    uint32_t x = 0;
    for (int c = tid; c < 512; c += 256) {
        const u32x4 a = xbuf[c];
        x ^= a[0] ^ a[1];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) x ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    x ^= sc[0].x ^ sc[1].y;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x ^= __shfl_xor(x, off, 64);
    if (lane == 0) out[blk * 4 + wave] = x;
}
__global__ void bump_kernel(uint32_t* tagp) { tagp[0] += 1; }
__global__ __launch_bounds__(256) void chain_kernel(u32x4* xbuf, const uint32_t* tagp, int spin) {
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
    for (int c = threadIdx.x; c < 512; c += 256) xbuf[c] = u32x4{tagp[0], (uint32_t)c, 1u, 2u};
}

__global__ void empty_kernel() {}

struct Timer {
    hipStream_t st;
    hipEvent_t e0, e1;
    Timer() { CK(hipStreamCreate(&st)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
    // returns us per step; fn(i) enqueues step i on st
    double run(int steps, const std::function<void(int)>& fn, bool graph = true, int reps = 5) {
        for (int i = 0; i < 64 && i < steps; ++i) fn(i);
        CK(hipStreamSynchronize(st));
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        if (graph) {
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int i = 0; i < steps; ++i) fn(i);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        }
        double best = 1e30;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, st));
            if (graph) CK(hipGraphLaunch(ge, st)); else for (int i = 0; i < steps; ++i) fn(i);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        if (ge) CK(hipGraphExecDestroy(ge));
        if (g) CK(hipGraphDestroy(g));
        return best * 1e3 / steps;
    }
};

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 4096;
    const int N = argc > 2 ? atoi(argv[2]) : 4096;
    const int M = argc > 3 ? atoi(argv[3]) : 1;
    const bool quick = argc > 4 && !strcmp(argv[4], "gemv");
    const int FLAGS = getenv("QL_STRICT") ? QL_FLAG_STRICT_ROUNDING : 0;
    const int G = K / 32;
    const size_t per = (size_t)K * N / 2 + (size_t)G * N * 2;
    int SETS = (int)(720e6 / per) + 1; if (SETS > 96) SETS = 96; if (SETS < 3) SETS = 3;
    const int STEPS = SETS * (per > 40e6 ? 4 : 12);
    printf("K=%d N=%d M=%d  bytes/step(weights+scales)=%zu  sets=%d steps=%d\n", K, N, M, per, SETS, STEPS);
    Timer T;
    // buffers
    std::vector<uint8_t*> wq(SETS); std::vector<void*> sc(SETS), pk(SETS);
    const size_t pkb = qlinear_w4g32_packed_bytes(N, K, 32, QL_DTYPE_F16);
    std::vector<uint8_t> hw((size_t)K / 2 * N); std::vector<uint16_t> hs((size_t)G * N);
    for (auto& b : hw) b = (uint8_t)rand();
    for (auto& s : hs) s = 0x1C00 + (rand() & 0x3FF);   // fp16 in [0.0039, 0.0078)
    for (int i = 0; i < SETS; ++i) {
        CK(hipMalloc(&wq[i], hw.size())); CK(hipMalloc(&sc[i], hs.size() * 2)); CK(hipMalloc(&pk[i], pkb));
        CK(hipMemcpy(wq[i], hw.data(), hw.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(sc[i], hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
        int rc = qlinear_w4g32_repack(wq[i], sc[i], pk[i], N, K, 32, QL_DTYPE_F16, T.st);
        if (rc) { printf("repack rc=%d\n", rc); return 1; }
    }
    void *A, *C, *ws; uint32_t* out;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMemset(A, 0x3c, (size_t)M * K * 2));
    CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&out, 1 << 20));
    const size_t wsb = qlinear_workspace_bytes(QL_OP_W4G32_FWD, M, N, K, 32);
    CK(hipMalloc(&ws, wsb ? wsb : 16));
    CK(hipStreamSynchronize(T.st));

    const double alg = (double)per + (double)M * K * 2 + (double)M * N * 2;
    auto report = [&](const char* name, double us, double bytes) {
        printf("%-44s %8.3f us/step  %8.1f GB/s  (%.1f%% of 8 TB/s)\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0);
    };
    double us;
    if (quick) {
        us = T.run(STEPS, [&](int i) { qlinear_w4g32_fwd_packed(A, pk[i % SETS], nullptr, C, M, N, K, 32, K, N, QL_DTYPE_F16, FLAGS, nullptr, 0, T.st); });
        char nm[64]; snprintf(nm, sizeof nm, "w4 packed gemv QL_VARIANT=%s", getenv("QL_VARIANT") ? getenv("QL_VARIANT") : "0");
        report(nm, us, alg);
        return 0;
    }
    if (argc > 4 && !strcmp(argv[4], "handoff")) {
        const size_t wbytes = (size_t)N * G * 16;
        const int quads = N / 4, nblk = (quads + 3) / 4, P = 2;
        u32x4* xbuf; uint32_t *flags, *tagp;
        CK(hipMalloc(&xbuf, 8192)); CK(hipMalloc(&flags, 64)); CK(hipMalloc(&tagp, 4));
        CK(hipMemset(flags, 0, 64)); CK(hipMemset(tagp, 0, 4));
        for (int spin : {0, 8, 16, 24}) {
            // separate launches: chain kernel (2 workgroups) then the GEMV pattern (KS = 1, reads the 8 KB)
            us = T.run(STEPS, [&](int i) {
                chain_kernel<<<P, 256, 0, T.st>>>(xbuf, tagp, spin);
                gemv_pattern_kernel<1, true, true><<<nblk, 256, K * 2, T.st>>>((const u32x4*)pk[i % SETS], (const uint2*)((const char*)pk[i % SETS] + wbytes),
                                                                              (const u32x4*)xbuf, out, G, quads, K / 8);
            });
            printf("spin %2d: separate launches (chain + pattern)      %8.3f us\n", spin, us);
            us = T.run(STEPS, [&](int) { chain_kernel<<<P, 256, 0, T.st>>>(xbuf, tagp, spin); });
            printf("spin %2d: chain kernel alone                        %8.3f us\n", spin, us);
            for (int delay : {0, 4, 8}) {
                us = T.run(STEPS, [&](int i) {
                    bump_kernel<<<1, 1, 0, T.st>>>(tagp);
                    handoff_kernel<<<P + nblk, 256, 0, T.st>>>((const u32x4*)pk[i % SETS], (const uint2*)((const char*)pk[i % SETS] + wbytes), xbuf, flags,
                                                                tagp, out, G, quads, P, spin, delay);
                });
                printf("spin %2d: one launch with hand-off (+ bump kernel), consumer delay %d  %8.3f us\n", spin, delay, us);
            }
        }
        us = T.run(STEPS, [&](int) { bump_kernel<<<1, 1, 0, T.st>>>(tagp); });
        printf("bump kernel alone %8.3f us\n", us);
        return 0;
    }
    if (argc > 4 && !strcmp(argv[4], "pattern")) {
        // bytes the GEMV reads: part 1 of the derived layout (weights + scales)
        const size_t wbytes = (size_t)N * G * 16, sbytes = (size_t)N * G * 2;
        const int quads = N / 4;
        const int lds = K * 2;
        auto run_pat = [&](auto kern, int ks, int ldsb, const char* name) {
            const int blocks = (quads + 4 / ks - 1) / (4 / ks);
            double t = T.run(STEPS, [&](int i) {
                kern<<<blocks, 256, ldsb, T.st>>>((const u32x4*)pk[i % SETS], (const uint2*)((const char*)pk[i % SETS] + wbytes),
                                                  (const u32x4*)A, out, G, quads, K / 8);
            });
            report(name, t, (double)(wbytes + sbytes));
        };
        {
            const size_t n16 = (wbytes + sbytes) / 16;
            const int waves = (int)((n16 + 9 * 64 - 1) / (9 * 64));
            us = T.run(STEPS, [&](int i) { read_kernel<9, true><<<(waves + 3) / 4, 256, 0, T.st>>>((const u32x4*)pk[i % SETS], out, n16); });
            report("pure read nt of part 1, 9x16B/thread", us, (double)(wbytes + sbytes));
        }
        run_pat(gemv_pattern_kernel<1, false, false>, 1, 0, "pattern KS=1 weights only");
        run_pat(gemv_pattern_kernel<2, false, false>, 2, 0, "pattern KS=2 weights only");
        run_pat(gemv_pattern_kernel<4, false, false>, 4, 0, "pattern KS=4 weights only");
        run_pat(gemv_pattern_kernel<4, false, true>, 4, 0, "pattern KS=4 weights + scales");
        run_pat(gemv_pattern_kernel<4, false, true>, 4, lds, "pattern KS=4 weights + scales, LDS allocated");
        run_pat(gemv_pattern_kernel<4, true, true>, 4, lds, "pattern KS=4 weights + scales + x staging");
        run_pat(gemv_pattern_kernel<2, true, true>, 2, lds, "pattern KS=2 weights + scales + x staging");
        run_pat(gemv_pattern_kernel<1, true, true>, 1, lds, "pattern KS=1 weights + scales + x staging");
        us = T.run(STEPS, [&](int i) { qlinear_w4g32_fwd_packed(A, pk[i % SETS], nullptr, C, M, N, K, 32, K, N, QL_DTYPE_F16, FLAGS, nullptr, 0, T.st); });
        report("w4 packed gemv (rotating)", us, alg);
        return 0;
    }
    us = T.run(STEPS, [&](int) { empty_kernel<<<1, 64, 0, T.st>>>(); });
    report("empty kernel (launch boundary)", us, 0);
    {
        const size_t n16 = pkb / 16;
        const int waves = (int)((n16 + 9 * 64 - 1) / (9 * 64));
        const int blocks = (waves + 3) / 4;
        us = T.run(STEPS, [&](int i) { read_kernel<9, true><<<blocks, 256, 0, T.st>>>((const u32x4*)pk[i % SETS], out, n16); });
        report("pure read nt, 9x16B/thread", us, (double)pkb);
        us = T.run(STEPS, [&](int i) { read_kernel<9, false><<<blocks, 256, 0, T.st>>>((const u32x4*)pk[i % SETS], out, n16); });
        report("pure read,    9x16B/thread", us, (double)pkb);
        const int waves2 = (int)((n16 + 18 * 64 - 1) / (18 * 64));
        us = T.run(STEPS, [&](int i) { read_kernel<18, true><<<(waves2 + 3) / 4, 256, 0, T.st>>>((const u32x4*)pk[i % SETS], out, n16); });
        report("pure read nt, 18x16B/thread", us, (double)pkb);
        const int waves3 = (int)((n16 + 4 * 64 - 1) / (4 * 64));
        us = T.run(STEPS, [&](int i) { read_kernel<4, true><<<(waves3 + 3) / 4, 256, 0, T.st>>>((const u32x4*)pk[i % SETS], out, n16); });
        report("pure read nt, 4x16B/thread", us, (double)pkb);
        us = T.run(STEPS, [&](int i) { read_kernel<9, true><<<blocks, 256, 0, T.st>>>((const u32x4*)pk[0], out, n16); });
        report("pure read nt, same buffer (cache-hot)", us, (double)pkb);
        // 8 streams per wave, each stream G*16 bytes long (one column), streams one column apart: the GEMV's pattern
        {
            const int stride16 = G;                      // one column = G units of 16 B
            const int cps = (G + 63) / 64;               // 1-KiB chunks per column (last one partial -> over-read)
            const size_t nblocks = n16 / ((size_t)8 * stride16);
            const int waves = (int)(nblocks * cps);
            us = T.run(STEPS, [&](int i) { stride_read_kernel<8><<<(waves + 3) / 4, 256, 0, T.st>>>((const u32x4*)pk[i % SETS], out, n16, stride16, cps); });
            report("strided read nt: 8 cols x 1KiB per wave", us, (double)pkb);
        }
    }
    us = T.run(STEPS, [&](int i) { qlinear_w4g32_fwd_packed(A, pk[i % SETS], nullptr, C, M, N, K, 32, K, N, QL_DTYPE_F16, FLAGS, nullptr, 0, T.st); });
    report("w4 packed gemv (rotating)", us, alg);
    us = T.run(STEPS, [&](int) { qlinear_w4g32_fwd_packed(A, pk[0], nullptr, C, M, N, K, 32, K, N, QL_DTYPE_F16, FLAGS, nullptr, 0, T.st); });
    report("w4 packed gemv (same weights, cache-hot)", us, alg);
    us = T.run(STEPS, [&](int i) { qlinear_w4g32_fwd(A, wq[i % SETS], sc[i % SETS], nullptr, C, M, N, K, 32, K, N, QL_DTYPE_F16, ws, wsb, T.st); });
    report("w4 canonical split-K + reduce (rotating)", us, alg);
    us = T.run(STEPS, [&](int i) { qlinear_w4g32_fwd_packed(A, pk[i % SETS], nullptr, C, M, N, K, 32, K, N, QL_DTYPE_F16, FLAGS, nullptr, 0, T.st); }, false);
    report("w4 packed gemv (rotating, eager launches)", us, alg);
    return 0;
}
