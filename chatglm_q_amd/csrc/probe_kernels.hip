// Floor probes of bench.py's roofline leg (round 5) - linked into libqlinear_hip_span.so ONLY (the probe build of the library; the
// product library does not contain them).  They answer, on the box the bench runs on and under the bench's own launch protocol, what
// one DEPENDENT launch of the headline kernel's size can cost at the least:
//   qlinear_probe_empty   an empty kernel on the headline GEMV's grid: the launch boundary alone
//   qlinear_probe_read    a pure streaming read of `bytes` (coalesced 1 KB per wave instruction, 8 loads in flight per thread, XOR,
//                         one 4-byte store per wave): launch boundary + one HBM round trip + the bytes, with no arithmetic and no staging
//   qlinear_probe_copy    a 16-byte-per-thread device copy of `bytes`: the sustained copy rate this chip reaches (the "ceiling" the
//                         guide quotes as 6.29 TB/s), measured instead of quoted
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe_empty_kernel() {}

template <int LOADS>
__global__ __launch_bounds__(256) void probe_read_kernel(const u32x4* __restrict__ src, uint32_t* __restrict__ out, size_t n16) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t base = wave * (size_t)LOADS * 64 + lane;
    u32x4 v[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        size_t idx = base + (size_t)i * 64;
        if (idx >= n16) idx = lane;
        v[i] = __builtin_nontemporal_load(src + idx);
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) x ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x ^= __shfl_xor(x, off, 64);
    if (lane == 0) out[wave] = x;
}

// The floor as a SWEEP (round 6, VERDICT r5 weak 3: "the floor is one probe shape"): the same pure read with LOADS 16-byte loads in flight
// per lane, any workgroup count (a wave owns a contiguous span and walks it in rounds of LOADS x 1 KB), default or non-temporal loads,
// or LDS-DMA (global_load_lds_dwordx4: no VGPR round trip; the 1 KB per wave instruction lands in a 16 KB-per-wave LDS ring).
// bench.py reports the minimum over the shapes.
template <int LOADS, int MODE>   // MODE 0: default loads, 1: non-temporal, 2: LDS-DMA
__global__ __launch_bounds__(256) void probe_read_sweep_kernel(const u32x4* __restrict__ src, uint32_t* __restrict__ out, size_t n16) {
    __shared__ u32x4 ring[MODE == 2 ? 4 * 16 * 64 : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 4 + wv, nwaves = (size_t)gridDim.x * 4;
    const size_t rounds_total = (n16 + 63) / 64;                        // 1 KB wave-rounds in the region
    const size_t r0 = rounds_total * wave / nwaves, r1 = rounds_total * (wave + 1) / nwaves;
    uint32_t x = 0;
    for (size_t r = r0; r < r1; r += LOADS) {
        if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                size_t idx = (r + i < r1 ? r + i : r0) * 64 + lane;
                if (idx >= n16) idx = lane;
                const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(&ring[(wv * 16 + (i & 15)) * 64]));
                const unsigned long long b0 = (unsigned long long)(uintptr_t)src;
                const unsigned long long base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b0 >> 32)) << 32) |
                                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b0);
                const unsigned voff = (unsigned)(idx * 16);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(base) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            x ^= ring[(wv * 16) * 64 + lane][0];
        } else {
            u32x4 v[LOADS];
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                size_t idx = (r + i < r1 ? r + i : r0) * 64 + lane;
                if (idx >= n16) idx = lane;
                v[i] = MODE == 1 ? __builtin_nontemporal_load(src + idx) : src[idx];
            }
#pragma unroll
            for (int i = 0; i < LOADS; ++i) x ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x ^= __shfl_xor(x, off, 64);
    if (lane == 0) out[wave] = x;
}

// variant 0: grid-stride, one 16-byte unit per thread and trip; variant 1: 4 units in flight per thread, non-temporal both ways
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void probe_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

}  // namespace

extern "C" int qlinear_probe_empty(int blocks, void* stream) {
    probe_empty_kernel<<<blocks > 0 ? blocks : 1, 256, 0, (hipStream_t)stream>>>();
    return (int)hipGetLastError();
}

// `out` holds at least qlinear_probe_read_waves(bytes) 32-bit words
extern "C" int64_t qlinear_probe_read_waves(int64_t bytes) {
    const int64_t n16 = bytes / 16, per_block = 8 * 256;
    return ((n16 + per_block - 1) / per_block) * 4;
}
extern "C" int qlinear_probe_read(const void* src, int64_t bytes, void* out, void* stream) {
    const int64_t n16 = bytes / 16, per_block = 8 * 256;
    if (!src || !out || n16 <= 0) return -1;
    probe_read_kernel<8><<<(unsigned)((n16 + per_block - 1) / per_block), 256, 0, (hipStream_t)stream>>>((const u32x4*)src, (uint32_t*)out, (size_t)n16);
    return (int)hipGetLastError();
}

// loads in {2, 4, 8, 9, 16}, blocks >= 1 (x 256 threads), mode 0 default / 1 non-temporal / 2 LDS-DMA; `out` holds >= 4 * blocks words
extern "C" int qlinear_probe_read_sweep(const void* src, int64_t bytes, void* out, int loads, int blocks, int mode, void* stream) {
    const int64_t n16 = bytes / 16;
    if (!src || !out || n16 <= 0 || blocks < 1 || mode < 0 || mode > 2 || (n16 * 16) >> 32) return -1;
    hipStream_t st = (hipStream_t)stream;
#define QL_PROBE(L)                                                                                                          \
    case L:                                                                                                                  \
        if (mode == 0) probe_read_sweep_kernel<L, 0><<<blocks, 256, 0, st>>>((const u32x4*)src, (uint32_t*)out, (size_t)n16);      \
        else if (mode == 1) probe_read_sweep_kernel<L, 1><<<blocks, 256, 0, st>>>((const u32x4*)src, (uint32_t*)out, (size_t)n16); \
        else probe_read_sweep_kernel<L, 2><<<blocks, 256, 0, st>>>((const u32x4*)src, (uint32_t*)out, (size_t)n16);                \
        break
    switch (loads) {
        QL_PROBE(2); QL_PROBE(4); QL_PROBE(8); QL_PROBE(9); QL_PROBE(16);
    default: return -1;
    }
#undef QL_PROBE
    return (int)hipGetLastError();
}

// variant: 0 .. 3 (the bench reports the best: the ceiling is what the chip can do, not what one loop shape does)
extern "C" int qlinear_probe_copy(void* dst, const void* src, int64_t bytes, int variant, void* stream) {
    const int64_t n16 = bytes / 16;
    if (!src || !dst || n16 <= 0) return -1;
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
    case 0: probe_copy_kernel<1, false><<<256 * 16, 256, 0, st>>>((const u32x4*)src, (u32x4*)dst, (size_t)n16); break;
    case 1: probe_copy_kernel<4, true><<<256 * 8, 256, 0, st>>>((const u32x4*)src, (u32x4*)dst, (size_t)n16); break;
    case 2: probe_copy_kernel<4, false><<<256 * 8, 256, 0, st>>>((const u32x4*)src, (u32x4*)dst, (size_t)n16); break;
    case 3: probe_copy_kernel<8, true><<<256 * 4, 256, 0, st>>>((const u32x4*)src, (u32x4*)dst, (size_t)n16); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
