// Quantised embedding gathers for gfx950 (QEmbedding.forward, chatglm_q/int4/qlinear.py:122-131 and
// chatglm_q/int8/qlinear.py:118-120).  Pure HBM-bound row gathers: one block per token, lanes
// along the embedding dimension.
#include "launch.h"
#include "ql_common.h"

namespace ql {

// int4: packing runs along the VOCABULARY axis: token t -> byte row t/2, nibble t%2, group t/group.
template <typename T>
__global__ __launch_bounds__(256) void qembedding_w4_kernel(const int64_t* __restrict__ ids, const uint8_t* __restrict__ Wq,
                                                            const T* __restrict__ S, T* __restrict__ out, int V, int D,
                                                            int group) {
    const int64_t t = ids[blockIdx.x];
    T* o = out + (int64_t)blockIdx.x * D;
    if (t < 0 || t >= V) {   // out-of-range id: defined output instead of a wild read
        for (int d = threadIdx.x; d < D; d += 256) Act<T>::store(o + d, 0.f);
        return;
    }
    const uint8_t* row = Wq + (t >> 1) * D;
    const T* srow = S + (t / group) * D;
    const int shift = (int)(t & 1) * 4;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float q = (float)((row[d] >> shift) & 0xF) - 8.0f;
        Act<T>::store(o + d, q * Act<T>::load(srow + d));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void qembedding_w8_kernel(const int64_t* __restrict__ ids, const int8_t* __restrict__ W,
                                                            const T* __restrict__ S, T* __restrict__ out, int V, int D) {
    const int64_t t = ids[blockIdx.x];
    T* o = out + (int64_t)blockIdx.x * D;
    if (t < 0 || t >= V) {
        for (int d = threadIdx.x; d < D; d += 256) Act<T>::store(o + d, 0.f);
        return;
    }
    const int8_t* row = W + t * D;
    for (int d = threadIdx.x; d < D; d += 256) Act<T>::store(o + d, (float)row[d] * Act<T>::load(S + d));
}

template <typename T>
static int launch_qe4(const int64_t* ids, const uint8_t* Wq, const void* S, void* out, int64_t count, int64_t V,
                      int64_t D, int64_t group, hipStream_t st) {
    qembedding_w4_kernel<T><<<(unsigned)count, 256, 0, st>>>(ids, Wq, (const T*)S, (T*)out, (int)V, (int)D, (int)group);
    return finish_launch();
}
template <typename T>
static int launch_qe8(const int64_t* ids, const int8_t* W, const void* S, void* out, int64_t count, int64_t V,
                      int64_t D, hipStream_t st) {
    qembedding_w8_kernel<T><<<(unsigned)count, 256, 0, st>>>(ids, W, (const T*)S, (T*)out, (int)V, (int)D);
    return finish_launch();
}

int qembedding_w4(int dtype, const int64_t* ids, const uint8_t* Wq, const void* S, void* out, int64_t count,
                  int64_t V, int64_t D, int64_t group, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F32: return launch_qe4<float>(ids, Wq, S, out, count, V, D, group, st);
    case QL_DTYPE_F16: return launch_qe4<f16>(ids, Wq, S, out, count, V, D, group, st);
    case QL_DTYPE_BF16: return launch_qe4<__bf16>(ids, Wq, S, out, count, V, D, group, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}
int qembedding_w8(int dtype, const int64_t* ids, const int8_t* W, const void* S, void* out, int64_t count,
                  int64_t V, int64_t D, hipStream_t st) {
    switch (dtype) {
    case QL_DTYPE_F32: return launch_qe8<float>(ids, W, S, out, count, V, D, st);
    case QL_DTYPE_F16: return launch_qe8<f16>(ids, W, S, out, count, V, D, st);
    case QL_DTYPE_BF16: return launch_qe8<__bf16>(ids, W, S, out, count, V, D, st);
    default: return QL_ERR_BAD_DTYPE;
    }
}

}  // namespace ql
