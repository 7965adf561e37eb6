/* CPU oracle, C restatement (TEST INFRASTRUCTURE ONLY - never linked into the product).
 *
 * Plain-C restatement of the reference's quantized-linear forward, used (a) as a second,
 * independent checker next to oracle/qlinear_oracle.py and (b) as the timed "port" CPU
 * baseline in bench.py.  Pinned to the reference through tests/golden (tests/test_oracle_c.py).
 *
 *   w4:  C = A . dequant(B)  with B (K/2, N) uint8, byte [k/2, n] = row 2*(k/2) in the low
 *        nibble, row 2*(k/2)+1 in the high nibble, value = (nibble - 8) * S[k/group, n],
 *        the product rounded to the activation dtype before the dot
 *        (chatglm_q/int4/qlinear.py:20-33,50; chatglm_q/int4/triton_ops.py:67-80).
 *   w8:  C = A . (W^T * S[n]) with W (N, K) int8 row-major
 *        (chatglm_q/int8/qlinear.py:38,90; chatglm_q/int8/triton_ops.py:62-73).
 *   Accumulation in double, one rounding to the activation dtype, bias added AFTER that
 *   rounding and rounded again (chatglm_q/int4/qlinear.py:90-94).
 *
 * dtype codes: 0 = f32, 1 = f16, 2 = bf16 (same as include/qlinear_hip.h).
 * Storage: f32 -> float, f16/bf16 -> uint16_t bit patterns.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float f32_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t bits_from_f32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
    if (exp == 0) {
        if (man == 0) return f32_from_bits(sign);
        float v = ldexpf((float)man, -24);
        return sign ? -v : v;
    }
    if (exp == 31) return f32_from_bits(sign | 0x7F800000u | (man << 13));
    return f32_from_bits(sign | ((exp + 112u) << 23) | (man << 13));
}

/* double -> f16, round-to-nearest-even in one step (no double rounding through float). */
static uint16_t double_to_half(double d) {
    if (isnan(d)) return 0x7E00u;
    uint16_t sign = signbit(d) ? 0x8000u : 0;
    double a = fabs(d);
    if (a >= 65520.0) return sign | 0x7C00u;                 /* rounds to inf */
    if (a < ldexp(1.0, -25)) return sign;                     /* below half the smallest subnormal */
    int e;
    (void)frexp(a, &e);                                       /* a = m * 2^e, m in [0.5, 1) */
    int exp = e - 1;                                          /* a = 1.xxx * 2^exp */
    if (exp < -14) exp = -14;                                 /* subnormal: fixed quantum 2^-24 */
    double q = ldexp(a, 10 - exp);                            /* integer part = 11-bit significand */
    double r = nearbyint(q);                                  /* default mode: ties-to-even */
    if (exp == -14 && r < 1024.0) return sign | (uint16_t)r;  /* subnormal */
    if (r >= 2048.0) { r *= 0.5; exp += 1; }
    if (exp > 15) return sign | 0x7C00u;
    return sign | (uint16_t)(((exp + 15) << 10) + ((int)r - 1024));
}

static float bf16_to_float(uint16_t b) { return f32_from_bits((uint32_t)b << 16); }

static uint16_t double_to_bf16(double d) {
    if (isnan(d)) return 0x7FC0u;
    float f = (float)d;
    uint32_t u = bits_from_f32(f);
    /* repair double rounding when the float landed exactly on a bf16 tie */
    if ((u & 0xFFFFu) == 0x8000u && isfinite(f)) {
        double err = d - (double)f;
        if (err != 0.0) {
            int away = (err > 0) == (f >= 0);
            u = away ? u + 1 : u - 1;
        }
    }
    uint32_t rounding = ((u >> 16) & 1u) + 0x7FFFu;
    return (uint16_t)((u + rounding) >> 16);
}

static double load_act(const void* p, int64_t i, int dtype) {
    switch (dtype) {
    case 0: return (double)((const float*)p)[i];
    case 1: return (double)half_to_float(((const uint16_t*)p)[i]);
    default: return (double)bf16_to_float(((const uint16_t*)p)[i]);
    }
}

/* round x to dtype, return the rounded value as double */
static double round_act(double x, int dtype) {
    switch (dtype) {
    case 0: return (double)(float)x;
    case 1: return (double)half_to_float(double_to_half(x));
    default: return (double)bf16_to_float(double_to_bf16(x));
    }
}

static void store_act(void* p, int64_t i, double x, int dtype) {
    switch (dtype) {
    case 0: ((float*)p)[i] = (float)x; break;
    case 1: ((uint16_t*)p)[i] = double_to_half(x); break;
    default: ((uint16_t*)p)[i] = double_to_bf16(x); break;
    }
}

/* Dense dequantised weight (K, N) as doubles holding dtype-rounded values. */
static double* dequant_w4(const uint8_t* B, const void* S, int64_t K, int64_t N, int64_t group, int dtype) {
    double* W = (double*)malloc(sizeof(double) * (size_t)K * (size_t)N);
    if (!W) return NULL;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < K; ++k) {
        const uint8_t* row = B + (k / 2) * N;
        int shift = (int)(k & 1) * 4;
        int64_t g = k / group;
        for (int64_t n = 0; n < N; ++n) {
            int q = (int)((row[n] >> shift) & 0xF) - 8;
            W[k * N + n] = round_act((double)q * load_act(S, g * N + n, dtype), dtype);
        }
    }
    return W;
}

static int matmul_dense(const void* A, const double* W, const void* bias, void* C,
                        int64_t M, int64_t N, int64_t K, int dtype) {
#pragma omp parallel for schedule(static)
    for (int64_t n0 = 0; n0 < N; n0 += 64) {
        int64_t n1 = n0 + 64 < N ? n0 + 64 : N;
        double acc[64];
        for (int64_t m = 0; m < M; ++m) {
            for (int64_t n = n0; n < n1; ++n) acc[n - n0] = 0.0;
            for (int64_t k = 0; k < K; ++k) {
                double a = load_act(A, m * K + k, dtype);
                const double* w = W + k * N;
                for (int64_t n = n0; n < n1; ++n) acc[n - n0] += a * w[n];
            }
            for (int64_t n = n0; n < n1; ++n) {
                double y = round_act(acc[n - n0], dtype);
                if (bias) y = y + load_act(bias, n, dtype);
                store_act(C, m * N + n, y, dtype);
            }
        }
    }
    return 0;
}

int oracle_w4_fwd(const void* A, const uint8_t* B, const void* S, const void* bias, void* C,
                  int64_t M, int64_t N, int64_t K, int64_t group, int dtype) {
    if (K % 2 || group <= 0 || K % group) return -1;
    double* W = dequant_w4(B, S, K, N, group, dtype);
    if (!W) return -2;
    int rc = matmul_dense(A, W, bias, C, M, N, K, dtype);
    free(W);
    return rc;
}

int oracle_w8_fwd(const void* A, const int8_t* Wnk, const void* S, const void* bias, void* C,
                  int64_t M, int64_t N, int64_t K, int dtype) {
    double* W = (double*)malloc(sizeof(double) * (size_t)K * (size_t)N);
    if (!W) return -2;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        double s = load_act(S, n, dtype);
        for (int64_t k = 0; k < K; ++k)
            W[k * N + n] = round_act((double)Wnk[n * K + k] * s, dtype);
    }
    int rc = matmul_dense(A, W, bias, C, M, N, K, dtype);
    free(W);
    return rc;
}

/* Row-wise symmetric int8 activation quantiser in float32 arithmetic
 * (chatglm_q/int8/quantizer.py:11-19). */
int oracle_act_quant_rowwise(const void* A, int8_t* Aq, float* a_scale, int64_t M, int64_t K, int dtype) {
    for (int64_t m = 0; m < M; ++m) {
        float mx = 0.f;
        for (int64_t k = 0; k < K; ++k) {
            float v = fabsf((float)load_act(A, m * K + k, dtype));
            if (v > mx) mx = v;
        }
        float s = mx / 127.0f;
        if (s < 1e-10f) s = 1e-10f;
        a_scale[m] = s;
        for (int64_t k = 0; k < K; ++k) {
            float q = nearbyintf((float)load_act(A, m * K + k, dtype) / s);
            if (q > 127.f) q = 127.f;
            if (q < -127.f) q = -127.f;
            Aq[m * K + k] = (int8_t)q;
        }
    }
    return 0;
}

/* W8A8: exact i8 x i8 -> i32, epilogue acc * (a_scale[m] * w_scale[n]) in float32
 * (chatglm_q/int8/qlinear.py:60-62 epilogue shape). */
int oracle_w8a8_fwd(const int8_t* Aq, const float* a_scale, const int8_t* Wnk, const void* S,
                    const void* bias, void* C, int64_t M, int64_t N, int64_t K, int dtype) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        float ws = (float)load_act(S, n, dtype);
        for (int64_t m = 0; m < M; ++m) {
            int32_t acc = 0;
            for (int64_t k = 0; k < K; ++k) acc += (int32_t)Aq[m * K + k] * (int32_t)Wnk[n * K + k];
            float comb = a_scale[m] * ws;
            double y = round_act((double)((float)acc * comb), dtype);
            if (bias) y = y + load_act(bias, n, dtype);
            store_act(C, m * N + n, y, dtype);
        }
    }
    return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
