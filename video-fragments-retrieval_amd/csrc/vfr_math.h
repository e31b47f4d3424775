// Canonical device math (rule R5 of the oracle): IEEE ops only, explicit fmaf, no contraction
// (the library is compiled with -ffp-contract=off), so every function below returns the same bits
// as its twin in oracle/vfr_oracle.c.
#pragma once
#include <hip/hip_runtime.h>

namespace vfr {

#ifdef VFR_MATH_PLAIN     /* cross-check build (tools/build_variant.sh plain "-DVFR_MATH_PLAIN" ...): the oracle's text, hipcc's IEEE divisions */
__device__ __forceinline__ float c_expf(float x)
{
    x = fminf(fmaxf(x, -80.0f), 80.0f);
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r) + 1.0f;
    float s = __uint_as_float((unsigned)((int)n + 127) << 23);
    return y * s;
}
__device__ __forceinline__ float c_sigmoidf(float x) { return 1.0f / (1.0f + c_expf(-x)); }
__device__ __forceinline__ float c_tanhf(float x)
{
    float ax = __builtin_fabsf(x);
    float e = c_expf(2.0f * ax);
    float t = 1.0f - 2.0f / (e + 1.0f);
    return __builtin_copysignf(t, x);
}

#else
// The same VALUES as the oracle's c_expf / c_sigmoidf / c_tanhf, instruction for instruction where it matters and CHEAPER
// where a cheaper form is provably the same bits (the gate epilogue of the fused LSTM step is 4 % .. 8 % of that kernel's
// matrix-pipe time, and vector-ALU time is matrix-pipe time on gfx950):
//   * 2^n * y through v_ldexp_f32 instead of building the power of two from bits (n in [-116, 116], y in (0.5, 2): the product
//     is a normal number, a power-of-two scaling is exact either way);
//   * the clamp's lower side dropped where the argument cannot be below it (tanh: 2|x| >= 0), the negation of the sigmoid's
//     argument applied after the clamp (clamp(-x) == -clamp(x) for a symmetric range; NaN arguments end where they did);
//   * 1 / d and 2 / d for d in [1, 2^120) -- d = 1 + exp(.) here, never above 1 + e^80 -- as v_rcp_f32 + two Newton steps in
//     fma instead of the compiler's scaled IEEE division (5 instructions against 11): CORRECTLY ROUNDED for every float in that
//     range on gfx950 (tools/ubench/rcp_exact.hip compares all 1.0e9 of them with the IEEE quotient, and the three functions
//     below with their plain restatements over all 2^32 arguments: profiles/r4_rcp_exact.txt), and 2 * (1 / d) == 2 / d exactly.
// c_expf_clamped: x already in [-80, 80]
__device__ __forceinline__ float c_expf_clamped(float x)
{
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r) + 1.0f;
    return __builtin_amdgcn_ldexpf(y, (int)n);
}
__device__ __forceinline__ float c_expf(float x) { return c_expf_clamped(fminf(fmaxf(x, -80.0f), 80.0f)); }
// 1 / d, correctly rounded, for 1 <= d < 2^120 (see above)
__device__ __forceinline__ float c_rcp_ge1(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float c_sigmoidf(float x)
{
    // clamp(-x) as -clamp'(x) with the two sides taken in the order that sends a NaN where the oracle's clamp(-x) sends it (-80)
    const float t = fmaxf(fminf(x, 80.0f), -80.0f);
    return c_rcp_ge1(1.0f + c_expf_clamped(-t));
}
__device__ __forceinline__ float c_tanhf(float x)
{
    float ax = __builtin_fabsf(x);
    float e = c_expf_clamped(fminf(2.0f * ax, 80.0f));
    float t = 1.0f - 2.0f * c_rcp_ge1(e + 1.0f);
    return __builtin_copysignf(t, x);
}

#endif

// x / L for a small integer L known at compile time (a moment's length, 1 .. 21): the product with RN(1 / L), one exact
// residual, one correction -- 3 instructions against the 11 of the compiler's scaled IEEE division, and the CORRECTLY ROUNDED
// quotient for every float x in [2^-60, 2^100] and every L in 1 .. 21 (enumerated on gfx950: tools/ubench/rcp_exact.hip,
// profiles/r4_rcp_exact.txt).  Callers keep x inside that range (or take the IEEE division).
__device__ __forceinline__ float c_div_small(float x, int L)
{
    const float fl = (float)L, rl = 1.0f / fl;
    if ((L & (L - 1)) == 0) return x * rl;                 // a power of two: exact
    const float q0 = x * rl;
    const float r = __builtin_fmaf(-fl, q0, x);
    return __builtin_fmaf(r, rl, q0);
}
constexpr float C_DIV_SMALL_LO = 8.673617379884035e-19f;   // 2^-60
constexpr float C_DIV_SMALL_HI = 1.2676506002282294e30f;   // 2^100

// order-preserving key of a non-negative fp32 distance and a 32-bit moment id
__device__ __forceinline__ unsigned long long make_key(float d, unsigned id)
{
    return ((unsigned long long)__float_as_uint(d) << 32) | id;
}
// position of span (s,e) in utils.generate_moments(n) order (model/utils.py:71-75)
__device__ __forceinline__ int moment_index(int n, int s, int e)
{
    return s == e ? s : n + (s * (2 * n - s - 1)) / 2 + (e - s - 1);
}

}  // namespace vfr
