// Canonical device math (rule R5 of the oracle): IEEE ops only, explicit fmaf, no contraction
// (the library is compiled with -ffp-contract=off), so every function below returns the same bits
// as its twin in oracle/vfr_oracle.c.
#pragma once
#include <hip/hip_runtime.h>

namespace vfr {

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
