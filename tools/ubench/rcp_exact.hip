// Exhaustive proof-by-enumeration for the cheaper forms in csrc/vfr_math.h (gfx950):
//   1. c_rcp_ge1(d) == 1.0f / d (hipcc's correctly rounded IEEE division) for EVERY float d in [1, 2^120);
//   1b. c_div_small(x, L) == x / (float)L for EVERY float x in [2^-60, 2^100] and L = 1 .. 21;
//   2. c_expf, c_sigmoidf, c_tanhf == their plain restatements (the oracle's text: power of two built from bits, IEEE divisions,
//      clamp of the negated argument) for EVERY one of the 2^32 float arguments, NaNs compared by class.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I video-fragments-retrieval_amd/csrc
//                              tools/ubench/rcp_exact.hip -o tools/ubench/rcp_exact.bin && tools/ubench/rcp_exact.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "vfr_math.h"

namespace ref {   // the functions as oracle/vfr_oracle.c states them
__device__ __forceinline__ float expf_(float x)
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
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf_(-x)); }
__device__ __forceinline__ float tanhf_(float x)
{
    float ax = __builtin_fabsf(x);
    float e = expf_(2.0f * ax);
    float t = 1.0f - 2.0f / (e + 1.0f);
    return __builtin_copysignf(t, x);
}
}  // namespace ref

__device__ __forceinline__ bool same(float a, float b)
{
    const bool na = a != a, nb = b != b;
    return (na && nb) || (!na && !nb && __float_as_uint(a) == __float_as_uint(b));
}

// what: 0 = rcp over bit patterns [lo, hi); 1 / 2 / 3 = exp / sigmoid / tanh over [lo, hi); 4 = x / L, L = 1 .. 21
__global__ void check(int what, unsigned long long lo, unsigned long long hi, unsigned long long *bad, unsigned *first)
{
    unsigned long long n = 0;
    for (unsigned long long b = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)b);
        bool ok;
        if (what == 0) ok = same(vfr::c_rcp_ge1(x), 1.0f / x) && same(2.0f * vfr::c_rcp_ge1(x), 2.0f / x);
        else if (what == 1) ok = same(vfr::c_expf(x), ref::expf_(x));
        else if (what == 2) ok = same(vfr::c_sigmoidf(x), ref::sigmoidf_(x));
        else if (what == 3) ok = same(vfr::c_tanhf(x), ref::tanhf_(x));
        else {
            ok = true;
#pragma unroll
            for (int L = 1; L <= 21; ++L) ok = ok && same(vfr::c_div_small(x, L), x / (float)L);
        }
        if (!ok) { if (n == 0) atomicMin(first, (unsigned)b); ++n; }
    }
    if (n) atomicAdd(bad, n);
}

int main()
{
    unsigned long long *bad; unsigned *first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4);
    const char *names[5] = {"1/d and 2/d, d in [1, 2^120)", "c_expf, all 2^32 arguments", "c_sigmoidf, all 2^32 arguments", "c_tanhf, all 2^32 arguments",
                            "x / L, x in [2^-60, 2^100], L 1..21"};
    int rc = 0;
    for (int what = 0; what < 5; ++what) {
        // (2^-60 = 0x21800000, 2^100 = 0x71800000: the range is closed, hence + 1)
        const unsigned long long lo = what == 0 ? 0x3F800000ull : what == 4 ? 0x21800000ull : 0ull, hi = what == 0 ? 0x7B800000ull : what == 4 ? 0x71800001ull : (1ull << 32);
        unsigned long long z = 0; unsigned f = 0xFFFFFFFFu;
        hipMemcpy(bad, &z, 8, hipMemcpyHostToDevice); hipMemcpy(first, &f, 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, what, lo, hi, bad, first);
        hipDeviceSynchronize();
        hipMemcpy(&z, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
        printf("%-34s  %llu values  mismatches %llu", names[what], hi - lo, z);
        if (z) { printf("  (smallest failing bit pattern 0x%08X)", f); rc = 1; }
        printf("\n");
    }
    return rc;
}
