// Micro-benchmark (gfx950): which instruction classes of a SECOND wave run beside a wave that streams v_mfma_f32_16x16x4_f32
// on the same SIMD?  512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run the MFMA loop, waves 4-7 (their SIMD
// partners) the other loop.  Three launches per case: MFMA role alone, other role alone, both; times are s_memtime ticks of
// the slowest wave of the role.  Iteration counts are tuned so that both roles alone take about the same time T: beside each
// other they take T if the two classes overlap completely and 2T if they exclude each other.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/coissue.hip -o tools/ubench/coissue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { X_FMA, X_PKFMA, X_EXP, X_RCP, X_IADD, X_IMUL, X_CNDMASK, X_DSREAD, X_MFMA, X_MFMA_BF16, X_GLOAD, NX };

template <int X>
__device__ __forceinline__ float other_loop(int iters, const float *lds, const float *gmem, int lane)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0f + 0.001f * (lane + i);
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = lane * 7 + i;
    f32x4 acc[4] = {};
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 hb;
    for (int i = 0; i < 8; ++i) hb[i] = (__bf16)(0.5f + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (X == X_FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (X == X_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (X == X_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (X == X_IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (X == X_IMUL) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (X == X_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : "vcc");
            }
            if (X == X_PKFMA) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 v = {a[i], a[i + 1]};
                    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v));
                    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v));
                    a[i] = v.x; a[i + 1] = v.y;
                }
            }
            if (X == X_DSREAD) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = *(const volatile f32x4 *)(lds + ((lane * 4 + i * 256 + it * 4) & 4095));
                    a[i] += v.x; a[i + 4] += v.w;
                }
            }
            if (X == X_GLOAD) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = *(const volatile f32x4 *)(gmem + ((lane * 4 + i * 256 + it * 1024) & 65535));
                    a[i] += v.x; a[i + 4] += v.w;
                }
            }
            if (X == X_MFMA) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[i + 4], acc[i], 0, 0, 0);
            }
            if (X == X_MFMA_BF16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb, hb, acc[i], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    return s;
}

template <bool BF16>
__device__ __forceinline__ float mfma_loop(int iters, int lane)
{
    f32x4 acc[8] = {};
    const float a = 1.0f + lane, b = 0.5f;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 hb;
    for (int i = 0; i < 8; ++i) hb[i] = (__bf16)(0.5f + i + lane);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = BF16 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb, hb, acc[i], 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    return s;
}

// mode bit 0: the MFMA role runs, bit 1: the other role runs
template <int X, bool BF16>
__global__ void __launch_bounds__(512) k(float *out, unsigned long long *ticks, const float *gmem, int it_mfma, int it_other, int mode)
{
    __shared__ __attribute__((aligned(16))) float lds[4096 + 16];
    for (int i = threadIdx.x; i < 4096 + 16; i += blockDim.x) lds[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, role = wave >> 2;
    float s = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0 && (mode & 1)) s = mfma_loop<BF16>(it_mfma, lane);
    if (role == 1 && (mode & 2)) s = other_loop<X>(it_other, lds, gmem, lane);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) atomicMax(&ticks[role], t1 - t0);
}

template <int X, bool BF16 = false> void run(const char *name, float *out, unsigned long long *ticks, const float *gmem, double per_iter_guess)
{
    const int it_mfma = 4000;                      // 8 MFMAs of 32 cycles per iteration -> ~1.02 M cycles
    auto launch = [&](int im, int io, int mode, unsigned long long *h) {
        hipMemset(ticks, 0, 16);
        hipLaunchKernelGGL((k<X, BF16>), dim3(256), dim3(512), 0, 0, out, ticks, gmem, im, io, mode);
        hipDeviceSynchronize();
        hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
    };
    unsigned long long h[2];
    launch(it_mfma, 100, 3, h);                    // warm-up
    launch(it_mfma, 0, 1, h);
    const double t_mfma = (double)h[0];
    int it_other = 2000;
    launch(0, it_other, 2, h);
    it_other = (int)(it_other * t_mfma / (double)h[1]);     // other role alone ~ as long as the MFMA role alone
    launch(0, it_other, 2, h);
    const double t_other = (double)h[1];
    launch(it_mfma, it_other, 3, h);
    const double both_m = (double)h[0], both_o = (double)h[1];
    const double both = both_m > both_o ? both_m : both_o;
    // share of the shorter role hidden behind the longer one: 1 = complete overlap, 0 = the two add up
    const double lo = t_mfma < t_other ? t_mfma : t_other, hi = t_mfma > t_other ? t_mfma : t_other;
    const double hidden = (t_mfma + t_other - both) / lo;
    printf("%-34s mfma alone %8.0f  other alone %8.0f  together: mfma %8.0f other %8.0f  -> %3.0f %% of the shorter role hidden (%.2f x the longer)\n",
           name, t_mfma, t_other, both_m, both_o, 100 * hidden, both / hi);
}

int main()
{
    float *out, *gmem; unsigned long long *ticks;
    hipMalloc(&out, sizeof(float) * 512 * 256); hipMalloc(&ticks, 16); hipMalloc(&gmem, 65536 * 4 + 64);
    hipMemset(gmem, 0, 65536 * 4 + 64);
    printf("s_memtime ticks (= core clocks here: 32 per back-to-back 16x16x4 MFMA); one 512-thread workgroup per CU: waves 0-3 stream v_mfma_f32_16x16x4_f32, waves 4-7 the other class\n");
    run<X_FMA>("v_fma_f32", out, ticks, gmem, 0);
    run<X_PKFMA>("v_pk_fma_f32", out, ticks, gmem, 0);
    run<X_EXP>("v_exp_f32", out, ticks, gmem, 0);
    run<X_RCP>("v_rcp_f32", out, ticks, gmem, 0);
    run<X_IADD>("v_add_u32", out, ticks, gmem, 0);
    run<X_IMUL>("v_mul_lo_u32", out, ticks, gmem, 0);
    run<X_CNDMASK>("v_cndmask_b32", out, ticks, gmem, 0);
    run<X_DSREAD>("ds_read_b128 (+2 v_add_f32 each)", out, ticks, gmem, 0);
    run<X_GLOAD>("global_load_dwordx4 L2 hits (+2 adds)", out, ticks, gmem, 0);
    run<X_MFMA>("v_mfma_f32_16x16x4_f32 (sanity)", out, ticks, gmem, 0);
    run<X_MFMA_BF16>("v_mfma_f32_16x16x32_bf16", out, ticks, gmem, 0);
    printf("the same with waves 0-3 streaming v_mfma_f32_16x16x32_bf16\n");
    run<X_FMA, true>("v_fma_f32", out, ticks, gmem, 0);
    run<X_PKFMA, true>("v_pk_fma_f32", out, ticks, gmem, 0);
    run<X_EXP, true>("v_exp_f32", out, ticks, gmem, 0);
    run<X_IADD, true>("v_add_u32", out, ticks, gmem, 0);
    run<X_DSREAD, true>("ds_read_b128 (+2 v_add_f32 each)", out, ticks, gmem, 0);
    run<X_MFMA, true>("v_mfma_f32_16x16x4_f32", out, ticks, gmem, 0);
    return hipGetLastError() != hipSuccess;
}
