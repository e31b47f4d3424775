// Micro-benchmark: cost of wave-uniform operand delivery on gfx950.
//   mode 0: ds_read_b32  all lanes same address     mode 1: ds_read_b64 same address
//   mode 2: ds_read_b128 all lanes same address     mode 3: ds_read_b128 per-lane distinct (conflict-free)
//   mode 4: v_readlane_b32 (VGPR lane -> SGPR)       mode 5: ds_read_b128 same address within 16-lane groups, 4 distinct
// Each wave issues N reads, every result folded into a v_fma so nothing is dead.  Prints cycles/instr per wave
// with W waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i * 1e-3f;
    __syncthreads();
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    float mine = (float)lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int base = ((it + u) & 63) * 16;
            if (MODE == 0) {
                float a = *(volatile float *)&lds[base];
                acc0 = __builtin_fmaf(a, a, acc0);
            } else if (MODE == 1) {
                f2 a = *(volatile f2 *)&lds[base];
                acc0 = __builtin_fmaf(a.x, a.x, acc0); acc1 = __builtin_fmaf(a.y, a.y, acc1);
            } else if (MODE == 2) {
                f4 a = *(volatile f4 *)&lds[base];
                acc0 = __builtin_fmaf(a.x, a.x, acc0); acc1 = __builtin_fmaf(a.y, a.y, acc1);
                acc2 = __builtin_fmaf(a.z, a.z, acc2); acc3 = __builtin_fmaf(a.w, a.w, acc3);
            } else if (MODE == 3) {
                f4 a = *(volatile f4 *)&lds[(base + lane * 4) & 4095];
                acc0 = __builtin_fmaf(a.x, a.x, acc0); acc1 = __builtin_fmaf(a.y, a.y, acc1);
                acc2 = __builtin_fmaf(a.z, a.z, acc2); acc3 = __builtin_fmaf(a.w, a.w, acc3);
            } else if (MODE == 4) {
                float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), (it + u) & 63));
                acc0 = __builtin_fmaf(a, mine, acc0);
            } else {
                f4 a = *(volatile f4 *)&lds[base + (lane >> 4) * 4];
                acc0 = __builtin_fmaf(a.x, a.x, acc0); acc1 = __builtin_fmaf(a.y, a.y, acc1);
                acc2 = __builtin_fmaf(a.z, a.z, acc2); acc3 = __builtin_fmaf(a.w, a.w, acc3);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves_per_block)
{
    const int blocks = 256 * 2, iters = 2000;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)blocks * waves_per_block * 64 * 4);
    hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, iters);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= blocks;
    printf("%-44s waves/block %2d: %7.2f cyc per instr per wave (wall %.3f ms)\n", name, waves_per_block,
           avg / (iters * 16.0), ms);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {1, 4, 8, 16}) {
        run<0>("ds_read_b32 same addr (broadcast)", w);
        run<1>("ds_read_b64 same addr", w);
        run<2>("ds_read_b128 same addr", w);
        run<3>("ds_read_b128 distinct conflict-free", w);
        run<5>("ds_read_b128 4 distinct addrs (16-lane groups)", w);
        run<4>("v_readlane_b32 + fma", w);
    }
    return 0;
}
