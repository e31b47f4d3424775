// Micro-benchmark (gfx950): sustained issue rate of v_mfma_f32_32x32x2_f32 in the instruction mixes of the chain GEMM's
// inner block, one and two waves per SIMD.  Reports cycles per MFMA per SIMD (s_memtime) and the implied TFLOP/s at the
// measured wall time.   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_rates.hip -o tools/ubench/mfma_rates.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

enum { BARE, CNDMASK, CNDMASK_LDS, CNDMASK_LDS_BARRIER, LDS_SPREAD, LDS_SPREAD2, LDS_CARRY, LDS_CARRY_BARRIER, LDS_RING3, M16_BARE, M16_LDS, NTESTS };

template <int T>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *ticks, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * 256 * 36];
    for (int i = threadIdx.x; i < 2 * 256 * 36; i += blockDim.x) lds[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31;
    const bool h = lane >= 32;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float *ap = &lds[((wave >> 1) * 64 + l31) * 36], *wp = &lds[(128 + (wave & 1) * 64 + l31) * 36];
    float a0s = 1.0f + lane, a1s = 2.0f, b0s = 3.0f, b1s = 0.5f;
    f4 ca[2], cb[2];
    ca[0] = *(const f4 *)(ap); ca[1] = *(const f4 *)(ap + 32 * 36); cb[0] = *(const f4 *)(wp); cb[1] = *(const f4 *)(wp + 32 * 36);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (T == M16_BARE || T == M16_LDS) {
            // 16x16x4 form: wave tile 64x64 = 4x4 tiles, 16 MFMAs (32 cycles each) per k-step of 4; fragments are ONE dword
            // per lane and tile row block (lane reads A[16*blk + (lane&15)][k + (lane>>4)] directly: ds_read_b32)
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            static_assert(sizeof(f32x4) == 16, "");
            f32x4 (*c16)[4] = reinterpret_cast<f32x4 (*)[4]>(&acc[0][0]);     // 16 x f32x4 = the same 64 registers
            const float *ap16 = &lds[((wave >> 1) * 64 + (lane & 15)) * 36 + (lane >> 4)];
            const float *wp16 = &lds[(128 + (wave & 1) * 64 + (lane & 15)) * 36 + (lane >> 4)];
            float fa[2][4], fb[2][4];
#pragma unroll
            for (int b = 0; b < 4; ++b) { fa[0][b] = T == M16_LDS ? ap16[b * 16 * 36] : a0s + b; fb[0][b] = T == M16_LDS ? wp16[b * 16 * 36] : b0s + b; }
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const int cur = k4 & 1, nxt = cur ^ 1;
                if (T == M16_LDS) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) { fa[nxt][b] = ap16[b * 16 * 36 + ((k4 + 1) & 7) * 4]; fb[nxt][b] = wp16[b * 16 * 36 + ((k4 + 1) & 7) * 4]; }
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) { fa[nxt][b] = fa[cur][b]; fb[nxt][b] = fb[cur][b]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        c16[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][i], fb[cur][j], c16[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (T == LDS_RING3) {
            // ring of 3 fragment sets: the reads of slice k4+2 are spread one per two MFMAs over slice k4
            f4 ra[3][2], rb[3][2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                ra[d][0] = *(const f4 *)(ap + d * 4); ra[d][1] = *(const f4 *)(ap + 32 * 36 + d * 4);
                rb[d][0] = *(const f4 *)(wp + d * 4); rb[d][1] = *(const f4 *)(wp + 32 * 36 + d * 4);
            }
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const int cur = k4 % 3, nxt = (k4 + 2) % 3;
                const int kk = (k4 + 2) & 7;
                ra[nxt][0] = *(const f4 *)(ap + kk * 4); ra[nxt][1] = *(const f4 *)(ap + 32 * 36 + kk * 4);
                rb[nxt][0] = *(const f4 *)(wp + kk * 4); rb[nxt][1] = *(const f4 *)(wp + 32 * 36 + kk * 4);
                const f4 a0 = ra[cur][0], a1 = ra[cur][1], b0 = rb[cur][0], b1 = rb[cur][1];
                {
                    const float fa0 = h ? a0.y : a0.x, fa1 = h ? a1.y : a1.x, fb0 = h ? b0.y : b0.x, fb1 = h ? b1.y : b1.x;
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1, acc[1][1], 0, 0, 0);
                }
                {
                    const float fa0 = h ? a0.w : a0.z, fa1 = h ? a1.w : a1.z, fb0 = h ? b0.w : b0.z, fb1 = h ? b1.w : b1.z;
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1, acc[1][1], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (T == BARE) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0s, b0s, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0s, b1s, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1s, b0s, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1s, b1s, acc[1][1], 0, 0, 0);
            }
        } else {
            f4 fa[2][2], fb[2][2];
            if (T >= LDS_CARRY) {
                fa[0][0] = ca[0]; fa[0][1] = ca[1]; fb[0][0] = cb[0]; fb[0][1] = cb[1];
            } else if (T >= CNDMASK_LDS) {
                fa[0][0] = *(const f4 *)(ap); fa[0][1] = *(const f4 *)(ap + 32 * 36);
                fb[0][0] = *(const f4 *)(wp); fb[0][1] = *(const f4 *)(wp + 32 * 36);
            } else {
                fa[0][0] = fa[0][1] = fb[0][0] = fb[0][1] = (f4){a0s, a1s, b0s, b1s};
                fa[1][0] = fa[1][1] = fb[1][0] = fb[1][1] = (f4){a1s, a0s, b1s, b0s};
            }
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const int cur = k4 & 1, nxt = cur ^ 1;
                if (T >= CNDMASK_LDS && k4 + 1 < 8) {
                    fa[nxt][0] = *(const f4 *)(ap + (k4 + 1) * 4); fa[nxt][1] = *(const f4 *)(ap + 32 * 36 + (k4 + 1) * 4);
                    fb[nxt][0] = *(const f4 *)(wp + (k4 + 1) * 4); fb[nxt][1] = *(const f4 *)(wp + 32 * 36 + (k4 + 1) * 4);
                }
                if (T >= LDS_CARRY && k4 == 7) {          // next iteration's slice 0, issued under this slice's MFMAs
                    const int o = ((it + 1) & 1) * 16;
                    ca[0] = *(const f4 *)(ap + o); ca[1] = *(const f4 *)(ap + 32 * 36 + o);
                    cb[0] = *(const f4 *)(wp + o); cb[1] = *(const f4 *)(wp + 32 * 36 + o);
                }
                if (T < LDS_SPREAD) __builtin_amdgcn_sched_barrier(0);
                const f4 a0 = fa[cur][0], a1 = fa[cur][1], b0 = fb[cur][0], b1 = fb[cur][1];
                {
                    const float fa0 = h ? a0.y : a0.x, fa1 = h ? a1.y : a1.x, fb0 = h ? b0.y : b0.x, fb1 = h ? b1.y : b1.x;
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1, acc[1][1], 0, 0, 0);
                }
                {
                    const float fa0 = h ? a0.w : a0.z, fa1 = h ? a1.w : a1.z, fb0 = h ? b0.w : b0.z, fb1 = h ? b1.w : b1.z;
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1, acc[1][1], 0, 0, 0);
                }
                if (T == LDS_SPREAD) {                    // one fragment read after every second MFMA
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                if (T == LDS_SPREAD2) {                   // one MFMA first, then read / 2 MFMA alternating
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (T == CNDMASK_LDS_BARRIER || T == LDS_CARRY_BARRIER) __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int T> void run(const char *name, float *out, unsigned long long *ticks)
{
    const int iters = 2000;                       // 64 MFMAs per iteration per wave
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = 256 * wps;             // 4-wave blocks: wps blocks per CU
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, out, ticks, 10);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8]; hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
        const double mfma_per_simd = (double)iters * 64 * wps;
        const double tf = (double)blocks * 4 * iters * 64 * 4096.0 / (ms * 1e-3) / 1e12;
        printf("%-44s %d wave(s)/SIMD: %7.1f s_memtime ticks per MFMA per SIMD, %7.3f ms -> %6.1f TFLOP/s (%.0f %% of 157.3), %.2f G ticks/s\n",
               name, wps, (double)h[0] / mfma_per_simd, ms, tf, 100 * tf / 157.3, (double)h[0] / (ms * 1e6));
    }
}

int main()
{
    float *out; unsigned long long *ticks;
    hipMalloc(&out, sizeof(float) * 512 * 256); hipMalloc(&ticks, 8 * 512);
    run<BARE>("bare MFMA, 4 accumulators", out, ticks);
    run<CNDMASK>("+ half-select v_cndmask per operand", out, ticks);
    run<CNDMASK_LDS>("+ ds_read_b128 fragments (GEMM inner block)", out, ticks);
    run<CNDMASK_LDS_BARRIER>("+ one barrier per 64 MFMAs", out, ticks);
    run<LDS_CARRY>("fragments carried across iterations (no restart)", out, ticks);
    run<LDS_CARRY_BARRIER>("carried fragments + barrier per 64 MFMAs", out, ticks);
    run<LDS_RING3>("ring of 3, reads of slice+2 spread (2 MFMA,1 read)", out, ticks);
    run<M16_BARE>("16x16x4 MFMA bare (x2 = per 4096 flop)", out, ticks);
    run<M16_LDS>("16x16x4 MFMA + ds_read_b32 fragments", out, ticks);
    run<LDS_SPREAD>("fragment reads spread: (2 MFMA, 1 read) x4", out, ticks);
    run<LDS_SPREAD2>("fragment reads spread: MFMA,(read,2 MFMA)..", out, ticks);
    return hipGetLastError() != hipSuccess;
}
