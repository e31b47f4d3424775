// Micro-benchmark (gfx950): issue cost of the instructions the moment scorer is built from, relative to v_fma_f32.
// Every test is an inline-asm block of 16 instructions repeated ITERS times by every wave; W waves per SIMD.
// Output: ns per wave-instruction per SIMD and the ratio to v_fma_f32 (== 4 cycles per wave64 instruction).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/issue_rates.hip -o tools/ubench/issue_rates && tools/ubench/issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define REP16(X) X X X X X X X X X X X X X X X X

enum { T_FMA, T_PKFMA, T_PKADD_OPSEL, T_PKFMA_SGPR, T_LDS128_BCAST, T_LDS128_LANE, T_LDS64_BCAST, T_LDS32_BCAST,
       T_READLANE, T_DPP_WAVESHR, T_DPP_ROWSHR, T_CMP_SGPR, T_MIX_BCAST, T_MIX_PK_BCAST, T_BPERMUTE, T_SQRT, T_COUNT };

template <int T>
__global__ void __launch_bounds__(512) k(float *out, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i * 1e-3f + seed;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    f2 p0 = {seed, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f;
    f2 q = {seed * 0.5f, seed * 0.25f};
    f4 r0, r1, r2, r3;
    r0 = r1 = r2 = r3 = (f4){0.f, 0.f, 0.f, 0.f};
    unsigned bc = 0;                                    // wave-uniform LDS byte address
    unsigned la = (unsigned)lane * 16u;                 // per-lane address
    for (int it = 0; it < iters; ++it) {
        if (T == T_FMA) {
            asm volatile(REP16("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                               "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
        } else if (T == T_PKFMA) {
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                               "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        } else if (T == T_PKADD_OPSEL) {               // both halves read the LOW half of the second source, negated
            asm volatile(REP16("v_pk_add_f32 %0, %0, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                               "v_pk_add_f32 %2, %2, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                               "v_pk_add_f32 %4, %4, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %5, %5, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                               "v_pk_add_f32 %6, %6, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %7, %7, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        } else if (T == T_PKFMA_SGPR) {                // second source from an SGPR pair
            asm volatile(REP16("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                               "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(q));
        } else if (T == T_LDS128_BCAST || T == T_LDS128_LANE) {
            const unsigned ad = (T == T_LDS128_BCAST) ? bc : la;
            asm volatile(REP16("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                               "ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(ad));
        } else if (T == T_LDS64_BCAST) {
            asm volatile(REP16("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:1024\n ds_read_b64 %2, %4 offset:2048\n ds_read_b64 %3, %4 offset:3072\n"
                               "ds_read_b64 %0, %4 offset:4096\n ds_read_b64 %1, %4 offset:5120\n ds_read_b64 %2, %4 offset:6144\n ds_read_b64 %3, %4 offset:7168\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(bc));
        } else if (T == T_LDS32_BCAST) {
            asm volatile(REP16("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:1024\n ds_read_b32 %2, %4 offset:2048\n ds_read_b32 %3, %4 offset:3072\n"
                               "ds_read_b32 %0, %4 offset:4096\n ds_read_b32 %1, %4 offset:5120\n ds_read_b32 %2, %4 offset:6144\n ds_read_b32 %3, %4 offset:7168\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(bc));
        } else if (T == T_READLANE) {                  // VGPR lane -> SGPR, then consumed by a VALU op (the real use)
            asm volatile(REP16("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 5\n v_readlane_b32 s22, %0, 7\n v_readlane_b32 s23, %0, 9\n"
                               "v_readlane_b32 s24, %0, 11\n v_readlane_b32 s25, %0, 13\n v_readlane_b32 s26, %0, 15\n v_readlane_b32 s27, %0, 17\n")
                         "s_nop 4\n v_add_f32 %1, s20, %1\n v_add_f32 %1, s27, %1\n"
                         : "+v"(a0), "+v"(a1) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (T == T_DPP_WAVESHR) {
            asm volatile(REP16("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (T == T_DPP_ROWSHR) {
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (T == T_CMP_SGPR) {                  // compare into an SGPR pair + scalar popcount/add (rank counting)
            asm volatile(REP16("v_cmp_lt_f32 s[20:21], %0, %1\n s_bcnt1_i32_b64 s22, s[20:21]\n s_add_u32 s23, s23, s22\n"
                               "v_cmp_lt_f32 s[24:25], %1, %0\n s_bcnt1_i32_b64 s26, s[24:25]\n s_add_u32 s27, s27, s26\n"
                               "v_cmp_lt_f32 s[20:21], %0, %1\n s_bcnt1_i32_b64 s22, s[20:21]\n s_add_u32 s23, s23, s22\n"
                               "v_cmp_lt_f32 s[24:25], %1, %0\n s_bcnt1_i32_b64 s26, s[24:25]\n s_add_u32 s27, s27, s26\n"
                               "v_cmp_lt_f32 s[20:21], %0, %1\n s_bcnt1_i32_b64 s22, s[20:21]\n s_add_u32 s23, s23, s22\n"
                               "v_cmp_lt_f32 s[24:25], %1, %0\n s_bcnt1_i32_b64 s26, s[24:25]\n s_add_u32 s27, s27, s26\n"
                               "v_cmp_lt_f32 s[20:21], %0, %1\n s_bcnt1_i32_b64 s22, s[20:21]\n s_add_u32 s23, s23, s22\n"
                               "v_cmp_lt_f32 s[24:25], %1, %0\n s_bcnt1_i32_b64 s26, s[24:25]\n s_add_u32 s27, s27, s26\n")
                         : "+v"(a0), "+v"(a1) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        } else if (T == T_MIX_BCAST) {                 // the scorer's inner mix today: 1 broadcast b128 per 12 VALU
            asm volatile(REP16("ds_read_b128 %8, %10\n"
                               "v_sub_f32 %0, %9, %0\n v_add_f32 %0, %9, %0\n v_fma_f32 %1, %0, %0, %1\n v_sub_f32 %2, %9, %2\n v_add_f32 %2, %9, %2\n v_fma_f32 %3, %2, %2, %3\n"
                               "v_sub_f32 %4, %9, %4\n v_add_f32 %4, %9, %4\n v_fma_f32 %5, %4, %4, %5\n v_sub_f32 %6, %9, %6\n v_add_f32 %6, %9, %6\n v_fma_f32 %7, %6, %6, %7\n"
                               "s_waitcnt lgkmcnt(1)\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(r0) : "v"(seed), "v"(bc));
        } else if (T == T_MIX_PK_BCAST) {              // packed variant: 1 broadcast b128 per 6 packed VALU
            asm volatile(REP16("ds_read_b128 %8, %10\n"
                               "v_pk_add_f32 %0, %0, %9 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %0, %0, %9\n v_pk_fma_f32 %1, %0, %0, %1\n"
                               "v_pk_add_f32 %2, %2, %9 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, %2, %9\n v_pk_fma_f32 %3, %2, %2, %3\n"
                               "s_waitcnt lgkmcnt(1)\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "=&v"(r0) : "v"(q), "v"(bc));
        } else if (T == T_BPERMUTE) {
            asm volatile(REP16("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n"
                               "ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n"
                               "s_waitcnt lgkmcnt(0)\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(la >> 2));
        } else if (T == T_SQRT) {
            asm volatile(REP16("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                               "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y +
                                                 r0.x + r1.y + r2.z + r3.w;
}

struct Test { const char *name; int instr_per_iter; void (*launch)(float *, int, int, int); };

template <int T> void launch(float *out, int blocks, int threads, int iters)
{
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
}

int main()
{
    float *out;
    hipMalloc(&out, sizeof(float) * 512 * 512 * 4);
    const Test tests[] = {
        {"v_fma_f32", 128, launch<T_FMA>}, {"v_pk_fma_f32", 128, launch<T_PKFMA>}, {"v_pk_add_f32 op_sel+neg", 128, launch<T_PKADD_OPSEL>},
        {"v_pk_add_f32 sgpr src", 128, launch<T_PKFMA_SGPR>}, {"ds_read_b128 uniform addr", 128, launch<T_LDS128_BCAST>},
        {"ds_read_b128 per-lane", 128, launch<T_LDS128_LANE>}, {"ds_read_b64 uniform addr", 128, launch<T_LDS64_BCAST>},
        {"ds_read_b32 uniform addr", 128, launch<T_LDS32_BCAST>}, {"v_readlane_b32", 128, launch<T_READLANE>},
        {"v_mov_b32_dpp wave_shr:1", 128, launch<T_DPP_WAVESHR>}, {"v_add_f32_dpp row_shr:1", 128, launch<T_DPP_ROWSHR>},
        {"v_cmp->sgpr + s_bcnt1 + s_add (per cmp)", 128, launch<T_CMP_SGPR>}, {"mix: b128 bcast + 12 VALU (per group)", 16, launch<T_MIX_BCAST>},
        {"mix: b128 bcast + 6 pk VALU (per group)", 16, launch<T_MIX_PK_BCAST>}, {"ds_bpermute_b32", 128, launch<T_BPERMUTE>},
        {"v_sqrt_f32", 128, launch<T_SQRT>}};
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double fma_ns[3] = {0, 0, 0};
    printf("%-44s %12s %12s %12s   (ns per wave-instruction per SIMD; xN = ratio to v_fma_f32 at the same occupancy)\n", "instruction",
           "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    for (const Test &t : tests) {
        printf("%-44s", t.name);
        for (int wi = 0; wi < 3; ++wi) {
            const int wps = 1 << wi;                   // waves per SIMD
            const int threads = 256, blocks = 256 * wps;   // blocks of 4 waves (one per SIMD), wps blocks per CU
            t.launch(out, blocks, threads, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            t.launch(out, blocks, threads, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double ns = ms * 1e6 / ((double)iters * t.instr_per_iter * wps);
            if (&t == &tests[0]) fma_ns[wi] = ns;
            printf(" %7.3f x%4.2f", ns, ns / fma_ns[wi]);
        }
        printf("\n");
    }
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
