// Chain GEMM "NT" for gfx950:  out[m][n] = epi( Cin[m][n] | 0  +  sum_k A[m][k] * W[n][k] ).
//
// Every output element is ONE k-ascending fp32 fma chain (oracle rule R1).  Two kernels compute
// exactly that chain and therefore the same bits:
//   * gemm_nt_valu  -- LDS-tiled v_fma_f32 kernel (64x64 tile, 4x4 per thread); the simple form,
//                      kept as the on-device cross-check (vfr_set_option("gemm", 0));
//   * gemm_nt_mfma  -- v_mfma_f32_32x32x2_f32 kernel (128x128 tile, 4 waves x 2x2 MFMA tiles),
//                      the production path: the MFMA's accumulate order over k IS the chain.
// Used by: visual MLP (model/models.py:21-26), BiLSTM input/recurrent projections and lang_fc
// (model/models.py:40-47), BERT-branch Linear (:31), VGG fc6/fc7 (get_rgb_features.py:126).
#include "vfr_common.h"

namespace vfr {

// ------------------------------------------------------------------------------------------------
// VALU chain kernel
// ------------------------------------------------------------------------------------------------
constexpr int VBM = 64, VBN = 64, VBK = 16, VPAD = 4;

__global__ __launch_bounds__(256) void gemm_nt_valu(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) float As[VBK][VBM + VPAD];
    __shared__ __attribute__((aligned(16))) float Ws[VBK][VBN + VPAD];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * VBM;
    const int n0 = blockIdx.y * VBN;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t m = m0 + ty * 4 + i;
            int n = n0 + tx * 4 + j;
            acc[i][j] = (g.Cin && m < g.M && n < g.N) ? g.Cin[m * g.ldc + n] : 0.0f;
        }

    for (int k0 = 0; k0 < g.K; k0 += VBK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int e = tid + 256 * i, r = e >> 4, kk = e & 15;
            int64_t m = m0 + r;
            int n = n0 + r, k = k0 + kk;
            As[kk][r] = (m < g.M && k < g.K) ? g.A[m * g.lda + k] : 0.0f;
            Ws[kk][r] = (n < g.N && k < g.K) ? g.W[(int64_t)n * g.ldw + k] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < VBK; ++kk) {
            float4 a = *reinterpret_cast<const float4 *>(&As[kk][ty * 4]);
            float4 w = *reinterpret_cast<const float4 *>(&Ws[kk][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w}, wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(av[i], wv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t m = m0 + ty * 4 + i;
            int n = n0 + tx * 4 + j;
            if (m >= g.M || n >= g.N) continue;
            float v = acc[i][j];
            if (g.epi & EPI_BIAS2) v = v + (g.bias[n] + g.bias2[n]);
            else if (g.epi & EPI_BIAS) v = v + g.bias[n];
            if (g.epi & EPI_RELU) v = v > 0.0f ? v : 0.0f;
            g.out[m * g.ldo + n] = v;
        }
}

int gemm_nt(const GemmArgs &g, hipStream_t st)
{
    if (g.M == 0 || g.N == 0) return VFR_OK;
    VFR_REQUIRE(g.A && g.W && g.out && g.M > 0 && g.N > 0 && g.K >= 0, VFR_EINVAL, "gemm_nt: bad argument");
    VFR_REQUIRE(!(g.epi & (EPI_BIAS | EPI_BIAS2)) || g.bias, VFR_EINVAL, "gemm_nt: bias flag without bias");
    VFR_REQUIRE(!(g.epi & EPI_BIAS2) || g.bias2, VFR_EINVAL, "gemm_nt: bias2 flag without bias2");
    dim3 grid((unsigned)cdiv(g.M, VBM), (unsigned)cdiv(g.N, VBN));
    hipLaunchKernelGGL(gemm_nt_valu, grid, dim3(256), 0, st, g);
    VFR_CHECK_LAUNCH("gemm_nt_valu");
    return VFR_OK;
}

}  // namespace vfr

extern "C" int vfr_linear_f32(const float *A, int64_t M, int K, const float *W, const float *b, int N, int relu,
                              float *out, vfr_stream_t stream)
{
    VFR_REQUIRE(A && W && out && M >= 0 && K > 0 && N > 0, VFR_EINVAL, "vfr_linear_f32: bad argument");
    vfr::GemmArgs g{};
    g.A = A; g.lda = K; g.W = W; g.ldw = K; g.out = out; g.ldo = N; g.M = M; g.N = N; g.K = K;
    g.bias = b;
    g.epi = (b ? vfr::EPI_BIAS : 0) | (relu ? vfr::EPI_RELU : 0);
    return vfr::gemm_nt(g, vfr::as_stream(stream));
}
