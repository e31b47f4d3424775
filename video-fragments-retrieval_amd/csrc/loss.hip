// f2: Trainer.ranking_loss (model/main.py:214-232) -- the loss of train_epoch / test_epoch -- forward and backward.
//
//   per sample i:  c_x = mean over the rows r of set x with mask[r] == i of || (x[r] - lang[i]) + eps ||   (x = posit|intra|inter)
//   loss = sum_i relu(c_posit - c_intra + b) + lamb * relu(c_posit - c_inter + b)
//
// The reference loops over the samples in Python with three pairwise_distance calls, three boolean-mask gathers and a host
// sync (maskp.max().item()) per batch.  Here: one launch for the row distances (a row per lane, the oracle's k-ascending
// chain), one for the per-sample means / hinge terms / loss (row order, so the sums are the oracle's), and for the backward
// one launch for the three row gradients and one for grad_lang (a thread per (sample, k), rows in order: deterministic,
// no float atomics).  Tiny kernels: this is API coverage for the training drivers, not a hot spot.
#include "vfr_common.h"

namespace vfr {

// rows of the three sets back to back: [0,P) posit, [P,P+Nn) intra, [P+Nn, 2P+Nn) inter
__global__ __launch_bounds__(256) void ranking_row_dist_kernel(const float *__restrict__ posit, const float *__restrict__ intra,
                                                               const float *__restrict__ inter, const float *__restrict__ lang,
                                                               const int64_t *__restrict__ maskp, const int64_t *__restrict__ maskn,
                                                               int64_t P, int64_t Nn, int S, int D, float eps,
                                                               float *__restrict__ dist)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= 2 * P + Nn) return;
    const float *x;
    int64_t i;
    if (r < P) { x = posit + r * D; i = maskp[r]; }
    else if (r < P + Nn) { x = intra + (r - P) * D; i = maskn[r - P]; }
    else { x = inter + (r - P - Nn) * D; i = maskp[r - P - Nn]; }
    if (i < 0 || i >= S) { dist[r] = 0.0f; return; }          // a row of no sample: never read
    const float *l = lang + i * D;
    float acc = 0.0f;
    for (int k = 0; k < D; ++k) {
        const float d = (x[k] - l[k]) + eps;
        acc = __builtin_fmaf(d, d, acc);
    }
    dist[r] = __builtin_sqrtf(acc);
}

// per_sample [S,8] = c_posit, c_intra, c_inter, t1, t2, n_posit, n_intra, loss_i.  One wave per sample: the masks are
// scanned 64 rows at a time (ballot), the matching rows' distances added in row order by a uniform loop over the set bits --
// the sums of the reference's boolean-mask means, same order, without every thread walking all P rows one dependent load
// at a time (154 us for 256 samples).
__global__ __launch_bounds__(256) void ranking_reduce_kernel(const float *__restrict__ dist, const int64_t *__restrict__ maskp,
                                                             const int64_t *__restrict__ maskn, int64_t P, int64_t Nn, int S,
                                                             float b, float lamb, float *__restrict__ per_sample)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= S) return;
    float sp = 0.0f, sn = 0.0f, si = 0.0f;
    int64_t np_ = 0, nn_ = 0;
    for (int64_t r0 = 0; r0 < P; r0 += 64) {
        unsigned long long bits = __ballot(r0 + lane < P && maskp[r0 + lane] == i);
        while (bits) {
            const int64_t r = r0 + __builtin_ctzll(bits);
            sp = sp + dist[r]; si = si + dist[P + Nn + r]; ++np_;
            bits &= bits - 1;
        }
    }
    for (int64_t r0 = 0; r0 < Nn; r0 += 64) {
        unsigned long long bits = __ballot(r0 + lane < Nn && maskn[r0 + lane] == i);
        while (bits) {
            const int64_t r = r0 + __builtin_ctzll(bits);
            sn = sn + dist[P + r]; ++nn_;
            bits &= bits - 1;
        }
    }
    if (lane) return;
    const float cp = sp / (float)np_, cn = sn / (float)nn_, ci = si / (float)np_;
    const float t1 = (cp - cn) + b, t2 = (cp - ci) + b;
    const float h1 = t1 > 0.0f ? t1 : (t1 != t1 ? t1 : 0.0f), h2 = t2 > 0.0f ? t2 : (t2 != t2 ? t2 : 0.0f);
    float *o = per_sample + (int64_t)i * 8;
    o[0] = cp; o[1] = cn; o[2] = ci; o[3] = t1; o[4] = t2; o[5] = (float)np_; o[6] = (float)nn_; o[7] = h1 + lamb * h2;
}

// loss[0] = sum_i loss_i, i ascending: the terms are fetched by the whole block, then added one after the other from LDS
__global__ __launch_bounds__(256) void ranking_sum_kernel(const float *__restrict__ per_sample, int S, float *__restrict__ loss)
{
    __shared__ float term[1024];
    float s = 0.0f;
    for (int base = 0; base < S; base += 1024) {
        const int n = S - base < 1024 ? S - base : 1024;
        for (int j = threadIdx.x; j < n; j += 256) term[j] = per_sample[(int64_t)(base + j) * 8 + 7];
        __syncthreads();
        if (threadIdx.x == 0)
            for (int j = 0; j < n; ++j) s = s + term[j];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = s;
}

// d loss / d c_x of sample i, scaled by the upstream gradient:  posit: g*(a1 + lamb*a2),  intra: -g*a1,  inter: -g*lamb*a2
__device__ __forceinline__ float set_coef(const float *o, int set, float gup, float lamb)
{
    const float a1 = o[3] > 0.0f ? 1.0f : 0.0f, a2 = o[4] > 0.0f ? 1.0f : 0.0f;
    return set == 0 ? gup * (a1 + lamb * a2) / o[5] : set == 1 ? -gup * a1 / o[6] : -gup * lamb * a2 / o[5];
}

__global__ __launch_bounds__(256) void ranking_grad_rows_kernel(const float *__restrict__ posit, const float *__restrict__ intra,
                                                                const float *__restrict__ inter, const float *__restrict__ lang,
                                                                const int64_t *__restrict__ maskp, const int64_t *__restrict__ maskn,
                                                                int64_t P, int64_t Nn, int S, int D, float eps, float lamb,
                                                                const float *__restrict__ dist, const float *__restrict__ per_sample,
                                                                const float *__restrict__ grad_loss, float *__restrict__ gposit,
                                                                float *__restrict__ gintra, float *__restrict__ ginter)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= 2 * P + Nn) return;
    const float *x; float *gx; int64_t i; int set;
    if (r < P) { x = posit + r * D; gx = gposit + r * D; i = maskp[r]; set = 0; }
    else if (r < P + Nn) { x = intra + (r - P) * D; gx = gintra + (r - P) * D; i = maskn[r - P]; set = 1; }
    else { x = inter + (r - P - Nn) * D; gx = ginter + (r - P - Nn) * D; i = maskp[r - P - Nn]; set = 2; }
    if (i < 0 || i >= S) { for (int k = 0; k < D; ++k) gx[k] = 0.0f; return; }
    const float w = set_coef(per_sample + i * 8, set, grad_loss[0], lamb) / dist[r];
    const float *l = lang + i * D;
    for (int k = 0; k < D; ++k) gx[k] = w * ((x[k] - l[k]) + eps);
}

// one block per sample, thread = embedding column: the block scans the masks 64 rows at a time (every wave the same ballot)
// and each thread adds its column's terms of the matching rows in row order -- posit and inter term of a row together, then
// the intra rows -- exactly the order of the one-thread-walks-all-rows form it replaces (122 us for 256 samples)
__global__ __launch_bounds__(128) void ranking_grad_lang_kernel(const float *__restrict__ posit, const float *__restrict__ intra,
                                                                const float *__restrict__ inter, const float *__restrict__ lang,
                                                                const int64_t *__restrict__ maskp, const int64_t *__restrict__ maskn,
                                                                int64_t P, int64_t Nn, int S, int D, float eps, float lamb,
                                                                const float *__restrict__ dist, const float *__restrict__ per_sample,
                                                                const float *__restrict__ grad_loss, float *__restrict__ glang)
{
    const int i = blockIdx.x, lane = threadIdx.x & 63;
    const float *o = per_sample + (int64_t)i * 8;
    const float gup = grad_loss[0];
    const float wp = set_coef(o, 0, gup, lamb), wn = set_coef(o, 1, gup, lamb), wi = set_coef(o, 2, gup, lamb);
    for (int k0 = 0; k0 < D; k0 += 128) {
        const int k = k0 + threadIdx.x, kc = k < D ? k : D - 1;
        const float lk = lang[(int64_t)i * D + kc];
        float acc = 0.0f;
        for (int64_t r0 = 0; r0 < P; r0 += 64) {
            unsigned long long bits = __ballot(r0 + lane < P && maskp[r0 + lane] == i);
            while (bits) {
                const int64_t r = r0 + __builtin_ctzll(bits);
                acc = acc - wp / dist[r] * ((posit[r * D + kc] - lk) + eps);
                acc = acc - wi / dist[P + Nn + r] * ((inter[r * D + kc] - lk) + eps);
                bits &= bits - 1;
            }
        }
        for (int64_t r0 = 0; r0 < Nn; r0 += 64) {
            unsigned long long bits = __ballot(r0 + lane < Nn && maskn[r0 + lane] == i);
            while (bits) {
                const int64_t r = r0 + __builtin_ctzll(bits);
                acc = acc - wn / dist[P + r] * ((intra[r * D + kc] - lk) + eps);
                bits &= bits - 1;
            }
        }
        if (k < D) glang[(int64_t)i * D + k] = acc;
    }
}

}  // namespace vfr

extern "C" {

size_t vfr_ranking_loss_workspace_bytes(int64_t P, int64_t Nn, int S)
{
    if (P < 0 || Nn < 0 || S < 0) return 0;
    return vfr::align_up((size_t)(2 * P + Nn) * sizeof(float), 256) + vfr::align_up((size_t)S * 8 * sizeof(float), 256);
}

int vfr_ranking_loss_f32(const float *posit, const float *intra, const float *inter, const float *lang, const int64_t *maskp,
                         const int64_t *maskn, int64_t P, int64_t Nn, int S, int D, float b, float lamb, float eps,
                         float *loss, void *workspace, size_t workspace_bytes, vfr_stream_t stream)
{
    VFR_REQUIRE(posit && intra && inter && lang && maskp && maskn && loss && P >= 0 && Nn >= 0 && S > 0 && D > 0, VFR_EINVAL,
                "vfr_ranking_loss_f32: bad argument");
    VFR_REQUIRE(workspace && workspace_bytes >= vfr_ranking_loss_workspace_bytes(P, Nn, S), VFR_EWORKSPACE,
                "vfr_ranking_loss_f32: workspace %zu < %zu bytes", workspace_bytes, vfr_ranking_loss_workspace_bytes(P, Nn, S));
    hipStream_t st = vfr::as_stream(stream);
    float *dist = static_cast<float *>(workspace);
    float *per = reinterpret_cast<float *>(static_cast<char *>(workspace) + vfr::align_up((size_t)(2 * P + Nn) * sizeof(float), 256));
    const int64_t rows = 2 * P + Nn;
    if (rows > 0)
        hipLaunchKernelGGL(vfr::ranking_row_dist_kernel, dim3((unsigned)vfr::cdiv(rows, 256)), dim3(256), 0, st, posit, intra, inter,
                           lang, maskp, maskn, P, Nn, S, D, eps, dist);
    hipLaunchKernelGGL(vfr::ranking_reduce_kernel, dim3((unsigned)vfr::cdiv(S, 4)), dim3(256), 0, st, dist, maskp, maskn, P, Nn, S, b, lamb, per);
    hipLaunchKernelGGL(vfr::ranking_sum_kernel, dim3(1), dim3(256), 0, st, per, S, loss);
    VFR_CHECK_LAUNCH("ranking_loss kernels");
    return VFR_OK;
}

int vfr_ranking_loss_grad_f32(const float *posit, const float *intra, const float *inter, const float *lang,
                              const int64_t *maskp, const int64_t *maskn, int64_t P, int64_t Nn, int S, int D, float lamb,
                              float eps, const float *grad_loss, const void *workspace, float *grad_posit, float *grad_intra,
                              float *grad_inter, float *grad_lang, vfr_stream_t stream)
{
    VFR_REQUIRE(posit && intra && inter && lang && maskp && maskn && grad_loss && workspace && grad_posit && grad_intra &&
                    grad_inter && grad_lang && P >= 0 && Nn >= 0 && S > 0 && D > 0,
                VFR_EINVAL, "vfr_ranking_loss_grad_f32: bad argument");
    hipStream_t st = vfr::as_stream(stream);
    const float *dist = static_cast<const float *>(workspace);
    const float *per = reinterpret_cast<const float *>(static_cast<const char *>(workspace) +
                                                       vfr::align_up((size_t)(2 * P + Nn) * sizeof(float), 256));
    const int64_t rows = 2 * P + Nn;
    if (rows > 0)
        hipLaunchKernelGGL(vfr::ranking_grad_rows_kernel, dim3((unsigned)vfr::cdiv(rows, 256)), dim3(256), 0, st, posit, intra,
                           inter, lang, maskp, maskn, P, Nn, S, D, eps, lamb, dist, per, grad_loss, grad_posit, grad_intra,
                           grad_inter);
    hipLaunchKernelGGL(vfr::ranking_grad_lang_kernel, dim3((unsigned)S), dim3(128), 0, st, posit, intra,
                       inter, lang, maskp, maskn, P, Nn, S, D, eps, lamb, dist, per, grad_loss, grad_lang);
    VFR_CHECK_LAUNCH("ranking_loss_grad kernels");
    return VFR_OK;
}

}  // extern "C"
