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

// per_sample [S,8] = c_posit, c_intra, c_inter, t1, t2, n_posit, n_intra, loss_i ;  loss[0] = sum_i loss_i (in order)
__global__ __launch_bounds__(256) void ranking_reduce_kernel(const float *__restrict__ dist, const int64_t *__restrict__ maskp,
                                                             const int64_t *__restrict__ maskn, int64_t P, int64_t Nn, int S,
                                                             float b, float lamb, float *__restrict__ per_sample,
                                                             float *__restrict__ loss)
{
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        float sp = 0.0f, sn = 0.0f, si = 0.0f;
        int64_t np_ = 0, nn_ = 0;
        for (int64_t r = 0; r < P; ++r)
            if (maskp[r] == i) { sp = sp + dist[r]; si = si + dist[P + Nn + r]; ++np_; }
        for (int64_t r = 0; r < Nn; ++r)
            if (maskn[r] == i) { sn = sn + dist[P + r]; ++nn_; }
        const float cp = sp / (float)np_, cn = sn / (float)nn_, ci = si / (float)np_;
        const float t1 = (cp - cn) + b, t2 = (cp - ci) + b;
        const float h1 = t1 > 0.0f ? t1 : (t1 != t1 ? t1 : 0.0f), h2 = t2 > 0.0f ? t2 : (t2 != t2 ? t2 : 0.0f);
        float *o = per_sample + (int64_t)i * 8;
        o[0] = cp; o[1] = cn; o[2] = ci; o[3] = t1; o[4] = t2; o[5] = (float)np_; o[6] = (float)nn_; o[7] = h1 + lamb * h2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int i = 0; i < S; ++i) s = s + per_sample[(int64_t)i * 8 + 7];
        loss[0] = s;
    }
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

__global__ __launch_bounds__(256) void ranking_grad_lang_kernel(const float *__restrict__ posit, const float *__restrict__ intra,
                                                                const float *__restrict__ inter, const float *__restrict__ lang,
                                                                const int64_t *__restrict__ maskp, const int64_t *__restrict__ maskn,
                                                                int64_t P, int64_t Nn, int S, int D, float eps, float lamb,
                                                                const float *__restrict__ dist, const float *__restrict__ per_sample,
                                                                const float *__restrict__ grad_loss, float *__restrict__ glang)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)S * D) return;
    const int i = (int)(t / D), k = (int)(t - (int64_t)i * D);
    const float *o = per_sample + (int64_t)i * 8;
    const float gup = grad_loss[0], lk = lang[t];
    const float wp = set_coef(o, 0, gup, lamb), wn = set_coef(o, 1, gup, lamb), wi = set_coef(o, 2, gup, lamb);
    float acc = 0.0f;
    for (int64_t r = 0; r < P; ++r)
        if (maskp[r] == i) {
            acc = acc - wp / dist[r] * ((posit[r * D + k] - lk) + eps);
            acc = acc - wi / dist[P + Nn + r] * ((inter[r * D + k] - lk) + eps);
        }
    for (int64_t r = 0; r < Nn; ++r)
        if (maskn[r] == i) acc = acc - wn / dist[P + r] * ((intra[r * D + k] - lk) + eps);
    glang[t] = acc;
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
    hipLaunchKernelGGL(vfr::ranking_reduce_kernel, dim3(1), dim3(256), 0, st, dist, maskp, maskn, P, Nn, S, b, lamb, per, loss);
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
    hipLaunchKernelGGL(vfr::ranking_grad_lang_kernel, dim3((unsigned)vfr::cdiv((int64_t)S * D, 256)), dim3(256), 0, st, posit, intra,
                       inter, lang, maskp, maskn, P, Nn, S, D, eps, lamb, dist, per, grad_loss, grad_lang);
    VFR_CHECK_LAUNCH("ranking_loss_grad kernels");
    return VFR_OK;
}

}  // extern "C"
