// a4 + a7: clip encoder = make_visual_features (model/data.py:204-213) + CALModel.visual_fc
// (model/models.py:21-26,55-56), factored so the [n, 2F+2] concat never exists in HBM:
//
//   S  [C, hid]  = seg [C, F]  x W1[:, 0:F]^T          chain GEMM, streams the clip features once
//   Cx [Nv, hid] = ctx [Nv, F] x W1[:, F:2F]^T         chain GEMM, once per VIDEO (not per clip)
//   h  = relu(((S + Cx[v]) + (te0*W1[:,2F] (+) te1*W1[:,2F+1])) + b1)      fused row epilogue
//   out[C, D]    = h x W2^T + b2                        chain GEMM
//
// Algorithmic bytes per video: 4*(n+1)*F (features) + weights once; FLOP: n*2*(F+2)*hid + 2*F*hid
// + n*2*hid*D  (factored figure).
#include "vfr_common.h"

namespace vfr {

__global__ __launch_bounds__(256) void visual_hidden_kernel(float *__restrict__ S, const float *__restrict__ Cx,
                                                            const int32_t *__restrict__ clip_off, int Nv,
                                                            int total_clips, int hid, const float *__restrict__ W1,
                                                            int ldw, int F, const float *__restrict__ b1)
{
    const int row = blockIdx.x;                 // one clip row per block
    if (row >= total_clips) return;
    int lo = 0, hi = Nv;                        // largest v with clip_off[v] <= row (uniform: scalar loads)
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (clip_off[mid] <= row) lo = mid; else hi = mid;
    }
    const int v = lo, c0 = clip_off[v], n = clip_off[v + 1] - c0, t = row - c0;
    const float te0 = (float)t / (float)n, te1 = (float)(t + 1) / (float)n;
    for (int j = threadIdx.x; j < hid; j += blockDim.x) {
        float te = __builtin_fmaf(te1, W1[(int64_t)j * ldw + 2 * F + 1],
                                  __builtin_fmaf(te0, W1[(int64_t)j * ldw + 2 * F], 0.0f));
        float x = ((S[(int64_t)row * hid + j] + Cx[(int64_t)v * hid + j]) + te) + b1[j];
        S[(int64_t)row * hid + j] = x > 0.0f ? x : 0.0f;
    }
}

// per clip row: its video and the temporal endpoint features (t/n, (t+1)/n); per hidden unit: the two endpoint weights
__global__ __launch_bounds__(256) void visual_rowinfo_kernel(const int32_t *__restrict__ clip_off, int Nv, int total_clips,
                                                             int hid, const float *__restrict__ W1, int ldw, int F,
                                                             int *__restrict__ row_vid, float *__restrict__ row_te,
                                                             float *__restrict__ w0, float *__restrict__ w1)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < hid) { w0[i] = W1[(int64_t)i * ldw + 2 * F]; w1[i] = W1[(int64_t)i * ldw + 2 * F + 1]; }
    if (i >= total_clips) return;
    int lo = 0, hi = Nv;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (clip_off[mid] <= i) lo = mid; else hi = mid;
    }
    const int c0 = clip_off[lo], n = clip_off[lo + 1] - c0, t = i - c0;
    row_vid[i] = lo;
    row_te[2 * i] = (float)t / (float)n;
    row_te[2 * i + 1] = (float)(t + 1) / (float)n;
}

}  // namespace vfr

extern "C" {

size_t vfr_visual_mlp_workspace_bytes(int total_clips, int Nv, int F, int hid)
{
    if (total_clips < 0 || Nv < 0 || hid < 0 || F < 0) return 0;
    return vfr::align_up((size_t)total_clips * hid * sizeof(float), 256) +
           vfr::align_up((size_t)Nv * hid * sizeof(float), 256) + 2 * vfr::align_up((size_t)hid * F * sizeof(float), 256) +
           vfr::align_up((size_t)total_clips * 3 * sizeof(float), 256) + vfr::align_up((size_t)2 * hid * sizeof(float), 256);
}

int vfr_visual_mlp_f32(const float *seg, const float *ctx, const int32_t *clip_offsets, int Nv, int total_clips,
                       int F, const float *W1, const float *b1, const float *W2, const float *b2, int hid, int D,
                       float *out, void *workspace, size_t workspace_bytes, vfr_stream_t stream)
{
    VFR_REQUIRE(seg && ctx && clip_offsets && W1 && b1 && W2 && b2 && out && Nv >= 0 && total_clips >= 0 && F > 0 &&
                    hid > 0 && D > 0,
                VFR_EINVAL, "vfr_visual_mlp_f32: bad argument");
    if (total_clips == 0) return VFR_OK;
    VFR_REQUIRE(workspace && workspace_bytes >= vfr_visual_mlp_workspace_bytes(total_clips, Nv, F, hid), VFR_EWORKSPACE,
                "vfr_visual_mlp_f32: workspace %zu < %zu bytes", workspace_bytes,
                vfr_visual_mlp_workspace_bytes(total_clips, Nv, F, hid));
    hipStream_t st = vfr::as_stream(stream);
    float *S = static_cast<float *>(workspace);
    float *Cx = reinterpret_cast<float *>(static_cast<char *>(workspace) +
                                          vfr::align_up((size_t)total_clips * hid * sizeof(float), 256));
    const int ldw = 2 * F + 2;
    // W1's rows are 2F+2 floats, so neither half starts 16-byte aligned: copy [W1_seg | W1_ctx] into dense
    // aligned buffers (2 * hid * F floats, ~16 MB) so the GEMM stages them with 16-byte loads
    char *wsb = static_cast<char *>(workspace);
    float *Wseg = reinterpret_cast<float *>(wsb + vfr::align_up((size_t)total_clips * hid * sizeof(float), 256) +
                                            vfr::align_up((size_t)Nv * hid * sizeof(float), 256));
    float *Wctx = reinterpret_cast<float *>(reinterpret_cast<char *>(Wseg) + vfr::align_up((size_t)hid * F * sizeof(float), 256));
    if (int rc = vfr::repack_rows(W1, ldw, hid, F, Wseg, st)) return rc;
    if (int rc = vfr::repack_rows(W1 + F, ldw, hid, F, Wctx, st)) return rc;
    char *tail = reinterpret_cast<char *>(Wctx) + vfr::align_up((size_t)hid * F * sizeof(float), 256);
    int *row_vid = reinterpret_cast<int *>(tail);
    float *row_te = reinterpret_cast<float *>(tail) + total_clips;
    float *w0 = reinterpret_cast<float *>(tail + vfr::align_up((size_t)total_clips * 3 * sizeof(float), 256)), *w1 = w0 + hid;
    vfr::GemmArgs g{};
    g.A = ctx; g.lda = F; g.W = Wctx; g.ldw = F; g.out = Cx; g.ldo = hid; g.M = Nv; g.N = hid; g.K = F;
    g.site = vfr::SITE_GEMM_VIS_CTX;
    if (int rc = vfr::gemm_nt(g, st)) return rc;                        // per-video context chains first: the seg GEMM's epilogue adds them
    g.A = seg; g.W = Wseg; g.out = S; g.M = total_clips; g.site = vfr::SITE_GEMM_VIS_SEG;
    if (vfr::opt_gemm() != 0) {
        {
        vfr::ProfScope prof(vfr::SITE_VIS_HIDDEN, st);
        const int nthr = total_clips > hid ? total_clips : hid;
        hipLaunchKernelGGL(vfr::visual_rowinfo_kernel, dim3((unsigned)vfr::cdiv(nthr, 256)), dim3(256), 0, st, clip_offsets, Nv,
                           total_clips, hid, W1, ldw, F, row_vid, row_te, w0, w1);
        }
        VFR_CHECK_LAUNCH("visual_rowinfo_kernel");
        g.epi = vfr::EPI_VIS; g.bias = b1; g.vis_row = row_vid; g.vis_te = row_te; g.vis_cx = Cx; g.vis_w0 = w0; g.vis_w1 = w1;
        if (int rc = vfr::gemm_nt(g, st)) return rc;                    // S = relu(hidden) straight out of the MFMA epilogue
    } else {
        if (int rc = vfr::gemm_nt(g, st)) return rc;
        {
        vfr::ProfScope prof(vfr::SITE_VIS_HIDDEN, st);
        hipLaunchKernelGGL(vfr::visual_hidden_kernel, dim3(total_clips), dim3(256), 0, st, S, Cx, clip_offsets, Nv,
                           total_clips, hid, W1, ldw, F, b1);
        }
        VFR_CHECK_LAUNCH("visual_hidden_kernel");
    }
    vfr::GemmArgs g2{};
    g2.A = S; g2.lda = hid; g2.W = W2; g2.ldw = hid; g2.out = out; g2.ldo = D; g2.M = total_clips; g2.N = D;
    g2.K = hid; g2.bias = b2; g2.epi = vfr::EPI_BIAS; g2.site = vfr::SITE_GEMM_VIS_OUT;
    return vfr::gemm_nt(g2, st);
}

}  // extern "C"
