// f4: the ResNet-152 variant of the per-frame extractor (get_rgb_features.py:127-131): torchvision resnet152 with its last
// child (fc) removed = conv1 / bn1 / relu / maxpool, layer1..4 (Bottleneck x 3 / 8 / 36 / 3), avgpool -> [T, 2048].
//
// Canonical arithmetic (matches oracle/vfr_oracle.c, f4 section):
//   eval-mode BatchNorm folded into the convolution: invstd = 1 / sqrt(var + eps), alpha = gamma * invstd,
//   w' = w * alpha (per output channel), beta = fma(-mean, alpha, b);
//   conv: one fma chain per output over k = (ky*KW + kx)*Cin + ci ascending (tap-major, channel-minor), padding contributes
//   fma(0, w, acc) = acc; then v = acc + beta, v = v + identity (the block's input or its downsample branch), ReLU;
//   maxpool 3x3 / 2 (padding never wins); global average = row-major sum of the 7x7 plane / 49.
//
// Layout: NHWC activations like the VGG pipeline.  Every convolution is the MFMA chain GEMM of gemm.hip:
//   1x1 stride 1         dense GEMM on the activation itself ([B*H*W, Cin] x [Cout, Cin]^T)
//   3x3 stride 1 pad 1   the implicit-GEMM loader (same as VGG)
//   3x3 / 1x1 stride 2 and the 7x7 / 2 stem: an im2col pass ([B*Ho*Wo, KH*KW*Cin], zeros for padding) + the dense GEMM
// with bias (beta), residual and ReLU in the GEMM epilogue (EPI_BIAS | EPI_RES | EPI_RELU).
#include <vector>

#include "vfr_common.h"

namespace vfr {

// w [Cout, Cin, KH, KW] + bn [4][Cout] (gamma, b, mean, var) -> wf [Cout][KH*KW*Cinp] tap-major (zero for ci >= Cin), beta [Cout],
// for up to FOLD_BATCH convolutions in ONE launch (a ResNet-152 has 155: one launch each was 1.2 ms of launches for 0.1 ms of
// work): blockIdx.y = convolution (its fields are wave-uniform: scalar loads from the kernel arguments), blockIdx.x strides
// over its elements
constexpr int FOLD_BATCH = 64;
struct FoldBatch {
    const float *w[FOLD_BATCH], *bn[FOLD_BATCH];
    float *wf[FOLD_BATCH], *beta[FOLD_BATCH];
    unsigned short cout[FOLD_BATCH], cin[FOLD_BATCH], cinp[FOLD_BATCH], taps[FOLD_BATCH];
    int n;
    float eps;
};
__global__ __launch_bounds__(256) void bn_fold_repack_batch_kernel(FoldBatch b)
{
    const int j = blockIdx.y;
    const int Cout = b.cout[j], Cin = b.cin[j], Cinp = b.cinp[j], taps = b.taps[j], K = taps * Cinp;
    const float *__restrict__ w = b.w[j], *__restrict__ bn = b.bn[j];
    float *__restrict__ wf = b.wf[j], *__restrict__ beta = b.beta[j];
    // a block takes whole output channels: alpha once per channel (wave-uniform), no per-element division
    for (int co = blockIdx.x; co < Cout; co += gridDim.x) {
        const float invstd = 1.0f / __builtin_sqrtf(bn[3 * Cout + co] + b.eps);
        const float alpha = bn[co] * invstd;
        if (threadIdx.x == 0) beta[co] = __builtin_fmaf(-bn[2 * Cout + co], alpha, bn[Cout + co]);
        const float *wr = w + (int64_t)co * Cin * taps;
        float *dst = wf + (int64_t)co * K;
        for (int tap = 0; tap < taps; ++tap)
            for (int ci = threadIdx.x; ci < Cinp; ci += 256)
                dst[tap * Cinp + ci] = ci < Cin ? wr[(int64_t)ci * taps + tap] * alpha : 0.0f;
    }
}

// NHWC x [B, H, W, C] (C % 4 == 0) -> col [B*Ho*Wo, KH*KW*C]: column (tap, c) of output pixel (n, oy, ox) = x[n][oy*s+ky-p][ox*s+kx-p][c]
__global__ __launch_bounds__(256) void im2col_nhwc_kernel(const float *__restrict__ x, int64_t B, int H, int W, int C, int KH, int KW,
                                                          int stride, int pad, int Ho, int Wo, float *__restrict__ col)
{
    const int C4 = C / 4, taps = KH * KW;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;            // float4 index into col
    if (i >= B * Ho * Wo * (int64_t)taps * C4) return;
    const int c4 = (int)(i % C4);
    int64_t r = i / C4;
    const int tap = (int)(r % taps); r /= taps;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int64_t n = r / Ho;
    const int ky = tap / KW, kx = tap - ky * KW;
    const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = reinterpret_cast<const float4 *>(x)[((n * H + iy) * W + ix) * (int64_t)C4 + c4];
    reinterpret_cast<float4 *>(col)[i] = v;
}

__global__ __launch_bounds__(256) void maxpool3s2_nhwc_kernel(const float *__restrict__ x, int64_t B, int H, int W, int C, int Ho, int Wo,
                                                              float *__restrict__ y)
{
    const int C4 = C / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * Ho * Wo * (int64_t)C4) return;
    const int c4 = (int)(i % C4);
    int64_t r = i / C4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int64_t n = r / Ho;
    const float ninf = -__builtin_inff();
    float4 m = make_float4(ninf, ninf, ninf, ninf);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const float4 v = reinterpret_cast<const float4 *>(x)[((n * H + iy) * W + ix) * (int64_t)C4 + c4];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    reinterpret_cast<float4 *>(y)[i] = m;
}

// NHWC [B, HW, C] -> [B, C]: sequential sum over the plane in row-major order, / HW
__global__ __launch_bounds__(256) void global_avgpool_nhwc_kernel(const float *__restrict__ x, int64_t B, int HW, int C, float *__restrict__ y)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int c = (int)(i % C);
    const int64_t n = i / C;
    float acc = 0.0f;
    for (int p = 0; p < HW; ++p) acc = acc + x[(n * HW + p) * (int64_t)C + c];
    y[i] = acc / (float)HW;
}

// THWC uint8 -> NHWC fp32 with a zero 4th channel, ((x / 255) - mean[c]) / std[c] (get_rgb_features.py:64-69; as vgg.hip)
__global__ __launch_bounds__(256) void resnet_normalize_nhwc4_kernel(const uint8_t *__restrict__ in, int64_t pixels, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pixels) return;
    float4 o;
    float v;
    v = (float)in[i * 3 + 0]; v = v / 255.0f; v = v - 0.485f; o.x = v / 0.229f;
    v = (float)in[i * 3 + 1]; v = v / 255.0f; v = v - 0.456f; o.y = v / 0.224f;
    v = (float)in[i * 3 + 2]; v = v / 255.0f; v = v - 0.406f; o.z = v / 0.225f;
    o.w = 0.0f;
    reinterpret_cast<float4 *>(out)[i] = o;
}

struct ResConv { int cin, cout, k, stride, pad; };
struct ResPlan {
    std::vector<ResConv> convs;          // execution order: stem, then per block conv1, conv2, conv3 [, downsample]
    std::vector<int> down;               // per block: index of its downsample conv or -1
    size_t act_elems, col_elems, wf_elems, beta_elems;
    int h_out, w_out, c_out;
    bool ok;
};
static int conv_out(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }
static ResPlan plan_resnet(int chunk, int H, int W, const int *blocks, int width)
{
    ResPlan p{};
    p.ok = width > 0 && width % 4 == 0 && H >= 32 && W >= 32;
    auto add = [&](int cin, int cout, int k, int s, int pad) { p.convs.push_back({cin, cout, k, s, pad}); };
    add(3, width, 7, 2, 3);
    int h = conv_out(H, 7, 2, 3), w = conv_out(W, 7, 2, 3);
    size_t act = (size_t)chunk * h * w * width, col = (size_t)chunk * h * w * 49 * 4;
    h = conv_out(h, 3, 2, 1); w = conv_out(w, 3, 2, 1);
    int cin = width;
    for (int li = 0; li < 4; ++li) {
        const int mid = width << li;
        if (blocks[li] <= 0) p.ok = false;
        for (int b = 0; b < blocks[li]; ++b) {
            const int s = (b == 0 && li > 0) ? 2 : 1;
            add(cin, mid, 1, 1, 0);
            add(mid, mid, 3, s, 1);
            add(mid, 4 * mid, 1, 1, 0);
            const int ho = conv_out(h, 3, s, 1), wo = conv_out(w, 3, s, 1);
            if (b == 0) { p.down.push_back((int)p.convs.size()); add(cin, 4 * mid, 1, s, 0); } else p.down.push_back(-1);
            size_t a1 = (size_t)chunk * h * w * (size_t)(cin > mid ? cin : mid), a2 = (size_t)chunk * ho * wo * 4 * mid;
            act = a1 > act ? a1 : act; act = a2 > act ? a2 : act;
            if (s == 2) {
                size_t c3 = (size_t)chunk * ho * wo * 9 * mid, c1 = (size_t)chunk * ho * wo * cin;
                col = c3 > col ? c3 : col; col = c1 > col ? c1 : col;
            }
            h = ho; w = wo; cin = 4 * mid;
        }
    }
    if (h < 1 || w < 1) p.ok = false;
    p.h_out = h; p.w_out = w; p.c_out = cin;
    p.act_elems = act; p.col_elems = col;
    for (const ResConv &c : p.convs) {
        const int cinp = c.cin < 4 ? 4 : c.cin;
        p.wf_elems += align_up((size_t)c.cout * c.k * c.k * cinp, 64);
        p.beta_elems += align_up((size_t)c.cout, 64);
    }
    return p;
}
constexpr int RESNET_FRAME_CHUNK = 160;     // (64: layer3 GEMMs of 12 544 rows = half a round of 128-row tiles; 160: one DiDeMo video per pass)
static int resnet_chunk(int T)
{
    if (T <= RESNET_FRAME_CHUNK) return T;
    const int n = (T + RESNET_FRAME_CHUNK - 1) / RESNET_FRAME_CHUNK;
    return (T + n - 1) / n;
}

}  // namespace vfr

extern "C" {

size_t vfr_resnet_pool_workspace_bytes(int T, int H, int W, const int *blocks_host, int width)
{
    if (T < 0 || H <= 0 || W <= 0 || !blocks_host || width <= 0) return 0;
    const int chunk = vfr::resnet_chunk(T);
    vfr::ResPlan p = vfr::plan_resnet(chunk, H, W, blocks_host, width);
    const size_t x0 = vfr::align_up((size_t)chunk * H * W * 4 * sizeof(float), 256);
    return x0 + 4 * vfr::align_up(p.act_elems * sizeof(float), 256) + vfr::align_up(p.col_elems * sizeof(float), 256) +
           vfr::align_up(p.wf_elems * sizeof(float), 256) + vfr::align_up(p.beta_elems * sizeof(float), 256);
}

size_t vfr_resnet_folded_bytes(const int *blocks_host, int width)
{
    if (!blocks_host || width <= 0) return 0;
    vfr::ResPlan p = vfr::plan_resnet(1, 224, 224, blocks_host, width);
    return vfr::align_up(p.wf_elems * sizeof(float), 256) + vfr::align_up(p.beta_elems * sizeof(float), 256);
}

}  // extern "C"

namespace vfr {

// fold + repack every convolution of the plan into [wf region | beta region] at `folded` (FOLD_BATCH convolutions per launch)
static int resnet_fold(const ResPlan &p, const float *const *conv_w_host, const float *const *bn_host, float bn_eps, void *folded,
                       std::vector<float *> &wf, std::vector<float *> &beta, bool launch, hipStream_t st)
{
    float *wq = static_cast<float *>(folded);
    float *bq = reinterpret_cast<float *>(static_cast<char *>(folded) + align_up(p.wf_elems * sizeof(float), 256));
    wf.resize(p.convs.size()); beta.resize(p.convs.size());
    FoldBatch fb{};
    fb.eps = bn_eps;
    unsigned most = 0;
    auto flush = [&]() {
        if (fb.n == 0) return;
        unsigned blocks = most;                                   // output channels of the widest convolution, capped
        blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
        hipLaunchKernelGGL(bn_fold_repack_batch_kernel, dim3(blocks, (unsigned)fb.n), dim3(256), 0, st, fb);
        fb.n = 0; most = 0;
    };
    for (size_t i = 0; i < p.convs.size(); ++i) {
        const ResConv &c = p.convs[i];
        const int cinp = c.cin < 4 ? 4 : c.cin, taps = c.k * c.k;
        const int64_t n = (int64_t)c.cout * taps * cinp;
        wf[i] = wq; beta[i] = bq;
        if (launch) {
            VFR_REQUIRE(conv_w_host[i] && bn_host[i], VFR_EINVAL, "vfr_resnet: null weight pointer for convolution %zu", i);
            VFR_REQUIRE(n < (1ll << 31) && c.cout < 65536 && cinp < 65536, VFR_EUNSUPPORTED, "vfr_resnet: convolution %zu is too large", i);
            if (fb.n == FOLD_BATCH) flush();
            const int j = fb.n++;
            fb.w[j] = conv_w_host[i]; fb.bn[j] = bn_host[i]; fb.wf[j] = wq; fb.beta[j] = bq;
            fb.cout[j] = (unsigned short)c.cout; fb.cin[j] = (unsigned short)c.cin; fb.cinp[j] = (unsigned short)cinp; fb.taps[j] = (unsigned short)taps;
            most = (unsigned)c.cout > most ? (unsigned)c.cout : most;
        }
        wq += align_up((size_t)n, 64); bq += align_up((size_t)c.cout, 64);
    }
    if (launch) {
        ProfScope prof(SITE_REPACK, st);
        flush();
        VFR_CHECK_LAUNCH("bn_fold_repack_batch_kernel");
    }
    return VFR_OK;
}

// the stack on folded weights: wf / beta per convolution; workspace = [x0 | 4 activation buffers | im2col buffer]
static int resnet_run(const ResPlan &p, const uint8_t *frames_thwc, int T, int chunk, int H, int W, const int *blocks_host, int width,
                      const std::vector<float *> &wf, const std::vector<float *> &beta, float *out, char *base, hipStream_t st)
{
    auto carve = [&](size_t bytes) { char *q = base; base += align_up(bytes, 256); return reinterpret_cast<float *>(q); };
    float *x0 = carve((size_t)chunk * H * W * 4 * sizeof(float));
    float *buf[4];
    for (int i = 0; i < 4; ++i) buf[i] = carve(p.act_elems * sizeof(float));
    float *col = carve(p.col_elems * sizeof(float));

    // one convolution: x [bt, h, w, cinp] -> y [bt, ho, wo, cout]
    auto conv = [&](size_t ci, const float *x, int bt, int h, int w, float *y, const float *res, bool relu) -> int {
        const ResConv &c = p.convs[ci];
        const int cinp = c.cin < 4 ? 4 : c.cin, ho = conv_out(h, c.k, c.stride, c.pad), wo = conv_out(w, c.k, c.stride, c.pad);
        GemmArgs g{};
        g.W = wf[ci]; g.out = y; g.ldo = c.cout; g.M = (int64_t)bt * ho * wo; g.N = c.cout; g.bias = beta[ci];
        g.epi = EPI_BIAS | (res ? EPI_RES : 0) | (relu ? EPI_RELU : 0); g.res = res; g.ldr = c.cout; g.site = SITE_CONV;
        if (c.k == 1 && c.stride == 1) {
            g.A = x; g.lda = cinp; g.K = cinp; g.ldw = cinp;
        } else if (c.k == 3 && c.stride == 1 && c.pad == 1) {
            g.A = x; g.K = 9 * cinp; g.ldw = 9 * cinp; g.conv_h = h; g.conv_w = w; g.conv_cin = cinp;
        } else {
            const int K = c.k * c.k * cinp;
            const int64_t n4 = g.M * (K / 4);
            {
                ProfScope prof(SITE_REPACK, st);
                hipLaunchKernelGGL(im2col_nhwc_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, st, x, (int64_t)bt, h, w, cinp, c.k, c.k,
                                   c.stride, c.pad, ho, wo, col);
            }
            g.A = col; g.lda = K; g.K = K; g.ldw = K;
        }
        return gemm_nt(g, st);
    };

    for (int t0 = 0; t0 < T; t0 += chunk) {
        const int bt = (T - t0) < chunk ? (T - t0) : chunk;
        {
            ProfScope prof(SITE_NORMALIZE, st);
            const int64_t pixels = (int64_t)bt * H * W;
            hipLaunchKernelGGL(resnet_normalize_nhwc4_kernel, dim3((unsigned)cdiv(pixels, 256)), dim3(256), 0, st,
                               frames_thwc + (size_t)t0 * H * W * 3, pixels, x0);
        }
        int h = conv_out(H, 7, 2, 3), w = conv_out(W, 7, 2, 3);
        if (int rc = conv(0, x0, bt, H, W, buf[0], nullptr, true)) return rc;
        const int hp = conv_out(h, 3, 2, 1), wp = conv_out(w, 3, 2, 1);
        {
            ProfScope prof(SITE_POOL2D, st);
            const int64_t n = (int64_t)bt * hp * wp * (width / 4);
            hipLaunchKernelGGL(maxpool3s2_nhwc_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, buf[0], (int64_t)bt, h, w, width, hp, wp, buf[1]);
        }
        VFR_CHECK_LAUNCH("resnet stem");
        h = hp; w = wp;
        int xi = 1;                                                      // buf[xi] holds the block input
        size_t ci = 1, blk = 0;
        for (int li = 0; li < 4; ++li)
            for (int b = 0; b < blocks_host[li]; ++b, ++blk) {
                const int s = p.convs[ci + 1].stride;
                const int ho = conv_out(h, 3, s, 1), wo = conv_out(w, 3, s, 1);
                float *x = buf[xi], *t1 = buf[(xi + 1) & 3], *t2 = buf[(xi + 2) & 3], *idb = buf[(xi + 3) & 3];
                const float *identity = x;
                if (p.down[blk] >= 0) {
                    if (int rc = conv((size_t)p.down[blk], x, bt, h, w, idb, nullptr, false)) return rc;
                    identity = idb;
                }
                if (int rc = conv(ci, x, bt, h, w, t1, nullptr, true)) return rc;
                if (int rc = conv(ci + 1, t1, bt, h, w, t2, nullptr, true)) return rc;
                if (int rc = conv(ci + 2, t2, bt, ho, wo, t1, identity, true)) return rc;   // t1 is free again: the block output
                xi = (xi + 1) & 3;
                ci += p.down[blk] >= 0 ? 4 : 3;
                h = ho; w = wo;
            }
        {
            ProfScope prof(SITE_POOL2D, st);
            hipLaunchKernelGGL(global_avgpool_nhwc_kernel, dim3((unsigned)cdiv((int64_t)bt * p.c_out, 256)), dim3(256), 0, st, buf[xi], (int64_t)bt,
                               h * w, p.c_out, out + (size_t)t0 * p.c_out);
        }
        VFR_CHECK_LAUNCH("resnet stack");
    }
    return VFR_OK;
}

}  // namespace vfr

extern "C" {

int vfr_resnet_fold_f32(const int *blocks_host, int width, const float *const *conv_w_host, const float *const *bn_host, float bn_eps,
                        void *folded, size_t folded_bytes, vfr_stream_t stream)
{
    using namespace vfr;
    VFR_REQUIRE(blocks_host && conv_w_host && bn_host && folded && width > 0, VFR_EINVAL, "vfr_resnet_fold_f32: bad argument");
    ResPlan p = plan_resnet(1, 224, 224, blocks_host, width);
    VFR_REQUIRE(p.ok, VFR_EUNSUPPORTED, "vfr_resnet_fold_f32: needs a width that is a multiple of 4 and positive block counts");
    VFR_REQUIRE(folded_bytes >= vfr_resnet_folded_bytes(blocks_host, width), VFR_EWORKSPACE, "vfr_resnet_fold_f32: buffer %zu < %zu bytes",
                folded_bytes, vfr_resnet_folded_bytes(blocks_host, width));
    std::vector<float *> wf, beta;
    return resnet_fold(p, conv_w_host, bn_host, bn_eps, folded, wf, beta, true, as_stream(stream));
}

int vfr_resnet_pool_folded_f32(const uint8_t *frames_thwc, int T, int H, int W, const int *blocks_host, int width, const void *folded,
                               size_t folded_bytes, float *out, void *workspace, size_t workspace_bytes, vfr_stream_t stream)
{
    using namespace vfr;
    VFR_REQUIRE(frames_thwc && blocks_host && folded && out && T >= 0 && H > 0 && W > 0 && width > 0, VFR_EINVAL,
                "vfr_resnet_pool_folded_f32: bad argument");
    if (T == 0) return VFR_OK;
    const int chunk = resnet_chunk(T);
    ResPlan p = plan_resnet(chunk, H, W, blocks_host, width);
    VFR_REQUIRE(p.ok, VFR_EUNSUPPORTED, "vfr_resnet_pool_folded_f32: needs a width that is a multiple of 4, positive block counts and frames of at least 32x32");
    VFR_REQUIRE(folded_bytes >= vfr_resnet_folded_bytes(blocks_host, width), VFR_EWORKSPACE, "vfr_resnet_pool_folded_f32: folded buffer %zu < %zu bytes",
                folded_bytes, vfr_resnet_folded_bytes(blocks_host, width));
    VFR_REQUIRE(workspace && workspace_bytes >= vfr_resnet_pool_workspace_bytes(T, H, W, blocks_host, width), VFR_EWORKSPACE,
                "vfr_resnet_pool_folded_f32: workspace %zu < %zu bytes", workspace_bytes, vfr_resnet_pool_workspace_bytes(T, H, W, blocks_host, width));
    std::vector<float *> wf, beta;
    if (int rc = resnet_fold(p, nullptr, nullptr, 0.0f, const_cast<void *>(folded), wf, beta, false, as_stream(stream))) return rc;
    return resnet_run(p, frames_thwc, T, chunk, H, W, blocks_host, width, wf, beta, out, static_cast<char *>(workspace), as_stream(stream));
}

int vfr_resnet_pool_f32(const uint8_t *frames_thwc, int T, int H, int W, const int *blocks_host, int width,
                        const float *const *conv_w_host, const float *const *bn_host, float bn_eps, float *out, void *workspace,
                        size_t workspace_bytes, vfr_stream_t stream)
{
    using namespace vfr;
    VFR_REQUIRE(frames_thwc && blocks_host && conv_w_host && bn_host && out && T >= 0 && H > 0 && W > 0 && width > 0, VFR_EINVAL,
                "vfr_resnet_pool_f32: bad argument");
    if (T == 0) return VFR_OK;
    const int chunk = resnet_chunk(T);
    ResPlan p = plan_resnet(chunk, H, W, blocks_host, width);
    VFR_REQUIRE(p.ok, VFR_EUNSUPPORTED, "vfr_resnet_pool_f32: needs a width that is a multiple of 4, positive block counts and frames of at least 32x32");
    VFR_REQUIRE(workspace && workspace_bytes >= vfr_resnet_pool_workspace_bytes(T, H, W, blocks_host, width), VFR_EWORKSPACE,
                "vfr_resnet_pool_f32: workspace %zu < %zu bytes", workspace_bytes, vfr_resnet_pool_workspace_bytes(T, H, W, blocks_host, width));
    hipStream_t st = as_stream(stream);
    // the folded weights live at the END of the workspace (fold + repack of every convolution, every call: callers that keep the
    // model resident fold once with vfr_resnet_fold_f32 and use vfr_resnet_pool_folded_f32)
    const size_t fbytes = vfr_resnet_folded_bytes(blocks_host, width);
    char *fold_at = static_cast<char *>(workspace) + vfr_resnet_pool_workspace_bytes(T, H, W, blocks_host, width) - fbytes;
    std::vector<float *> wf, beta;
    if (int rc = resnet_fold(p, conv_w_host, bn_host, bn_eps, fold_at, wf, beta, true, st)) return rc;
    return resnet_run(p, frames_thwc, T, chunk, H, W, blocks_host, width, wf, beta, out, static_cast<char *>(workspace), st);
}

}  // extern "C"
