// a1 + a2: per-frame RGB feature extractor = ImageNet normalisation (get_rgb_features.py:64-69) +
// torchvision VGG-19 "E" features / avgpool / classifier[0..4] (get_rgb_features.py:122-126).
//
// Canonical arithmetic (matches oracle/vfr_oracle.c):
//   normalise : ((u8 / 255) - mean[c]) / std[c]
//   conv3x3   : one fma chain per output over k = (ky*3 + kx)*Cin + ci ascending (tap-major, channel-minor:
//               what the NHWC implicit GEMM consumes), zero padding contributes fma(0, w, acc) = acc; + bias; ReLU
//   maxpool   : max of the 2x2 window;  adaptive avgpool 7x7: row-major window sum / count
//   fc6 / fc7 : chain GEMM + bias + ReLU (gemm.hip)
#include "vfr_common.h"

namespace vfr {

__constant__ float c_mean[3] = {0.485f, 0.456f, 0.406f};
__constant__ float c_std[3] = {0.229f, 0.224f, 0.225f};

__global__ __launch_bounds__(256) void frames_normalize_kernel(const uint8_t *__restrict__ in, int64_t total, int H,
                                                               int W, float *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // index into TCHW output
    if (i >= total) return;
    int x = (int)(i % W);
    int64_t r = i / W;
    int y = (int)(r % H); r /= H;
    int c = (int)(r % 3);
    int64_t t = r / 3;
    float v = (float)in[((t * H + y) * W + x) * 3 + c];
    v = v / 255.0f;
    v = v - c_mean[c];
    out[i] = v / c_std[c];
}

// direct convolution: one thread per output pixel, output channels tiled 8 per thread for input reuse
constexpr int CO_T = 8;
__global__ __launch_bounds__(256) void conv3x3_relu_kernel(const float *__restrict__ x, int B, int Cin, int H, int W,
                                                           const float *__restrict__ w, const float *__restrict__ b,
                                                           int Cout, float *__restrict__ y)
{
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*H*W
    const int co0 = blockIdx.y * CO_T;
    if (pix >= (int64_t)B * H * W) return;
    const int ox = (int)(pix % W);
    const int oy = (int)((pix / W) % H);
    const int n = (int)(pix / ((int64_t)W * H));
    float acc[CO_T];
#pragma unroll
    for (int j = 0; j < CO_T; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy + ky - 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox + kx - 1;
            const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
            for (int ci = 0; ci < Cin; ++ci) {            // canonical chain order: (ky, kx, ci) ascending
                const float xv = in ? x[(((int64_t)n * Cin + ci) * H + iy) * W + ix] : 0.0f;
#pragma unroll
                for (int j = 0; j < CO_T; ++j) {
                    const int co = co0 + j;
                    const float wv = co < Cout ? w[(((int64_t)co * Cin + ci) * 3 + ky) * 3 + kx] : 0.0f;
                    acc[j] = __builtin_fmaf(xv, wv, acc[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CO_T; ++j) {
        const int co = co0 + j;
        if (co >= Cout) continue;
        float v = acc[j] + b[co];
        y[(((int64_t)n * Cout + co) * H + oy) * W + ox] = v > 0.0f ? v : 0.0f;
    }
}

__global__ __launch_bounds__(256) void maxpool2_kernel(const float *__restrict__ x, int64_t planes, int H, int W,
                                                       float *__restrict__ y)
{
    const int Ho = H / 2, Wo = W / 2;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ho * Wo) return;
    int ox = (int)(i % Wo);
    int oy = (int)((i / Wo) % Ho);
    int64_t p = i / ((int64_t)Wo * Ho);
    const float *xp = x + p * H * W + (int64_t)(2 * oy) * W + 2 * ox;
    y[i] = fmaxf(fmaxf(xp[0], xp[1]), fmaxf(xp[W], xp[W + 1]));
}

__global__ __launch_bounds__(256) void adaptive_avgpool7_kernel(const float *__restrict__ x, int64_t planes, int H,
                                                                int W, float *__restrict__ y)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * 49) return;
    int ox = (int)(i % 7), oy = (int)((i / 7) % 7);
    int64_t p = i / 49;
    int y0 = (oy * H) / 7, y1 = ((oy + 1) * H + 6) / 7, x0 = (ox * W) / 7, x1 = ((ox + 1) * W + 6) / 7;
    float acc = 0.0f;
    for (int iy = y0; iy < y1; ++iy)
        for (int ix = x0; ix < x1; ++ix) acc = acc + x[p * H * W + (int64_t)iy * W + ix];
    y[i] = acc / (float)((y1 - y0) * (x1 - x0));
}


// ------------------------------------------------------------------------------------------------
// NHWC pipeline of the whole stack (vfr_vgg_fc7_f32): the input frames are THWC already, every conv is an
// implicit GEMM on the MFMA chain kernel (gemm.hip, conv loader), activations stay channel-minor so the
// A-operand gathers are 16-byte vectors.  The 3 input channels are padded to 4 (zero channel x zero weights:
// fma(0, 0, acc) = acc, so the chain is unchanged).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frames_normalize_nhwc4_kernel(const uint8_t *__restrict__ in, int64_t pixels,
                                                                     float *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pixels) return;
    float4 o;
    float v;
    v = (float)in[i * 3 + 0]; v = v / 255.0f; v = v - c_mean[0]; o.x = v / c_std[0];
    v = (float)in[i * 3 + 1]; v = v / 255.0f; v = v - c_mean[1]; o.y = v / c_std[1];
    v = (float)in[i * 3 + 2]; v = v / 255.0f; v = v - c_mean[2]; o.z = v / c_std[2];
    o.w = 0.0f;
    reinterpret_cast<float4 *>(out)[i] = o;
}

// w [Cout, Cin, 3, 3] -> wr [Cout, 9 * Cinp], k = (ky*3 + kx) * Cinp + ci, zero for ci >= Cin
__global__ __launch_bounds__(256) void conv_weight_repack_kernel(const float *__restrict__ w, int Cout, int Cin, int Cinp,
                                                                 float *__restrict__ wr)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K = 9 * Cinp;
    if (i >= (int64_t)Cout * K) return;
    const int co = (int)(i / K), k = (int)(i - (int64_t)co * K), tap = k / Cinp, ci = k - tap * Cinp;
    wr[i] = ci < Cin ? w[((int64_t)co * Cin + ci) * 9 + tap] : 0.0f;
}

// First convolution of the stack (3 input channels travelling padded to 4): K = 36 is a single K-tile of the MFMA kernel --
// all prologue and 1.9 GB of output per 150-frame video, 23 TF.  Direct form: thread = output pixel, its 3x3x3 neighbourhood
// in 27 registers, the weights wave-uniform (scalar loads), one 27-term chain per output channel in the GEMM's order
// (k = tap * 4 + ci ascending; the padded channel's products are +0 onto a chain that is never -0: skipped), + bias, ReLU.
// Weights for it: w [Cout, 3, 3, 3] -> wq [Cout / 4][27][4], the four output channels of a quad adjacent per k = tap * 3 + ci
// (one scalar 16-byte load feeds four chains).
__global__ __launch_bounds__(256) void conv_c4_weight_quad_kernel(const float *__restrict__ w, int Cout, float *__restrict__ wq)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cout * 27) return;
    const int j = i & 3, k = (i >> 2) % 27, q = i / 108, tap = k / 3, ci = k - 3 * tap, co = 4 * q + j;
    wq[i] = w[(co * 3 + ci) * 9 + tap];
}
// T16: Cout a multiple of 16 -- 16 channels per round, handed through a wave-private LDS tile so that a store instruction writes
// 16 pixels x 64 contiguous bytes (4 lanes per pixel) instead of 64 pixels x 16 bytes a 256-byte stride apart (the output is
// 1.9 GB per 150-frame video: partial-line writes were the whole cost of the first version).
// halo != 0: the output goes into the interior of a [B][H + 2][W + 2][Cout] tensor (the halo-padded stack: gemm.hip CHALO).
template <bool T16>
__global__ __launch_bounds__(256) void conv3x3_c4_direct_kernel(const float4 *__restrict__ x, int64_t pixels, int H, int W,
                                                                const float4 *__restrict__ wq, const float *__restrict__ b, int Cout,
                                                                float *__restrict__ y, int halo)
{
    auto opix = [&](int64_t q) -> int64_t {                              // output pixel index of input pixel q
        if (!halo) return q;
        const int64_t qn = q / ((int64_t)H * W);
        const int qr = (int)(q - qn * H * W), qy = qr / W, qx = qr - qy * W;
        return (qn * (H + 2) + qy + 1) * (W + 2) + qx + 1;
    };
    __shared__ __attribute__((aligned(16))) float tile_s[T16 ? 4 * 64 * 16 : 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pc = p < pixels ? p : pixels - 1;                     // (rows past the end: recomputed, never stored)
    const int hw = H * W;
    const int64_t n = pc / hw;
    const int rem = (int)(pc - n * hw), oy = rem / W, ox = rem - oy * W;
    float xin[27];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const float4 v = x[ok ? (n * H + iy) * W + ix : pc];
        xin[tap * 3 + 0] = ok ? v.x : 0.0f; xin[tap * 3 + 1] = ok ? v.y : 0.0f; xin[tap * 3 + 2] = ok ? v.z : 0.0f;
    }
    auto quad = [&](int co) -> float4 {
        float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const float4 *wk = wq + (co >> 2) * 27;
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const float4 wv = wk[k];
            a[0] = __builtin_fmaf(xin[k], wv.x, a[0]); a[1] = __builtin_fmaf(xin[k], wv.y, a[1]);
            a[2] = __builtin_fmaf(xin[k], wv.z, a[2]); a[3] = __builtin_fmaf(xin[k], wv.w, a[3]);
        }
        float4 o;
        o.x = a[0] + b[co]; o.y = a[1] + b[co + 1]; o.z = a[2] + b[co + 2]; o.w = a[3] + b[co + 3];
        o.x = o.x > 0.0f ? o.x : 0.0f; o.y = o.y > 0.0f ? o.y : 0.0f; o.z = o.z > 0.0f ? o.z : 0.0f; o.w = o.w > 0.0f ? o.w : 0.0f;
        return o;
    };
    if constexpr (T16) {
        float *tile = tile_s + wv * 64 * 16;
        const int64_t pw = (int64_t)blockIdx.x * 256 + wv * 64;         // the wave's first pixel
        int64_t op[4];                                                   // where this lane's four transposed stores of a round go
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int64_t q = pw + (lane >> 2) + 16 * j; op[j] = opix(q < pixels ? q : pixels - 1); }
        for (int co = 0; co < Cout; co += 16) {
            float4 o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = quad(co + 4 * j);
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<float4 *>(tile + lane * 16 + 4 * j) = o[j];
            __builtin_amdgcn_s_waitcnt(0xC07F);                          // lgkmcnt(0): the tile is the wave's own
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pl = (lane >> 2) + 16 * j;
                const float4 v = *reinterpret_cast<const float4 *>(tile + pl * 16 + 4 * (lane & 3));
                if (pw + pl < pixels) *reinterpret_cast<float4 *>(y + op[j] * Cout + co + 4 * (lane & 3)) = v;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        if (p >= pixels) return;
        float4 *yo = reinterpret_cast<float4 *>(y + opix(p) * Cout);
        for (int co = 0; co < Cout; co += 4) yo[co >> 2] = quad(co);
    }
}

__global__ __launch_bounds__(256) void maxpool2_nhwc_kernel(const float *__restrict__ x, int64_t B, int H, int W, int C,
                                                            float *__restrict__ y)
{
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Ho * Wo * C4) return;
    const int c4 = (int)(i % C4);
    int64_t r = i / C4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int64_t n = r / Ho;
    const float4 *p = reinterpret_cast<const float4 *>(x) + (((n * H + 2 * oy) * W + 2 * ox) * (int64_t)C4 + c4);
    const float4 a = p[0], b = p[C4], c = p[(int64_t)W * C4], d = p[(int64_t)W * C4 + C4];
    float4 o;
    o.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x));
    o.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
    o.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z));
    o.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
    reinterpret_cast<float4 *>(y)[i] = o;
}

// zero border of a halo-padded NHWC tensor [B][H + 2][W + 2][C] (C % 4 == 0): the 2 (W + 2) + 2 H border pixels of every image
__global__ __launch_bounds__(256) void halo_border_zero_kernel(float *__restrict__ x, int64_t B, int H, int W, int C)
{
    const int C4 = C / 4, nb = 2 * (W + 2) + 2 * H;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * nb * C4) return;
    const int c4 = (int)(i % C4);
    const int64_t r = i / C4;
    const int bi = (int)(r % nb);
    const int64_t n = r / nb;
    int py, px;
    if (bi < W + 2) { py = 0; px = bi; }
    else if (bi < 2 * (W + 2)) { py = H + 1; px = bi - (W + 2); }
    else { const int k = bi - 2 * (W + 2); py = 1 + (k >> 1); px = (k & 1) ? W + 1 : 0; }
    reinterpret_cast<float4 *>(x)[((n * (H + 2) + py) * (W + 2) + px) * C4 + c4] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// NHWC [B,H,W,C] -> [B, C*49] in torch's flatten order (channel-major), windows as AdaptiveAvgPool2d((7,7)); halo != 0: the input
// is the interior of a [B][H + 2][W + 2][C] tensor
__global__ __launch_bounds__(256) void adaptive_avgpool7_nhwc_kernel(const float *__restrict__ x, int64_t B, int H, int W,
                                                                     int C, float *__restrict__ y, int halo)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C * 49) return;
    const int ox = (int)(i % 7), oy = (int)((i / 7) % 7), c = (int)((i / 49) % C);
    const int64_t n = i / (49 * (int64_t)C);
    const int y0 = (oy * H) / 7, y1 = ((oy + 1) * H + 6) / 7, x0 = (ox * W) / 7, x1 = ((ox + 1) * W + 6) / 7;
    float acc = 0.0f;
    for (int iy = y0; iy < y1; ++iy)
        for (int ix = x0; ix < x1; ++ix)
            acc = acc + (halo ? x[((n * (H + 2) + iy + 1) * (W + 2) + ix + 1) * (int64_t)C + c] : x[((n * H + iy) * W + ix) * (int64_t)C + c]);
    y[i] = acc / (float)((y1 - y0) * (x1 - x0));
}

static int run_conv(const float *x, int B, int Cin, int H, int W, const float *w, const float *b, int Cout, float *y,
                    hipStream_t st)
{
    dim3 grid((unsigned)cdiv((int64_t)B * H * W, 256), (unsigned)cdiv(Cout, CO_T));
    hipLaunchKernelGGL(conv3x3_relu_kernel, grid, dim3(256), 0, st, x, B, Cin, H, W, w, b, Cout, y);
    VFR_CHECK_LAUNCH("conv3x3_relu_kernel");
    return VFR_OK;
}
static int run_maxpool(const float *x, int64_t planes, int H, int W, float *y, hipStream_t st)
{
    int64_t n = planes * (H / 2) * (W / 2);
    if (n == 0) return VFR_OK;
    hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, x, planes, H, W, y);
    VFR_CHECK_LAUNCH("maxpool2_kernel");
    return VFR_OK;
}
static int run_avgpool(const float *x, int64_t planes, int H, int W, float *y, hipStream_t st)
{
    hipLaunchKernelGGL(adaptive_avgpool7_kernel, dim3((unsigned)cdiv(planes * 49, 256)), dim3(256), 0, st, x, planes, H,
                       W, y);
    VFR_CHECK_LAUNCH("adaptive_avgpool7_kernel");
    return VFR_OK;
}

// Frames per pass through the conv stack.  The late layers are small GEMMs per frame (conv5: 196 rows x 512): at 32 frames
// they fill less than one round of the chip's 512 workgroup slots, at 150 (one DiDeMo video, get_rgb_features.py:47-61) a
// round is ~90 % full.  The two ping-pong activation buffers cost 2 x 12.8 MB per frame -- 3.9 GB at 150, nothing of 288 GB.
// T frames are split into equal chunks of at most VGG_FRAME_CHUNK.
constexpr int VGG_FRAME_CHUNK = 160;
static int vgg_chunk(int T)
{
    if (T <= VGG_FRAME_CHUNK) return T;
    const int n = (T + VGG_FRAME_CHUNK - 1) / VGG_FRAME_CHUNK;
    return (T + n - 1) / n;
}

struct VggPlan { size_t act_elems, wr_elems; int c_last, h_last, w_last; bool ok; };
static VggPlan plan_vgg(int chunk, int H, int W, const int *cfg, int ncfg)
{
    VggPlan p{0, 0, 4, H, W, true};                      // the 3 input channels travel padded to 4
    size_t cur = (size_t)chunk * 4 * H * W;
    p.act_elems = cur;
    int cin = 4;
    auto padded = [&](int c, int h, int w) { return (size_t)chunk * c * (h + 2) * (w + 2); };    // (a halo-padded stage needs this much)
    for (int i = 0; i < ncfg; ++i) {
        if (cfg[i] > 0) {
            if (cfg[i] % 4) p.ok = false;                 // NHWC float4 path
            p.wr_elems += align_up((size_t)cfg[i] * 9 * cin, 64);
            cin = p.c_last = cfg[i];
        } else {
            if (p.h_last < 2 || p.w_last < 2) p.ok = false;
            p.h_last /= 2; p.w_last /= 2;
        }
        cur = padded(p.c_last, p.h_last, p.w_last);
        if (cur > p.act_elems) p.act_elems = cur;
    }
    if (p.h_last < 1 || p.w_last < 1) p.ok = false;
    return p;
}

}  // namespace vfr

extern "C" {

int vfr_frames_normalize_f32(const uint8_t *frames_thwc, int T, int H, int W, float *out_tchw, vfr_stream_t stream)
{
    VFR_REQUIRE(frames_thwc && out_tchw && T >= 0 && H > 0 && W > 0, VFR_EINVAL, "vfr_frames_normalize_f32: bad argument");
    int64_t total = (int64_t)T * 3 * H * W;
    if (total == 0) return VFR_OK;
    hipLaunchKernelGGL(vfr::frames_normalize_kernel, dim3((unsigned)vfr::cdiv(total, 256)), dim3(256), 0,
                       vfr::as_stream(stream), frames_thwc, total, H, W, out_tchw);
    VFR_CHECK_LAUNCH("frames_normalize_kernel");
    return VFR_OK;
}

int vfr_conv3x3_relu_f32(const float *x, int B, int Cin, int H, int W, const float *w, const float *b, int Cout,
                         float *y, vfr_stream_t stream)
{
    VFR_REQUIRE(x && w && b && y && B >= 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, VFR_EINVAL,
                "vfr_conv3x3_relu_f32: bad argument");
    if (B == 0) return VFR_OK;
    return vfr::run_conv(x, B, Cin, H, W, w, b, Cout, y, vfr::as_stream(stream));
}

int vfr_maxpool2_f32(const float *x, int B, int C, int H, int W, float *y, vfr_stream_t stream)
{
    VFR_REQUIRE(x && y && B >= 0 && C > 0 && H > 1 && W > 1, VFR_EINVAL, "vfr_maxpool2_f32: bad argument");
    return vfr::run_maxpool(x, (int64_t)B * C, H, W, y, vfr::as_stream(stream));
}

int vfr_adaptive_avgpool7_f32(const float *x, int B, int C, int H, int W, float *y, vfr_stream_t stream)
{
    VFR_REQUIRE(x && y && B >= 0 && C > 0 && H > 0 && W > 0, VFR_EINVAL, "vfr_adaptive_avgpool7_f32: bad argument");
    if (B == 0) return VFR_OK;
    return vfr::run_avgpool(x, (int64_t)B * C, H, W, y, vfr::as_stream(stream));
}

size_t vfr_vgg_fc7_workspace_bytes(int T, int H, int W, const int *cfg_host, int ncfg, int fc_dim)
{
    if (T < 0 || H <= 0 || W <= 0 || !cfg_host || ncfg <= 0 || fc_dim <= 0) return 0;
    const int chunk = vfr::vgg_chunk(T);
    vfr::VggPlan p = vfr::plan_vgg(chunk, H, W, cfg_host, ncfg);
    size_t act = vfr::align_up(p.act_elems * sizeof(float), 256);
    size_t pooled = vfr::align_up((size_t)T * p.c_last * 49 * sizeof(float), 256);
    size_t h6 = vfr::align_up((size_t)T * fc_dim * sizeof(float), 256);
    size_t wr = vfr::align_up(p.wr_elems * sizeof(float), 256);
    return 2 * act + pooled + h6 + wr;
}

int vfr_vgg_fc7_f32(const uint8_t *frames_thwc, int T, int H, int W, const int *cfg_host, int ncfg,
                    const float *const *conv_w_host, const float *const *conv_b_host, const float *fc6_w,
                    const float *fc6_b, const float *fc7_w, const float *fc7_b, int fc_dim, float *out,
                    void *workspace, size_t workspace_bytes, vfr_stream_t stream)
{
    VFR_REQUIRE(frames_thwc && cfg_host && conv_w_host && conv_b_host && fc6_w && fc6_b && fc7_w && fc7_b && out &&
                    T >= 0 && H > 0 && W > 0 && ncfg > 0 && fc_dim > 0,
                VFR_EINVAL, "vfr_vgg_fc7_f32: bad argument");
    if (T == 0) return VFR_OK;
    const int chunk = vfr::vgg_chunk(T);
    vfr::VggPlan p = vfr::plan_vgg(chunk, H, W, cfg_host, ncfg);
    VFR_REQUIRE(p.ok, VFR_EUNSUPPORTED,
                "vfr_vgg_fc7_f32: needs conv widths that are multiples of 4 and %dx%d frames large enough for the pooling stages",
                H, W);
    VFR_REQUIRE(workspace && workspace_bytes >= vfr_vgg_fc7_workspace_bytes(T, H, W, cfg_host, ncfg, fc_dim),
                VFR_EWORKSPACE, "vfr_vgg_fc7_f32: workspace %zu < %zu bytes", workspace_bytes,
                vfr_vgg_fc7_workspace_bytes(T, H, W, cfg_host, ncfg, fc_dim));
    hipStream_t st = vfr::as_stream(stream);
    char *base = static_cast<char *>(workspace);
    const size_t act = vfr::align_up(p.act_elems * sizeof(float), 256);
    const size_t pooled_b = vfr::align_up((size_t)T * p.c_last * 49 * sizeof(float), 256);
    const size_t h6_b = vfr::align_up((size_t)T * fc_dim * sizeof(float), 256);
    float *bufA = reinterpret_cast<float *>(base), *bufB = reinterpret_cast<float *>(base + act);
    float *pooled = reinterpret_cast<float *>(base + 2 * act);
    float *h6 = reinterpret_cast<float *>(base + 2 * act + pooled_b);
    float *wr_base = reinterpret_cast<float *>(base + 2 * act + pooled_b + h6_b);
    const int K6 = p.c_last * 49;

    // the stack's first convolution (3 channels, K = 36) takes the direct kernel unless a pool is fused behind it
    const bool direct1 = vfr::opt_vgg_direct1() && cfg_host[0] > 0 && (cfg_host[0] % 4) == 0 &&
                         !(vfr::opt_vgg_fuse_pool() && ncfg > 1 && cfg_host[1] <= 0 && ((H | W) & 1) == 0);
    // halo-padded stack (gemm.hip CHALO: no tap masks, no selects in the convolution loader): every activation behind the first
    // convolution lives in the interior of a [B][h + 2][w + 2][C] tensor with a zero border.  Needs the direct first convolution,
    // every later convolution with C_in % 32 == 0, and every pool fusable into the convolution in front of it.
    bool halo = vfr::opt_vgg_halo() && direct1 && vfr::opt_vgg_fuse_pool();
    {
        int hh = H, ww = W, cc = cfg_host[0];
        bool prev_conv = true;
        for (int i = 1; i < ncfg && halo; ++i) {
            if (cfg_host[i] > 0) { if (cc % 32) halo = false; cc = cfg_host[i]; prev_conv = true; }
            else { if (!prev_conv || i == 1 || ((hh | ww) & 1)) halo = false; hh /= 2; ww /= 2; prev_conv = false; }
        }
    }
    auto zero_border = [&](float *buf, int bt, int h, int w, int c) {
        const int64_t n = (int64_t)bt * (2 * (w + 2) + 2 * h) * (c / 4);
        hipLaunchKernelGGL(vfr::halo_border_zero_kernel, dim3((unsigned)vfr::cdiv(n, 256)), dim3(256), 0, st, buf, (int64_t)bt, h, w, c);
    };
    // repack every conv weight once per call: [Cout,Cin,3,3] -> [Cout, 9*Cinp] tap-major (the chain order)
    {
        vfr::ProfScope prof(vfr::SITE_REPACK, st);
        float *wr = wr_base;
        int cin = 3, conv = 0;
        for (int i = 0; i < ncfg; ++i) {
            if (cfg_host[i] <= 0) continue;
            const int cinp = cin < 4 ? 4 : cin, cout = cfg_host[i];
            const int64_t n = (int64_t)cout * 9 * cinp;
            if (conv == 0 && direct1)             // (the direct kernel's quad layout: 27 * Cout floats in the same slot)
                hipLaunchKernelGGL(vfr::conv_c4_weight_quad_kernel, dim3((unsigned)vfr::cdiv((int64_t)cout * 27, 256)), dim3(256), 0, st,
                                   conv_w_host[conv], cout, wr);
            else
            hipLaunchKernelGGL(vfr::conv_weight_repack_kernel, dim3((unsigned)vfr::cdiv(n, 256)), dim3(256), 0, st,
                               conv_w_host[conv], cout, cin, cinp, wr);
            wr += vfr::align_up((size_t)n, 64);
            cin = cout;
            ++conv;
        }
    }
    VFR_CHECK_LAUNCH("conv_weight_repack_kernel");

    for (int t0 = 0; t0 < T; t0 += chunk) {
        const int bt = (T - t0) < chunk ? (T - t0) : chunk;
        {
            vfr::ProfScope prof(vfr::SITE_NORMALIZE, st);
            const int64_t pixels = (int64_t)bt * H * W;
            hipLaunchKernelGGL(vfr::frames_normalize_nhwc4_kernel, dim3((unsigned)vfr::cdiv(pixels, 256)), dim3(256), 0, st,
                               frames_thwc + (size_t)t0 * H * W * 3, pixels, bufA);
        }
        VFR_CHECK_LAUNCH("frames_normalize_nhwc4_kernel");
        float *cur = bufA, *nxt = bufB;
        const float *wr = wr_base;
        int c = 4, h = H, w = W, conv = 0;
        for (int i = 0; i < ncfg; ++i) {
            if (cfg_host[i] > 0) {
                const int cout = cfg_host[i];
                vfr::GemmArgs g{};
                g.A = cur; g.W = wr; g.ldw = 9 * c; g.out = nxt; g.ldo = cout; g.M = (int64_t)bt * h * w; g.N = cout;
                g.K = 9 * c; g.bias = conv_b_host[conv]; g.epi = vfr::EPI_BIAS | vfr::EPI_RELU; g.site = vfr::SITE_CONV;
                g.conv_h = h; g.conv_w = w; g.conv_cin = c;
                // a max-pool right behind this convolution rides in its epilogue (the window is a lane's accumulator quad):
                // the full-resolution activation is never written, the pool kernel never runs
                const bool pool = vfr::opt_vgg_fuse_pool() && i + 1 < ncfg && cfg_host[i + 1] <= 0 && ((h | w) & 1) == 0;
                if (pool) g.epi |= vfr::EPI_POOL2;
                if (halo) {                    // the destination's border for the geometry this layer writes
                    vfr::ProfScope prof(vfr::SITE_POOL2D, st);
                    zero_border(nxt, bt, pool ? h / 2 : h, pool ? w / 2 : w, cout);
                    g.conv_halo = conv > 0 ? 1 : 0;
                }
                if (conv == 0 && direct1) {
                    vfr::ProfScope prof(vfr::SITE_CONV, st);
                    if (cout % 16 == 0)
                        hipLaunchKernelGGL(vfr::conv3x3_c4_direct_kernel<true>, dim3((unsigned)vfr::cdiv(g.M, 256)), dim3(256), 0, st,
                                           reinterpret_cast<const float4 *>(cur), g.M, h, w, reinterpret_cast<const float4 *>(wr),
                                           conv_b_host[conv], cout, nxt, halo ? 1 : 0);
                    else
                        hipLaunchKernelGGL(vfr::conv3x3_c4_direct_kernel<false>, dim3((unsigned)vfr::cdiv(g.M, 256)), dim3(256), 0, st,
                                           reinterpret_cast<const float4 *>(cur), g.M, h, w, reinterpret_cast<const float4 *>(wr),
                                           conv_b_host[conv], cout, nxt, halo ? 1 : 0);
                } else
                if (int rc = vfr::gemm_nt(g, st)) return rc;
                wr += vfr::align_up((size_t)cout * 9 * c, 64);
                c = cout;
                ++conv;
                if (pool) { h /= 2; w /= 2; ++i; }
            } else {
                vfr::ProfScope prof(vfr::SITE_POOL2D, st);
                const int64_t n = (int64_t)bt * (h / 2) * (w / 2) * (c / 4);
                hipLaunchKernelGGL(vfr::maxpool2_nhwc_kernel, dim3((unsigned)vfr::cdiv(n, 256)), dim3(256), 0, st, cur,
                                   (int64_t)bt, h, w, c, nxt);
                h /= 2; w /= 2;
            }
            float *tmp = cur; cur = nxt; nxt = tmp;
        }
        VFR_CHECK_LAUNCH("vgg conv stack");
        {
            vfr::ProfScope prof(vfr::SITE_POOL2D, st);
            hipLaunchKernelGGL(vfr::adaptive_avgpool7_nhwc_kernel, dim3((unsigned)vfr::cdiv((int64_t)bt * c * 49, 256)),
                               dim3(256), 0, st, cur, (int64_t)bt, h, w, c, pooled + (size_t)t0 * K6, halo ? 1 : 0);
        }
        VFR_CHECK_LAUNCH("adaptive_avgpool7_nhwc_kernel");
    }
    if (int rc = vfr_linear_f32(pooled, T, K6, fc6_w, fc6_b, fc_dim, 1, h6, stream)) return rc;
    return vfr_linear_f32(h6, T, fc_dim, fc7_w, fc7_b, fc_dim, 1, out, stream);
}

}  // extern "C"
