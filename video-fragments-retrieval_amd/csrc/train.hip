// f2, second half: the encoders' backward (model/main.py:58-67 -- loss.backward() through CALModel.forward,
// model/models.py:54-66) as HIP kernels behind torch.autograd.Function (train.py).  The contractions are the chain GEMM of
// gemm.hip (vfr_linear_f32 on transposed operands); this file holds the pieces around them:
//   vfr_transpose_f32        out[c][r] = in[r][c]                    (dW = dY^T X needs both operands K-major)
//   vfr_colsum_f32           out[c] = sum_r in[r][c], fixed order    (bias gradients; deterministic, no float atomics)
//   vfr_relu_backward_f32    out = act > 0 ? grad : 0                (visual_fc's ReLU, model/models.py:23)
//   vfr_lstm_cell_forward_f32 / vfr_lstm_cell_backward_f32           (nn.LSTM's cell, gate order i, f, g, o; the forward
//                             keeps the activated gates and cell states of every step for the backward)
//   vfr_bilstm_train_forward_f32 / vfr_bilstm_train_backward_f32     (the whole recurrence of both directions: T launches of
//                             the fused step of the inference path, gates stored as well; T x (cell backward of both
//                             directions, dh = dpre W_hh of both directions as one split-K grid) for the way back)
#include "vfr_common.h"
#include "vfr_math.h"

namespace vfr {

__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, int64_t rows, int64_t cols,
                                                        float *__restrict__ out)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8 threads, 32 x 32 tile
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int64_t r = r0 + ty + i, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + i][tx] = in[r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int64_t c = c0 + ty + i, r = r0 + tx;
        if (r < rows && c < cols) out[c * rows + r] = tile[tx][ty + i];
    }
}

// column sums in a fixed order (deterministic, no float atomics): a block owns 64 columns; its 16 row phases each add every
// 16th row (coalesced 256-byte row segments), then the phases are added 0..15.  One launch, rows x cols / 1024 loads per thread.
__global__ __launch_bounds__(1024) void colsum_kernel(const float *__restrict__ in, int64_t rows, int cols, float *__restrict__ out)
{
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.0f;
    if (c < cols)
        for (int64_t r = ph; r < rows; r += 16) acc += in[r * cols + c];
    red[ph][cl] = acc;
    __syncthreads();
    if (ph == 0 && c < cols) {
        float t = red[0][cl];
#pragma unroll
        for (int i = 1; i < 16; ++i) t += red[i][cl];
        out[c] = t;
    }
}

__global__ __launch_bounds__(256) void relu_backward_kernel(const float *__restrict__ grad, const float *__restrict__ act,
                                                            int64_t n, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = act[i] > 0.0f ? grad[i] : 0.0f;
}

// gates = act(pre + xproj): pre [B,4H] = h_prev W_hh^T + b_hh, xproj row b at xproj + b * x_stride = x_t W_ih^T + b_ih.
// Stores the activated gates [B,4H] (i, f, g, o), c [B,H], h [B,H].
__global__ __launch_bounds__(256) void lstm_cell_forward_kernel(const float *__restrict__ pre, const float *__restrict__ xproj,
                                                                int64_t x_stride, const float *__restrict__ c_prev, int64_t B,
                                                                int H, float *__restrict__ gates, float *__restrict__ c,
                                                                float *__restrict__ h)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int64_t b = i / H;
    const int j = (int)(i - b * H);
    const float *p = pre + b * 4 * H, *x = xproj + b * x_stride;
    const float ig = c_sigmoidf(p[j] + x[j]);
    const float fg = c_sigmoidf(p[H + j] + x[H + j]);
    const float gg = c_tanhf(p[2 * H + j] + x[2 * H + j]);
    const float og = c_sigmoidf(p[3 * H + j] + x[3 * H + j]);
    const float cn = __builtin_fmaf(fg, c_prev[i], ig * gg);
    float *g4 = gates + b * 4 * H;
    g4[j] = ig; g4[H + j] = fg; g4[2 * H + j] = gg; g4[3 * H + j] = og;
    c[i] = cn;
    h[i] = og * c_tanhf(cn);
}

// one step of backpropagation through time: (dh, dc) at the step's output -> dpre [B,4H] (gradient of the gate
// pre-activations) and dc at the step's input (in place)
__global__ __launch_bounds__(256) void lstm_cell_backward_kernel(const float *__restrict__ dh, float *__restrict__ dc,
                                                                 const float *__restrict__ gates, const float *__restrict__ c_prev,
                                                                 const float *__restrict__ c_cur, int64_t B, int H,
                                                                 float *__restrict__ dpre)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int64_t b = i / H;
    const int j = (int)(i - b * H);
    const float *g4 = gates + b * 4 * H;
    const float ig = g4[j], fg = g4[H + j], gg = g4[2 * H + j], og = g4[3 * H + j];
    const float tc = c_tanhf(c_cur[i]);
    const float dhi = dh[i];
    const float dct = dc[i] + dhi * og * (1.0f - tc * tc);
    float *d4 = dpre + b * 4 * H;
    d4[j] = dct * gg * ig * (1.0f - ig);
    d4[H + j] = dct * c_prev[i] * fg * (1.0f - fg);
    d4[2 * H + j] = dct * ig * (1.0f - gg * gg);
    d4[3 * H + j] = dhi * tc * og * (1.0f - og);
    dc[i] = dct * fg;
}

// both directions' cell backward in one launch; the incoming dh is the sum of nparts partial products (the split-K ranges
// of the previous launch's dh = dpre W_hh, added in the order 0..nparts-1) or, at the last time step, the gradient of h_n
__global__ __launch_bounds__(256) void lstm_cell_backward_pair_kernel(const float *__restrict__ dh, int64_t dh_dir, int64_t dh_part,
                                                                      int64_t dh_ld, int nparts, float *__restrict__ dc,
                                                                      const float *__restrict__ gates, int64_t gates_dir,
                                                                      const float *__restrict__ c_prev, int64_t c_dir, int64_t B,
                                                                      int H, float *__restrict__ dpre)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * B * H) return;
    const int d = (int)(i / (B * H));
    const int64_t r = i - d * B * H, b = r / H;
    const int j = (int)(r - b * H);
    const float *p = dh + d * dh_dir + b * dh_ld + j;
    float dhi = p[0];
    for (int k = 1; k < nparts; ++k) dhi += p[k * dh_part];
    const float *g4 = gates + d * gates_dir + b * 4 * H;
    const float ig = g4[j], fg = g4[H + j], gg = g4[2 * H + j], og = g4[3 * H + j];
    const float cp = c_prev[d * c_dir + r], cc = c_prev[d * c_dir + B * H + r];      // states before / after this step
    const float tc = c_tanhf(cc);
    const float dct = dc[i] + dhi * og * (1.0f - tc * tc);
    float *d4 = dpre + d * gates_dir + b * 4 * H;
    d4[j] = dct * gg * ig * (1.0f - ig);
    d4[H + j] = dct * cp * fg * (1.0f - fg);
    d4[2 * H + j] = dct * ig * (1.0f - gg * gg);
    d4[3 * H + j] = dhi * tc * og * (1.0f - og);
    dc[i] = dct * fg;
}

static int bptt_splits(int64_t B, int H)
{
    // enough K ranges for two workgroups per CU, as long as a range stays a whole number of float4 (and at least 128 long)
    const int64_t tiles = 2 * cdiv(B, 32) * cdiv(H, 128);
    int s = 1;
    while (s < 8 && tiles * s < 512 && (4 * H) % (8 * s) == 0 && (4 * H) / (2 * s) >= 128) s *= 2;
    return s;
}

}  // namespace vfr

extern "C" {

int vfr_transpose_f32(const float *in, int64_t rows, int64_t cols, float *out, vfr_stream_t stream)
{
    VFR_REQUIRE(rows >= 0 && cols >= 0 && (rows * cols == 0 || (in && out)), VFR_EINVAL, "vfr_transpose_f32: bad argument");
    if (rows == 0 || cols == 0) return VFR_OK;
    dim3 grid((unsigned)vfr::cdiv(cols, 32), (unsigned)vfr::cdiv(rows, 32));
    VFR_REQUIRE(grid.y <= 65535u, VFR_EUNSUPPORTED, "vfr_transpose_f32: more than 2M rows");
    hipLaunchKernelGGL(vfr::transpose_kernel, grid, dim3(256), 0, vfr::as_stream(stream), in, rows, cols, out);
    VFR_CHECK_LAUNCH("transpose_kernel");
    return VFR_OK;
}

int vfr_colsum_f32(const float *in, int64_t rows, int cols, float *out, vfr_stream_t stream)
{
    VFR_REQUIRE(rows >= 0 && cols > 0 && out && (rows == 0 || in), VFR_EINVAL, "vfr_colsum_f32: bad argument");
    hipLaunchKernelGGL(vfr::colsum_kernel, dim3((unsigned)vfr::cdiv(cols, 64)), dim3(1024), 0, vfr::as_stream(stream), in, rows, cols, out);
    VFR_CHECK_LAUNCH("colsum_kernel");
    return VFR_OK;
}

int vfr_relu_backward_f32(const float *grad, const float *act, int64_t n, float *out, vfr_stream_t stream)
{
    VFR_REQUIRE(n >= 0 && (n == 0 || (grad && act && out)), VFR_EINVAL, "vfr_relu_backward_f32: bad argument");
    if (n == 0) return VFR_OK;
    hipLaunchKernelGGL(vfr::relu_backward_kernel, dim3((unsigned)vfr::cdiv(n, 256)), dim3(256), 0, vfr::as_stream(stream), grad, act, n, out);
    VFR_CHECK_LAUNCH("relu_backward_kernel");
    return VFR_OK;
}

int vfr_lstm_cell_forward_f32(const float *pre, const float *xproj, int64_t x_stride, const float *c_prev, int64_t B, int H,
                              float *gates, float *c, float *h, vfr_stream_t stream)
{
    VFR_REQUIRE(B >= 0 && H > 0 && x_stride >= 4 * (int64_t)H && (B == 0 || (pre && xproj && c_prev && gates && c && h)), VFR_EINVAL,
                "vfr_lstm_cell_forward_f32: bad argument");
    if (B == 0) return VFR_OK;
    hipLaunchKernelGGL(vfr::lstm_cell_forward_kernel, dim3((unsigned)vfr::cdiv(B * H, 256)), dim3(256), 0, vfr::as_stream(stream), pre,
                       xproj, x_stride, c_prev, B, H, gates, c, h);
    VFR_CHECK_LAUNCH("lstm_cell_forward_kernel");
    return VFR_OK;
}

int vfr_lstm_cell_backward_f32(const float *dh, float *dc, const float *gates, const float *c_prev, const float *c_cur, int64_t B,
                               int H, float *dpre, vfr_stream_t stream)
{
    VFR_REQUIRE(B >= 0 && H > 0 && (B == 0 || (dh && dc && gates && c_prev && c_cur && dpre)), VFR_EINVAL,
                "vfr_lstm_cell_backward_f32: bad argument");
    if (B == 0) return VFR_OK;
    hipLaunchKernelGGL(vfr::lstm_cell_backward_kernel, dim3((unsigned)vfr::cdiv(B * H, 256)), dim3(256), 0, vfr::as_stream(stream), dh, dc,
                       gates, c_prev, c_cur, B, H, dpre);
    VFR_CHECK_LAUNCH("lstm_cell_backward_kernel");
    return VFR_OK;
}

int vfr_bilstm_train_forward_f32(const float *X, int64_t B, int T, int E, int H, const float *Wih_f, const float *Whh_f,
                                 const float *bih_f, const float *bhh_f, const float *Wih_b, const float *Whh_b,
                                 const float *bih_b, const float *bhh_b, float *gates, float *cs, float *hs, vfr_stream_t stream)
{
    VFR_REQUIRE(B >= 0 && T > 0 && E > 0 && H > 0 && Wih_f && Whh_f && bih_f && bhh_f && Wih_b && Whh_b && bih_b && bhh_b &&
                    (B == 0 || (X && gates && cs && hs)), VFR_EINVAL, "vfr_bilstm_train_forward_f32: bad argument");
    if (B == 0) return VFR_OK;
    VFR_REQUIRE((E % 4) == 0 && (H % 4) == 0, VFR_EUNSUPPORTED, "vfr_bilstm_train_forward_f32: E and H must be multiples of 4");
    hipStream_t st = vfr::as_stream(stream);
    const float *Wih[2] = {Wih_f, Wih_b}, *Whh[2] = {Whh_f, Whh_b}, *bih[2] = {bih_f, bih_b}, *bhh[2] = {bhh_f, bhh_b};
    const size_t BH = (size_t)B * H;
    for (int d = 0; d < 2; ++d)              // zero initial state (model/models.py:50-52) = slot 0 of both directions
        if (hipMemsetAsync(cs + (size_t)d * (T + 1) * BH, 0, BH * sizeof(float), st) != hipSuccess ||
            hipMemsetAsync(hs + (size_t)d * (T + 1) * BH, 0, BH * sizeof(float), st) != hipSuccess)
            return vfr::fail(VFR_EHIP, "vfr_bilstm_train_forward_f32: buffer initialisation failed");
    for (int s = 0; s < T; ++s) {
        vfr::GemmArgs g[2]{};
        for (int d = 0; d < 2; ++d) {
            const int t = d ? T - 1 - s : s;
            float *c0 = cs + ((size_t)d * (T + 1) + s) * BH, *h0 = hs + ((size_t)d * (T + 1) + s) * BH;
            g[d].A = X + (size_t)t * E; g[d].lda = (int64_t)T * E; g[d].W = Wih[d]; g[d].ldw = E; g[d].K = E;
            g[d].A2 = h0; g[d].lda2 = H; g[d].W2 = Whh[d]; g[d].ldw2 = H;
            g[d].K2 = (s == 0 && vfr::opt_lstm_skip0()) ? 0 : H;          // h_0 = 0: the recurrent terms leave the chain as it is
            g[d].bias = bih[d]; g[d].bias2 = bhh[d]; g[d].M = B; g[d].N = 4 * H;
            g[d].lstm_cin = c0; g[d].lstm_c = c0 + BH; g[d].lstm_h = h0 + BH; g[d].lstm_ldh = H; g[d].lstm_H = H;
            g[d].out = h0 + BH; g[d].lstm_step = s; g[d].site = vfr::SITE_GEMM_LSTM_REC;
            g[d].lstm_gates = gates + ((size_t)d * T + s) * (size_t)B * 4 * H;
        }
        if (int rc = vfr::lstm_step_pair(g[0], g[1], st)) return rc;
    }
    return VFR_OK;
}

size_t vfr_bilstm_train_backward_workspace_bytes(int64_t B, int H)
{
    if (B <= 0 || H <= 0) return 0;
    return ((size_t)2 * vfr::bptt_splits(B, H) + 2) * (size_t)B * H * sizeof(float);
}

int vfr_bilstm_train_backward_f32(const float *gout, const float *gates, const float *cs, const float *WhhT_f, const float *WhhT_b,
                                  int64_t B, int T, int H, float *dpre, void *workspace, size_t workspace_bytes, vfr_stream_t stream)
{
    VFR_REQUIRE(B >= 0 && T > 0 && H > 0 && WhhT_f && WhhT_b && (B == 0 || (gout && gates && cs && dpre)), VFR_EINVAL,
                "vfr_bilstm_train_backward_f32: bad argument");
    if (B == 0) return VFR_OK;
    VFR_REQUIRE((H % 4) == 0, VFR_EUNSUPPORTED, "vfr_bilstm_train_backward_f32: H must be a multiple of 4");
    VFR_REQUIRE(workspace && workspace_bytes >= vfr_bilstm_train_backward_workspace_bytes(B, H), VFR_EWORKSPACE,
                "vfr_bilstm_train_backward_f32: workspace %zu < %zu bytes", workspace_bytes,
                vfr_bilstm_train_backward_workspace_bytes(B, H));
    hipStream_t st = vfr::as_stream(stream);
    const int S = vfr::bptt_splits(B, H);
    const size_t BH = (size_t)B * H;
    float *parts = (float *)workspace, *dc = parts + (size_t)2 * S * BH;       // dh partials [2][S][B][H], dc [2][B][H]
    if (hipMemsetAsync(dc, 0, 2 * BH * sizeof(float), st) != hipSuccess)
        return vfr::fail(VFR_EHIP, "vfr_bilstm_train_backward_f32: buffer initialisation failed");
    const int64_t gdir = (int64_t)T * B * 4 * H, cdir = (int64_t)(T + 1) * BH;
    const unsigned blocks = (unsigned)vfr::cdiv(2 * (int64_t)BH, 256);
    for (int s = T - 1; s >= 0; --s) {
        {
        vfr::ProfScope prof(vfr::SITE_LSTM_POINTWISE, st);
        if (s == T - 1)                      // the gradient of h_n = [forward final | reverse final]
            hipLaunchKernelGGL(vfr::lstm_cell_backward_pair_kernel, dim3(blocks), dim3(256), 0, st, gout, (int64_t)H, (int64_t)0,
                               (int64_t)2 * H, 1, dc, gates + (size_t)s * B * 4 * H, gdir, cs + (size_t)s * BH, cdir, B, H,
                               dpre + (size_t)s * B * 4 * H);
        else
            hipLaunchKernelGGL(vfr::lstm_cell_backward_pair_kernel, dim3(blocks), dim3(256), 0, st, parts, (int64_t)(S * BH),
                               (int64_t)BH, (int64_t)H, S, dc, gates + (size_t)s * B * 4 * H, gdir, cs + (size_t)s * BH, cdir, B, H,
                               dpre + (size_t)s * B * 4 * H);
        }
        VFR_CHECK_LAUNCH("lstm_cell_backward_pair_kernel");
        if (s == 0) break;
        vfr::GemmArgs g[2]{};
        for (int d = 0; d < 2; ++d) {        // gradient reaching h of step s - 1: dpre [B,4H] x W_hh [4H,H]
            g[d].A = dpre + (size_t)d * gdir + (size_t)s * B * 4 * H; g[d].lda = 4 * H;
            g[d].W = d ? WhhT_b : WhhT_f; g[d].ldw = 4 * H;
            g[d].out = parts + (size_t)d * S * BH; g[d].ldo = H; g[d].M = B; g[d].N = H; g[d].K = 4 * H;
            g[d].site = vfr::SITE_GEMM_LSTM_REC;
        }
        if (int rc = vfr::gemm_nt_splitk_pair(g[0], g[1], S, (int64_t)BH, st)) return rc;
    }
    return VFR_OK;
}

}  // extern "C"
