// f2, second half: the encoders' backward (model/main.py:58-67 -- loss.backward() through CALModel.forward,
// model/models.py:54-66) as HIP kernels behind torch.autograd.Function (train.py).  The contractions are the chain GEMM of
// gemm.hip (vfr_linear_f32 on transposed operands); this file holds the pieces around them:
//   vfr_transpose_f32        out[c][r] = in[r][c]                    (dW = dY^T X needs both operands K-major)
//   vfr_colsum_f32           out[c] = sum_r in[r][c], r ascending    (bias gradients; fixed order, no float atomics)
//   vfr_relu_backward_f32    out = act > 0 ? grad : 0                (visual_fc's ReLU, model/models.py:23)
//   vfr_lstm_cell_forward_f32 / vfr_lstm_cell_backward_f32           (nn.LSTM's cell, gate order i, f, g, o; the forward
//                             keeps the activated gates and cell states of every step for the backward)
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

// one thread per column, rows in ascending order: a deterministic fp32 sum (the bias gradients are tiny reductions)
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ in, int64_t rows, int cols, float *__restrict__ out)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float acc = 0.0f;
    for (int64_t r = 0; r < rows; ++r) acc += in[r * cols + c];
    out[c] = acc;
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
    hipLaunchKernelGGL(vfr::colsum_kernel, dim3((unsigned)vfr::cdiv(cols, 256)), dim3(256), 0, vfr::as_stream(stream), in, rows, cols, out);
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

}  // extern "C"
