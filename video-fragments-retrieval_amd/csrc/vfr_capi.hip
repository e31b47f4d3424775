// Error plumbing, options and the elementwise math probe of libvfr.so.
#include "vfr_common.h"
#include "vfr_math.cuh"

#include <cstring>

namespace vfr {

static thread_local char g_err[512] = "";
static int g_opt_gemm = 1;

char *error_buffer() { return g_err; }
int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
int opt_gemm() { return g_opt_gemm; }

__global__ void math_probe_kernel(int op, const float *x, const float *y, float *out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i], b = y ? y[i] : 0.0f, r;
    switch (op) {
    case 0: r = c_expf(a); break;
    case 1: r = c_sigmoidf(a); break;
    case 2: r = c_tanhf(a); break;
    case 3: r = a / b; break;
    case 4: r = __builtin_sqrtf(a); break;
    case 5: r = __builtin_fmaf(a, b, a); break;
    default: r = 0.0f;
    }
    out[i] = r;
}

}  // namespace vfr

extern "C" {

int vfr_version(void) { return 100; }
const char *vfr_last_error(void) { return vfr::g_err; }

int vfr_set_option(const char *name, int value)
{
    if (name && !strcmp(name, "gemm")) { vfr::g_opt_gemm = value; return VFR_OK; }
    return vfr::fail(VFR_EINVAL, "vfr_set_option: unknown option '%s'", name ? name : "(null)");
}
int vfr_get_option(const char *name)
{
    if (name && !strcmp(name, "gemm")) return vfr::g_opt_gemm;
    return vfr::fail(VFR_EINVAL, "vfr_get_option: unknown option '%s'", name ? name : "(null)");
}

int vfr_math_f32(int op, const float *x, const float *y, float *out, int64_t n, vfr_stream_t stream)
{
    VFR_REQUIRE(x && out && n >= 0 && op >= 0 && op <= 5, VFR_EINVAL, "vfr_math_f32: bad argument");
    if (n == 0) return VFR_OK;
    hipLaunchKernelGGL(vfr::math_probe_kernel, dim3((unsigned)vfr::cdiv(n, 256)), dim3(256), 0, vfr::as_stream(stream),
                       op, x, y, out, n);
    VFR_CHECK_LAUNCH("vfr_math_f32");
    return VFR_OK;
}

}  // extern "C"
