// Error plumbing, options and the elementwise math probe of libvfr.so.
#include "vfr_common.h"
#include "vfr_math.cuh"

#include <cstring>
#include <mutex>
#include <vector>

namespace vfr {

static thread_local char g_err[512] = "";
static int g_opt_gemm = 1;
static int g_opt_profile = 0;
static int g_opt_score_fast = 1;
static int g_opt_score_split = 0;
static int g_opt_gemm_small = 0;       // small dense GEMMs: 0 = 32-row tiles, 64 = 64-row tiles (cross-check)
static int g_opt_lstm_xcd = 1;         // 1: XCD-aware workgroup order of the fused LSTM step; 0: launch order (cross-check)
static int g_opt_lstm_skip0 = 1;       // 1: the first LSTM step skips its recurrent segment (h_0 = 0); 0: runs it (cross-check)
static int g_opt_score_tasks = 0;      // > 0: wave-tasks the scorer's plan aims for (experiment; 0 = automatic)
static int g_opt_score_pre_b = 0;      // > 0: videos in ladder stage B (experiment; 0 = Nv/8 capped at 1024)
static int g_opt_gemm_pp = 0;          // 1: ping-pong schedule (512-thread workgroups, two tile groups) for the large MFMA GEMMs (experiment)
static int g_opt_lstm_tile = 0;        // 0: by grid size, 1: 64-row tiles, 2: 128-row tiles (fused LSTM step)

struct ProfPair { int site; hipEvent_t a, b; };
static std::vector<ProfPair> g_pairs;          // recorded, not yet read
static std::vector<hipEvent_t> g_free;         // recycled events
static hipEvent_t g_open[SITE_COUNT];
static double g_total_ms[SITE_COUNT];
static long long g_launches[SITE_COUNT];
static std::mutex g_prof_mu;

bool profiling() { return g_opt_profile != 0; }
static hipEvent_t take_event()
{
    if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void prof_begin(int site, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t e = take_event();
    (void)hipEventRecord(e, st);
    g_open[site] = e;
}
void prof_end(int site, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t e = take_event();
    (void)hipEventRecord(e, st);
    g_pairs.push_back({site, g_open[site], e});
}

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
int opt_score_fast() { return g_opt_score_fast; }
int opt_score_split() { return g_opt_score_split; }
int opt_score_pre_b() { return g_opt_score_pre_b; }
int opt_score_tasks() { return g_opt_score_tasks; }
int opt_lstm_skip0() { return g_opt_lstm_skip0; }
int opt_lstm_xcd() { return g_opt_lstm_xcd; }
int opt_gemm_small() { return g_opt_gemm_small; }
int opt_lstm_tile() { return g_opt_lstm_tile; }
int opt_gemm_pp() { return g_opt_gemm_pp; }

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
    if (name && !strcmp(name, "profile")) { vfr::g_opt_profile = value; return VFR_OK; }
    if (name && !strcmp(name, "score_fast")) { vfr::g_opt_score_fast = value; return VFR_OK; }
    if (name && !strcmp(name, "score_split")) { vfr::g_opt_score_split = value; return VFR_OK; }
    if (name && !strcmp(name, "score_pre_b")) { vfr::g_opt_score_pre_b = value; return VFR_OK; }
    if (name && !strcmp(name, "score_tasks")) { vfr::g_opt_score_tasks = value; return VFR_OK; }
    if (name && !strcmp(name, "lstm_skip0")) { vfr::g_opt_lstm_skip0 = value; return VFR_OK; }
    if (name && !strcmp(name, "lstm_xcd")) { vfr::g_opt_lstm_xcd = value; return VFR_OK; }
    if (name && !strcmp(name, "gemm_small")) { vfr::g_opt_gemm_small = value; return VFR_OK; }
    if (name && !strcmp(name, "lstm_tile")) { vfr::g_opt_lstm_tile = value; return VFR_OK; }
    if (name && !strcmp(name, "gemm_pp")) { vfr::g_opt_gemm_pp = value; return VFR_OK; }
    return vfr::fail(VFR_EINVAL, "vfr_set_option: unknown option '%s'", name ? name : "(null)");
}
int vfr_get_option(const char *name)
{
    if (name && !strcmp(name, "gemm")) return vfr::g_opt_gemm;
    if (name && !strcmp(name, "profile")) return vfr::g_opt_profile;
    if (name && !strcmp(name, "score_fast")) return vfr::g_opt_score_fast;
    if (name && !strcmp(name, "score_split")) return vfr::g_opt_score_split;
    if (name && !strcmp(name, "score_pre_b")) return vfr::g_opt_score_pre_b;
    if (name && !strcmp(name, "score_tasks")) return vfr::g_opt_score_tasks;
    if (name && !strcmp(name, "lstm_skip0")) return vfr::g_opt_lstm_skip0;
    if (name && !strcmp(name, "lstm_xcd")) return vfr::g_opt_lstm_xcd;
    if (name && !strcmp(name, "gemm_small")) return vfr::g_opt_gemm_small;
    if (name && !strcmp(name, "lstm_tile")) return vfr::g_opt_lstm_tile;
    if (name && !strcmp(name, "gemm_pp")) return vfr::g_opt_gemm_pp;
    return vfr::fail(VFR_EINVAL, "vfr_get_option: unknown option '%s'", name ? name : "(null)");
}

int vfr_profile_sites(void) { return vfr::SITE_COUNT; }

const char *vfr_profile_site_name(int site)
{
    static const char *names[vfr::SITE_COUNT] = {
        "none", "gemm_vis_seg", "gemm_vis_ctx", "vis_hidden", "gemm_vis_out", "embed", "gemm_lstm_in", "gemm_lstm_rec",
        "lstm_pointwise", "gemm_lang_fc", "score_fused", "topk_merge", "score_dense", "score_own", "pool", "linear",
        "conv3x3", "pool2d", "normalize", "score_rank", "score_prepass", "repack", "exchange"};
    return site >= 0 && site < vfr::SITE_COUNT ? names[site] : "?";
}

int vfr_profile_read(int site, double *total_ms, int64_t *launches, int reset)
{
    VFR_REQUIRE(site >= 0 && site < vfr::SITE_COUNT, VFR_EINVAL, "vfr_profile_read: bad site %d", site);
    std::lock_guard<std::mutex> lk(vfr::g_prof_mu);
    for (auto &p : vfr::g_pairs) {            // fold every completed pair (caller synchronised the device)
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { vfr::g_total_ms[p.site] += ms; vfr::g_launches[p.site] += 1; }
        vfr::g_free.push_back(p.a); vfr::g_free.push_back(p.b);
    }
    vfr::g_pairs.clear();
    if (total_ms) *total_ms = vfr::g_total_ms[site];
    if (launches) *launches = vfr::g_launches[site];
    if (reset) for (int i = 0; i < vfr::SITE_COUNT; ++i) { vfr::g_total_ms[i] = 0; vfr::g_launches[i] = 0; }
    return VFR_OK;
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
