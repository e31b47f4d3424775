// Error plumbing, options and the elementwise math probe of libvfr.so.
#include "vfr_common.h"
#include "vfr_math.h"

#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

namespace vfr {

static thread_local char g_err[512] = "";
static std::atomic<int> g_opt_gemm{1};
static std::atomic<int> g_opt_profile{0};
static std::atomic<int> g_opt_score_fast{1};
static std::atomic<int> g_opt_score_split{0};
static std::atomic<int> g_opt_gemm_small{0};       // small dense GEMMs: 0 = 32-row tiles, 64 = 64-row tiles (cross-check)
static std::atomic<int> g_opt_lstm_xcd{1};         // 1: XCD-aware workgroup order of the fused LSTM step; 0: launch order (cross-check)
static std::atomic<int> g_opt_lstm_skip0{1};       // 1: the first LSTM step skips its recurrent segment (h_0 = 0); 0: runs it (cross-check)
static std::atomic<int> g_opt_score_tasks{0};      // > 0: wave-tasks the scorer's plan aims for (experiment; 0 = automatic)
static std::atomic<int> g_opt_score_smallq{64};    // batches of up to this many queries (<= 64) are scored with lanes = clips / videos (vfr_score_topk_mfma, f32); 0: never
static std::atomic<int> g_opt_score_pre_b{0};      // > 0: videos in ladder stage B (experiment; 0 = Nv/16 capped at 640)
static std::atomic<int> g_opt_gemm_pp{0};          // 1: ping-pong schedule (512-thread workgroups, two tile groups) for the large MFMA GEMMs (experiment)
static std::atomic<int> g_opt_mfma_min{128};        // banks of fewer videos go to the exact scorer (the pre-filter's fixed launches cost more); tests set 0
static std::atomic<int> g_opt_lstm_small{2};       // batches of up to this many queries (<= 4) take the vector-chain LSTM step (measured: 17 / 25 / 38 us per step at 1 / 2 / 4 queries against 34 for the MFMA tiles); 0: always the tiles
static std::atomic<int> g_opt_lstm_persist{1};     // 1: one or two queries run the whole sequence in ONE launch with the weights resident in LDS (lstm_seq_small_kernel); 0: one launch per step
static std::atomic<int> g_opt_vgg_fuse_pool{1};    // 1: a 2x2 max-pool behind a VGG convolution runs in that convolution's epilogue (EPI_POOL2); 0: its own kernel (cross-check)
static std::atomic<int> g_opt_vgg_direct1{1};      // 1: the first VGG convolution (3 channels, K = 36) as the direct kernel; 0: the implicit-GEMM MFMA kernel (cross-check)
static std::atomic<int> g_opt_smallq_select{1};    // 1: the few-queries path takes its top-k by video selection (smallq_select_kernel); 0: key array + selection tree (cross-check)
static std::atomic<int> g_opt_lstm_persist_min{2}; // batches ABOVE this many queries (up to 32, the model's shape) take the single-launch MFMA sequence kernel; 32: never
static std::atomic<int> g_opt_lstm_persist_fault{-1}; // TEST HOOK: workgroup of the single-launch sequence kernels that withholds its h of step 1 (-1: none)
static std::atomic<int> g_opt_lstm_persist_max{32}; // ... and up to this many.  33 .. 64 are possible (two parts of <= 32 queries taking turns on the same resident weights): bit-identical, but 0.70 ms per 64-query pass against 0.62 for the tile steps (the two-part kernel's 275 weight registers + two parts' state spill): off by default
static std::atomic<int> g_opt_lstm_multi{0};       // EXPERIMENT (off): 1 = all T steps of the fused BiLSTM in ONE launch (gemm.hip: lstm_steps_mfma_kernel; 128-row tiles with lstm_tile 2) -- same bits; measured no faster than one launch per step at 5 000 queries, 4-5 % faster around 2 500 (HISTORY.md)
static std::atomic<int> g_opt_lstm_fast{1};        // 1: the table-start LSTM step without selects in its K-loop where the launch qualifies; 0: always the general form (cross-check)
static std::atomic<int> g_opt_smallq_rank{8};       // N: few-queries path, video-selection form, from N queries on: rank counts with lane = video (smallq_rank_kernel); 0: always the moment kernel (cross-check)
static std::atomic<int> g_opt_kth_seed{1};          // 1: stage A of the top-k threshold ladder takes its seed by bisection on the score bits (topk_kth_seed_kernel); 0: by the merge kernel (cross-check)
static std::atomic<int> g_opt_vgg_halo{1};         // 1: the VGG stack on halo-padded activations where its shape allows (select-free convolution loader); 0: unpadded (cross-check)
static std::atomic<int> g_opt_lstm_small4{1};      // 1: a single query of the model's shape takes the four-wave vector-chain step (weights streamed by three loader waves); 0: the one-wave step (cross-check)
static std::atomic<int> g_opt_score_defer{8};      // MFMA pre-filter, whole-video early-out: skip the rank half of the moment triangle when at most this many lanes of a wave are left undecided by dmin / dmax (they are re-counted exactly); -1: off (cross-check)
static std::atomic<int> g_opt_score_hist{1};       // MFMA pre-filter, top-k: 1 = the main launch's tasks tighten their threshold from a histogram of the candidates found so far; 0 = stage B's threshold throughout (cross-check)
static std::atomic<int> g_opt_score_sort{1};       // MFMA pre-filter with rank keys: 1 = the pass runs on the batch sorted by difficulty (so that the whole-video early-out, a wave decision, sees homogeneous waves); 0 = caller's order (cross-check)
static std::atomic<int> g_opt_lstm_tile{0};        // 0: by grid size, 1: 64-row tiles, 2: 128-row tiles (fused LSTM step)

struct Opt { const char *name; std::atomic<int> *v; };
static const Opt g_opts[] = {
    {"gemm", &g_opt_gemm}, {"profile", &g_opt_profile}, {"score_fast", &g_opt_score_fast}, {"score_split", &g_opt_score_split},
    {"score_pre_b", &g_opt_score_pre_b}, {"score_smallq", &g_opt_score_smallq}, {"score_tasks", &g_opt_score_tasks}, {"lstm_skip0", &g_opt_lstm_skip0},
    {"lstm_xcd", &g_opt_lstm_xcd}, {"gemm_small", &g_opt_gemm_small}, {"lstm_tile", &g_opt_lstm_tile}, {"lstm_small4", &g_opt_lstm_small4}, {"lstm_persist", &g_opt_lstm_persist}, {"lstm_persist_min", &g_opt_lstm_persist_min}, {"lstm_fast", &g_opt_lstm_fast}, {"lstm_multi", &g_opt_lstm_multi}, {"lstm_persist_max", &g_opt_lstm_persist_max}, {"lstm_persist_fault", &g_opt_lstm_persist_fault}, {"vgg_fuse_pool", &g_opt_vgg_fuse_pool}, {"vgg_direct1", &g_opt_vgg_direct1}, {"vgg_halo", &g_opt_vgg_halo}, {"score_kth_seed", &g_opt_kth_seed}, {"score_smallq_rank", &g_opt_smallq_rank}, {"score_smallq_select", &g_opt_smallq_select}, {"gemm_pp", &g_opt_gemm_pp}, {"score_mfma_min", &g_opt_mfma_min}, {"lstm_small", &g_opt_lstm_small}, {"score_defer", &g_opt_score_defer}, {"score_sort", &g_opt_score_sort}, {"score_hist", &g_opt_score_hist},
};

struct ProfPair { int site; hipEvent_t a, b; };
static std::vector<ProfPair> g_pairs;          // recorded, not yet read
static std::vector<hipEvent_t> g_free;         // recycled events
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
// The begin event travels in the caller's ProfScope (not in a per-site global), so any number of host threads can bracket
// launches of the same site at once; only the event pool and the pair list are shared, under the mutex.
hipEvent_t prof_begin(hipStream_t st)
{
    hipEvent_t e;
    { std::lock_guard<std::mutex> lk(g_prof_mu); e = take_event(); }
    if (e) (void)hipEventRecord(e, st);
    return e;
}
void prof_end(int site, hipEvent_t begin, hipStream_t st)
{
    hipEvent_t e;
    { std::lock_guard<std::mutex> lk(g_prof_mu); e = take_event(); }
    if (e) (void)hipEventRecord(e, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (begin && e) g_pairs.push_back({site, begin, e});
    else { if (begin) g_free.push_back(begin); if (e) g_free.push_back(e); }
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
int opt_score_smallq() { return g_opt_score_smallq; }
int opt_score_tasks() { return g_opt_score_tasks; }
int opt_lstm_skip0() { return g_opt_lstm_skip0; }
int opt_lstm_xcd() { return g_opt_lstm_xcd; }
int opt_gemm_small() { return g_opt_gemm_small; }
int opt_lstm_tile() { return g_opt_lstm_tile; }
int opt_lstm_small4() { return g_opt_lstm_small4; }
int opt_lstm_persist() { return g_opt_lstm_persist; }
int opt_lstm_persist_min() { return g_opt_lstm_persist_min; }
int opt_lstm_fast() { return g_opt_lstm_fast; }
int opt_lstm_multi() { return g_opt_lstm_multi; }
int opt_lstm_persist_max() { return g_opt_lstm_persist_max; }
int opt_lstm_persist_fault() { return g_opt_lstm_persist_fault; }
int opt_vgg_fuse_pool() { return g_opt_vgg_fuse_pool; }
int opt_vgg_direct1() { return g_opt_vgg_direct1; }
int opt_vgg_halo() { return g_opt_vgg_halo; }
int opt_score_kth_seed() { return g_opt_kth_seed; }
int opt_score_smallq_rank() { return g_opt_smallq_rank; }
int opt_score_smallq_select() { return g_opt_smallq_select; }
int device_cu_count()
{
    static std::atomic<int> cus[VFR_MAX_DEVICES];                     // 0: not asked yet (a device has at least one CU)
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const bool cached = dev >= 0 && dev < VFR_MAX_DEVICES;
    if (cached && (n = cus[dev].load()) > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    if (cached && n > 0) cus[dev].store(n);
    return n;
}
static std::atomic<unsigned *> g_fault_word{nullptr};
unsigned *fault_word() { return g_fault_word.load(); }
int opt_lstm_small() { return g_opt_lstm_small; }
int opt_gemm_pp() { return g_opt_gemm_pp; }
int opt_mfma_min() { return g_opt_mfma_min; }
int opt_score_defer() { return g_opt_score_defer; }
int opt_score_sort() { return g_opt_score_sort; }
int opt_score_hist() { return g_opt_score_hist; }

constexpr int FILL_MAX = 6;               // regions per launch
struct FillArgs { unsigned *dst[FILL_MAX]; const unsigned *src[FILL_MAX]; unsigned long long words[FILL_MAX]; unsigned value[FILL_MAX]; unsigned long long start[FILL_MAX + 1]; };
__global__ __launch_bounds__(256) void fill_regions_kernel(FillArgs a)
{
    // 16 bytes per thread where the region allows (every region base the library fills is 256-byte aligned; a misaligned or odd
    // one falls back to words)
    const unsigned long long total = a.start[FILL_MAX];
    for (unsigned long long i = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (unsigned long long)gridDim.x * 1024) {
        int r = 0;
#pragma unroll
        for (int q = 1; q < FILL_MAX; ++q) r = i >= a.start[q] ? q : r;      // (empty regions share their successor's start: the last match wins)
        const unsigned long long off = i - a.start[r];
        unsigned *d = a.dst[r] + off;
        const unsigned long long left = a.words[r] - off;
        const bool vec = left >= 4 && (((uintptr_t)d) & 15) == 0 && (!a.src[r] || (((uintptr_t)(a.src[r] + off)) & 15) == 0);
        if (vec) {
            const uint4 v = a.src[r] ? *reinterpret_cast<const uint4 *>(a.src[r] + off) : make_uint4(a.value[r], a.value[r], a.value[r], a.value[r]);
            *reinterpret_cast<uint4 *>(d) = v;
        } else {
            for (unsigned long long j = 0; j < (left < 4 ? left : 4); ++j) d[j] = a.src[r] ? a.src[r][off + j] : a.value[r];
        }
    }
}
int fill_regions(const FillJob *jobs, int n, hipStream_t st)
{
    FillArgs a{};
    unsigned long long total = 0;
    if (n > FILL_MAX) return fail(VFR_EINVAL, "fill_regions: %d regions > %d", n, FILL_MAX);
    for (int r = 0; r < FILL_MAX; ++r) {
        a.start[r] = total;
        if (r < n && jobs[r].bytes) {
            if ((jobs[r].bytes & 3) || (((uintptr_t)jobs[r].dst) & 3) || (((uintptr_t)jobs[r].src) & 3) || !jobs[r].dst)
                return fail(VFR_EINVAL, "fill_regions: region %d is not a whole number of aligned 4-byte words", r);
            a.dst[r] = static_cast<unsigned *>(jobs[r].dst); a.src[r] = static_cast<const unsigned *>(jobs[r].src);
            a.words[r] = jobs[r].bytes / 4;
            a.value[r] = jobs[r].value32;
            // regions start at multiples of 4 words in the flat index space, so a thread's 4 words never straddle two regions
            total += (a.words[r] + 3) / 4 * 4;
        }
    }
    a.start[FILL_MAX] = total;
    if (total == 0) return VFR_OK;
    unsigned long long blocks = (total / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_regions_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? VFR_OK : fail(VFR_EHIP, "fill_regions: launch failed");
}

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
    for (const auto &o : vfr::g_opts)
        if (name && !strcmp(name, o.name)) { o.v->store(value); return VFR_OK; }
    return vfr::fail(VFR_EINVAL, "vfr_set_option: unknown option '%s'", name ? name : "(null)");
}
int vfr_get_option(const char *name)
{
    for (const auto &o : vfr::g_opts)
        if (name && !strcmp(name, o.name)) return o.v->load();
    return vfr::fail(VFR_EINVAL, "vfr_get_option: unknown option '%s'", name ? name : "(null)");
}

int vfr_set_fault_word(uint32_t *word_host)
{
    vfr::g_fault_word.store(reinterpret_cast<unsigned *>(word_host));
    return VFR_OK;
}
int vfr_poll_faults(void)
{
    unsigned *w = vfr::g_fault_word.load();
    if (!w) return 0;
    const unsigned bits = __atomic_exchange_n(w, 0u, __ATOMIC_ACQ_REL);
    if (bits & VFR_FAULT_SEQ_RESCUED)
        (void)vfr::fail(0, "vfr_bilstm_final_f32: a single-launch sequence encoder gave up waiting for a workgroup that was not resident "
                           "(another stream / process on the device); the batch was re-encoded by lstm_seq_rescue_kernel -- results "
                           "are correct, the call was slow.  Do not run such calls concurrently, or set lstm_persist 0");
    return (int)bits;
}

int vfr_profile_sites(void) { return vfr::SITE_COUNT; }

const char *vfr_profile_site_name(int site)
{
    static const char *names[vfr::SITE_COUNT] = {
        "none", "gemm_vis_seg", "gemm_vis_ctx", "vis_hidden", "gemm_vis_out", "embed", "gemm_lstm_in", "gemm_lstm_rec",
        "lstm_pointwise", "gemm_lang_fc", "score_fused", "topk_merge", "score_dense", "score_own", "pool", "linear",
        "conv3x3", "pool2d", "normalize", "score_rank", "score_prepass", "repack", "exchange", "score_prep", "score_pairs", "score_finish", "score_fallback"};
    return site >= 0 && site < vfr::SITE_COUNT ? names[site] : "?";
}

int vfr_profile_read(int site, double *total_ms, int64_t *launches, int reset)
{
    VFR_REQUIRE(site >= 0 && site < vfr::SITE_COUNT, VFR_EINVAL, "vfr_profile_read: bad site %d", site);
    std::lock_guard<std::mutex> lk(vfr::g_prof_mu);
    // fold every completed pair; a pair whose events have not finished (the caller did not synchronise that stream) stays
    // queued for the next read -- its events are never recycled while they may still be pending
    std::vector<vfr::ProfPair> pending;
    int failed = 0;
    for (auto &p : vfr::g_pairs) {
        float ms = 0.0f;
        const hipError_t e = hipEventElapsedTime(&ms, p.a, p.b);
        if (e == hipSuccess) { vfr::g_total_ms[p.site] += ms; vfr::g_launches[p.site] += 1; }
        else if (e == hipErrorNotReady) { (void)hipGetLastError(); pending.push_back(p); continue; }
        else { (void)hipGetLastError(); ++failed; }
        vfr::g_free.push_back(p.a); vfr::g_free.push_back(p.b);
    }
    vfr::g_pairs.swap(pending);
    if (total_ms) *total_ms = vfr::g_total_ms[site];
    if (launches) *launches = vfr::g_launches[site];
    if (reset) for (int i = 0; i < vfr::SITE_COUNT; ++i) { vfr::g_total_ms[i] = 0; vfr::g_launches[i] = 0; }
    if (failed) return vfr::fail(VFR_EHIP, "vfr_profile_read: %d event pair(s) could not be timed and were dropped", failed);
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
