// Shared host-side helpers for libvfr.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/vfr.h"

namespace vfr {

char *error_buffer();   // thread-local message buffer (vfr_capi.hip)
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(vfr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// option switches (vfr_set_option)
int opt_gemm();
int opt_score_fast();
int opt_score_split();
int opt_score_pre_b();
int opt_score_smallq();
int opt_score_smallq_select();
int opt_score_tasks();
int opt_lstm_skip0();
int opt_lstm_xcd();
int opt_gemm_small();
int opt_lstm_tile();
int opt_lstm_small4();
int opt_lstm_persist();
int opt_lstm_fast();
int opt_lstm_multi();
int opt_lstm_persist_min();
int opt_lstm_persist_max();
int opt_lstm_persist_fault();
int opt_vgg_fuse_pool();
int opt_vgg_direct1();
int opt_vgg_halo();
int opt_score_smallq_rank();
int opt_score_kth_seed();
int device_cu_count();                  // of the CURRENT device (cached per device)
constexpr int VFR_MAX_DEVICES = 64;     // per-device caches (CU count, kernel attributes)
unsigned *fault_word();                 // vfr_set_fault_word: device-accessible host word kernels raise fault bits in (nullable)
int opt_lstm_small();
int opt_gemm_pp();
int opt_mfma_min();
int opt_score_defer();
int opt_score_sort();
int opt_score_hist();

// ---- launch-site profiler (vfr_set_option("profile", 1)): HIP events recorded on the launch stream
// around every instrumented launch; vfr_profile_read() turns them into per-site totals after a sync.
enum Site : int {
    SITE_NONE = 0,
    SITE_GEMM_VIS_SEG, SITE_GEMM_VIS_CTX, SITE_VIS_HIDDEN, SITE_GEMM_VIS_OUT,
    SITE_EMBED, SITE_GEMM_LSTM_IN, SITE_GEMM_LSTM_REC, SITE_LSTM_POINTWISE, SITE_GEMM_LANG_FC,
    SITE_SCORE_FUSED, SITE_TOPK_MERGE, SITE_SCORE_DENSE, SITE_SCORE_OWN, SITE_POOL, SITE_LINEAR,
    SITE_CONV, SITE_POOL2D, SITE_NORMALIZE, SITE_SCORE_RANK, SITE_SCORE_PREPASS, SITE_REPACK,
    SITE_EXCHANGE, SITE_SCORE_PREP, SITE_SCORE_PAIRS, SITE_SCORE_FINISH, SITE_SCORE_FALLBACK,
    SITE_COUNT
};
bool profiling();
hipEvent_t prof_begin(hipStream_t st);
void prof_end(int site, hipEvent_t begin, hipStream_t st);
struct ProfScope {                       // owns its begin event: re-entrant across host threads, sites and streams
    int site; hipStream_t st; bool on; hipEvent_t begin;
    ProfScope(int s, hipStream_t t) : site(s), st(t), on(s != SITE_NONE && profiling()), begin(nullptr) { if (on) begin = prof_begin(st); }
    ~ProfScope() { if (on) prof_end(site, begin, st); }
    ProfScope(const ProfScope &) = delete;
    ProfScope &operator=(const ProfScope &) = delete;
};

#define VFR_REQUIRE(cond, code, ...)                         \
    do {                                                     \
        if (!(cond)) return ::vfr::fail((code), __VA_ARGS__); \
    } while (0)

#define VFR_CHECK_LAUNCH(what)                                                                   \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) return ::vfr::fail(VFR_EHIP, "%s: %s", (what), hipGetErrorString(e_)); \
    } while (0)

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Buffer initialisation as ONE kernel launch for up to four regions (each a multiple of 4 bytes, 4-byte aligned): value32 is
// stored to every word; src != null copies instead.  hipMemsetAsync / hipMemcpyAsync cost ~20 us of stream time apiece on
// this runtime (a kernel launch ~5), and the entry points issue several per call.
struct FillJob { void *dst; const void *src; size_t bytes; unsigned value32; };
int fill_regions(const FillJob *jobs, int n, hipStream_t st);
inline int fill_region(void *dst, unsigned value32, size_t bytes, hipStream_t st) { const FillJob j{dst, nullptr, bytes, value32}; return fill_regions(&j, 1, st); }
inline int copy_region(void *dst, const void *src, size_t bytes, hipStream_t st) { const FillJob j{dst, src, bytes, 0u}; return fill_regions(&j, 1, st); }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- chain GEMM "NT": out[m][n] = epi( Cin[m][n] | 0  (+)  sum_k A[m][k] * W[n][k] ), one k-ascending
// fp32 fma chain per output (oracle rule R1).  Epilogue flags:
enum : int {
    EPI_NONE = 0,
    EPI_BIAS = 1,      // + bias[n] after the chain
    EPI_BIAS2 = 2,     // + (bias[n] + bias2[n])   (LSTM: b_ih + b_hh)
    EPI_RELU = 4,      // max(x, 0) last
    EPI_VIS = 8,       // clip-encoder hidden layer: x = ((chain + vis_cx[vis_row[m]][n]) + fma(te1, vis_w1[n], fma(te0, vis_w0[n], 0))) + bias[n]
    EPI_RES = 16,      // + res[m][n] after the bias, before the ReLU (a residual block's identity branch)
    EPI_POOL2 = 32,    // implicit-GEMM convolution only: 2x2 / stride-2 max-pool of the (bias, ReLU) output fused into the epilogue.
                       // GEMM row m = ((n*(h/2) + y/2)*(w/2) + x/2)*4 + (y&1)*2 + (x&1): the four rows a lane's accumulator
                       // register quad holds are one pooling window, so the max needs no exchange; out is the POOLED NHWC
                       // tensor [M/4, N].  Needs even conv_h / conv_w.
};
struct GemmArgs {
    const float *A; int64_t lda;
    const float *W; int64_t ldw;
    const float *Cin; int64_t ldc;     // nullable
    const float *bias, *bias2;         // nullable
    const float *res; int64_t ldr;     // EPI_RES
    float *out; int64_t ldo;
    int64_t M; int N; int K;
    int epi;
    int site;                          // profiler site (SITE_NONE = not instrumented)
    // xcd_cols > 0 (dense MFMA kernel, set by gemm_nt): 1-D grid; workgroup L computes row tile (L / (8*xcd_cols))*8 + L % 8
    // and column tile (L / 8) % xcd_cols, so the xcd_cols column tiles that share an A row tile are dispatched 8 apart --
    // onto the SAME XCD (workgroups go round-robin over the 8 XCDs) -- and A streams from HBM once, not once per column tile
    int xcd_cols;
    // EPI_VIS (MFMA kernel only): vis_row[m] = video of clip row m, vis_te[m] = (t/n, (t+1)/n), vis_cx [videos, N] = the
    // per-video context chains, vis_w0 / vis_w1 [N] = the two temporal-endpoint weight columns
    const int *vis_row; const float *vis_te; const float *vis_cx; const float *vis_w0, *vis_w1;
    // implicit-GEMM 3x3 convolution (conv_cin > 0): A is an NHWC activation [B, conv_h, conv_w, conv_cin]; row m is the
    // output pixel (n, oy, ox), column k = (ky*3 + kx)*conv_cin + ci reads x[n][oy+ky-1][ox+kx-1][ci] (0 outside).
    // M = B*conv_h*conv_w, K = 9*conv_cin (conv_cin % 4 == 0), out [M, N] is the NHWC output.
    int conv_h, conv_w, conv_cin;
    // conv_halo != 0 (conv_cin % 32 == 0): A is [B][conv_h + 2][conv_w + 2][conv_cin] with a zero border and `out` is written into the
    // interior of a tensor padded the same way ([B][oh + 2][ow + 2][N]; oh x ow = the pooled size under EPI_POOL2)
    int conv_halo;
    // fused LSTM step (lstm_H > 0): K is segmented, A row = [A (K cols) | A2 (K2 cols)], W row = [W | W2]; the 128 tile
    // columns are 32 hidden units x 4 gates (permuted so one lane pair holds i,f,g,o of a unit); the epilogue adds
    // bias + bias2, applies the gate nonlinearities and writes c (in place) and h_out -- no gates array.
    // N (grid) = lstm_H hidden units, bias / bias2 are the [4H] b_ih / b_hh.
    const float *A2; int64_t lda2; const float *W2; int64_t ldw2; int K2;
    float *lstm_c; float *lstm_h; int64_t lstm_ldh; int lstm_H;
    const float *lstm_cin;             // previous cell state (c is ping-ponged like h: a joining row reads row 0's OLD state)
    // row bookkeeping of the fused step (all nullable): lstm_xrow[m] = query whose tokens feed GEMM row m;
    // lstm_mcount[s] = number of active rows at step s (rows are sorted so the active set is a prefix); rows that join
    // at step s (>= lstm_mcount[s-1], or >= 1 at s = 0) take their incoming state from row 0, the all-pad row.
    const int *lstm_xrow; const int *lstm_mcount; int lstm_step;
    // input-projection table (lstm_tok != null, K == 0): the x-part of every gate chain depends only on the token, so it
    // is computed once per VOCABULARY entry (Cin = P [vocab, ldc], columns in the permuted tile order) and GEMM row m's
    // accumulators start from P[lstm_tok[m]] (the clamped token of row m's query at this step) -- the same chain,
    // continued over h.
    const int *lstm_tok;
    // training forward (nullable): the activated gates i, f, g, o of every row, [M, 4H] -- what the backward needs
    float *lstm_gates;
};
int gemm_nt(const GemmArgs &g, hipStream_t st);
// two GEMMs of identical shape as ONE grid (blockIdx.z picks the problem): fills the chip when one alone leaves a
// partial last round (the forward and reverse LSTM directions)
int gemm_nt_pair(const GemmArgs &g0, const GemmArgs &g1, hipStream_t st);
// fused LSTM step for both directions (see GemmArgs::lstm_H)
int lstm_step_pair(const GemmArgs &g0, const GemmArgs &g1, hipStream_t st);
// all T fused steps of both directions in ONE launch (gemm.hip: lstm_steps_mfma_kernel).  g0 / g1 = the step-invariant
// arguments (W2 / ldw2 / lda2 / biases / Cin + ldc = the projection table / M = rows / lstm_H / lstm_ldh / lstm_mcount for the
// reverse direction); hbuf / cbuf = three rotating state buffers (buffer 0 zeroed; the final state is in buffer T % 3);
// sync = lstm_steps_sync_words() zeroed words (word 8 = the error word: non-zero when the launch gave up).
bool lstm_steps_supported(int64_t R, int H);
size_t lstm_steps_sync_words(int64_t R, int T);
int lstm_steps_run(const GemmArgs &g0, const GemmArgs &g1, float *const hbuf[3], float *const cbuf[3], const int *tokidx, const int *mcount,
                   unsigned *sync, int T, hipStream_t st);
// two GEMMs of identical shape, each cut into nsplit equal K ranges, as ONE grid of 32-row tiles: range j of problem d writes
// its partial chains to out_d + j * out_stride (the consumer adds the partials in the order 0..nsplit-1).  For the skinny
// products of backpropagation through time (dh = dpre W_hh: [B,4H] x [4H,H], B a few hundred) whose own grid is 64 workgroups.
int gemm_nt_splitk_pair(const GemmArgs &g0, const GemmArgs &g1, int nsplit, int64_t out_stride, hipStream_t st);
// W_ih [4H, E] -> [128*ceil(H/32), E] in the fused step's tile-column order (for the vocabulary input-projection table)
int lstm_permute_rows(const float *W, int H, int E, float *out, hipStream_t st);
// copy a [rows, cols] block out of a wider row-major matrix into a dense, 16-byte aligned buffer
int repack_rows(const float *src, int64_t ld_src, int rows, int cols, float *dst, hipStream_t st);

}  // namespace vfr
