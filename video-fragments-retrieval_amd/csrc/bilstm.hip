// a8: query encoder = CALModel GloVe branch (model/models.py:61-66): Embedding gather
// [-> unit-norm x learnable length, :62-64] -> 1-layer BiLSTM(H), zero initial state (:50-52), every
// one of the T steps processed (pads included, Q6) -> [h_fwd | h_bwd] -> Linear(2H, D).
//
//   X    [B*T, E]   = emb[tokens]                                   gather (+ optional normalise)
//   per step t:  gates = [x_t | h_d] x [Wih_d | Whh_d]^T  (one chain over E + H)  + (bih_d + bhh_d)
//                i,f,o = sigmoid, g = tanh; c' = fma(f, c, i*g); h' = o * tanh(c')
//                -- ONE MFMA launch per step for both directions: segmented-K loader, gate-permuted tile
//                   columns, pointwise epilogue (no hoisted Gin array, no gates array, no pointwise launch)
//   out  [B, D]     = [h_fwd | h_bwd] x Wfc^T + bfc
//
// 352.4 MFLOP per query; the recurrent GEMM [B,H]x[H,4H] is the MFMA-bound part.
#include <atomic>
#include <type_traits>

#include "vfr_common.h"
#include "vfr_math.h"

namespace vfr {

// plain gather (no learnable length): one thread per ELEMENT (a thread per row copied its E floats one after the other: 10 us
// for the 20 rows of a single-query request)
__global__ __launch_bounds__(256) void embed_gather_kernel(const int64_t *__restrict__ tokens, int64_t rows, int vocab,
                                                           const float *__restrict__ emb, int E, float *__restrict__ X)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * E) return;
    const int64_t r = i / E;
    const int k = (int)(i - r * E);
    int64_t tok = tokens ? tokens[r] : r;
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
    X[i] = emb[tok * E + k];
}
__global__ __launch_bounds__(256) void embed_kernel(const int64_t *__restrict__ tokens, int64_t rows, int vocab,
                                                    const float *__restrict__ emb, const float *__restrict__ len_tab,
                                                    int E, float *__restrict__ X)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    int64_t tok = tokens ? tokens[r] : r;                    // null: row r is vocabulary entry r
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);   // never read outside the table
    const float *e = emb + tok * E;
    float *x = X + r * E;
    if (!len_tab) {
        for (int k = 0; k < E; ++k) x[k] = e[k];
        return;
    }
    float acc = 0.0f;
    for (int k = 0; k < E; ++k) acc = __builtin_fmaf(e[k], e[k], acc);
    const float nrm = __builtin_sqrtf(acc) + 1e-5f, len = len_tab[tok];
    for (int k = 0; k < E; ++k) x[k] = (e[k] / nrm) * len;
}

static void launch_embed(const int64_t *tokens, int64_t rows, int vocab, const float *emb, const float *len_tab, int E, float *X,
                         hipStream_t st)
{
    if (!len_tab)
        hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)cdiv(rows * E, 256)), dim3(256), 0, st, tokens, rows, vocab, emb, E, X);
    else
        hipLaunchKernelGGL(embed_kernel, dim3((unsigned)cdiv(rows, 256)), dim3(256), 0, st, tokens, rows, vocab, emb, len_tab, E, X);
}

// gates [2][B,4H] (chains without bias) -> + (b_ih + b_hh) -> c [2][B,H] (in place), h into hcat [B,2H] at column d*H
__global__ __launch_bounds__(256) void lstm_pointwise_kernel(const float *__restrict__ gates, const float *__restrict__ bih_f,
                                                             const float *__restrict__ bhh_f, const float *__restrict__ bih_b,
                                                             const float *__restrict__ bhh_b, float *__restrict__ c,
                                                             float *__restrict__ hcat, int64_t B, int H)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t per = B * H;
    if (i >= 2 * per) return;
    int d = (int)(i / per);
    int64_t r = i - d * per, b = r / H;
    int j = (int)(r - b * H);
    const float *bih = d ? bih_b : bih_f, *bhh = d ? bhh_b : bhh_f;
    const float *g4 = gates + ((int64_t)d * B + b) * 4 * H;
    float ig = c_sigmoidf(g4[j] + (bih[j] + bhh[j]));
    float fg = c_sigmoidf(g4[H + j] + (bih[H + j] + bhh[H + j]));
    float gg = c_tanhf(g4[2 * H + j] + (bih[2 * H + j] + bhh[2 * H + j]));
    float og = c_sigmoidf(g4[3 * H + j] + (bih[3 * H + j] + bhh[3 * H + j]));
    float cn = __builtin_fmaf(fg, c[i], ig * gg);
    c[i] = cn;
    hcat[b * 2 * H + (int64_t)d * H + j] = og * c_tanhf(cn);
}

// ---- row bookkeeping for the fused path --------------------------------------------------------------------------
// Trailing pads: in the REVERSE direction a query first consumes its trailing pad tokens starting from the zero state,
// so during that prefix its (h, c) depends only on the number of pad steps -- it is the state of an all-pad query.
// That evolution is computed once (GEMM row 0 = a virtual all-pad query); queries are sorted by length (descending)
// so the rows that have reached a real token form a prefix, and a query joins at its last real token with row 0's
// state.  Each row's arithmetic is unchanged, so the result is bit-identical to stepping every query through its pads.
__global__ __launch_bounds__(256) void query_length_kernel(const int64_t *__restrict__ tokens, int64_t B, int T,
                                                           int *__restrict__ len)
{
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int l = 0;
    for (int t = 0; t < T; ++t)
        if (tokens[b * T + t] != 0) l = t + 1;          // position of the last non-pad token + 1
    len[b] = l;
}

// stable descending-length order: GEMM row m = 1 + (#queries longer than b) + (#earlier queries of the same length);
// row 0 is the virtual all-pad query (index B in the extended token array).  Two small kernels: per-block length
// histograms, then every query sums the bins above it, the earlier blocks' bin and its in-block predecessors.
constexpr int SORT_BLOCK = 256, SORT_BINS = 1025;          // lengths 0..T, T <= 1024
__global__ __launch_bounds__(SORT_BLOCK) void length_hist_kernel(const int *__restrict__ len, int64_t B, int T,
                                                                 int *__restrict__ hist /*[blocks][T+1]*/)
{
    __shared__ int h[SORT_BINS];
    for (int i = threadIdx.x; i <= T; i += SORT_BLOCK) h[i] = 0;
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x;
    if (b < B) atomicAdd(&h[len[b]], 1);
    __syncthreads();
    for (int i = threadIdx.x; i <= T; i += SORT_BLOCK) hist[(size_t)blockIdx.x * (T + 1) + i] = h[i];
}
__global__ __launch_bounds__(SORT_BLOCK) void sort_rows_kernel(const int *__restrict__ len, const int *__restrict__ hist,
                                                               int64_t B, int T, int *__restrict__ row_of,
                                                               int *__restrict__ xrow)
{
    const int64_t b = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x;
    if (b == 0) xrow[0] = (int)B;
    if (b >= B) return;
    const int lb = len[b], nblk = (int)((B + SORT_BLOCK - 1) / SORT_BLOCK);
    int pos = 0;
    for (int k = 0; k < nblk; ++k) {
        const int *hk = hist + (size_t)k * (T + 1);
        for (int l = lb + 1; l <= T; ++l) pos += hk[l];              // every longer query, any block
        if (k < (int)blockIdx.x) pos += hk[lb];                      // same length, earlier block
    }
    for (int64_t o = (int64_t)blockIdx.x * SORT_BLOCK; o < b; ++o) pos += len[o] == lb ? 1 : 0;   // same length, same block
    row_of[b] = pos + 1;
    xrow[pos + 1] = (int)b;
}

// mcount[s] = 1 + #{b : len[b] > T-1-s}: active rows of the reverse direction at step s (from the per-block histograms)
__global__ void active_rows_kernel(const int *__restrict__ hist, int nblk, int T, int *__restrict__ mcount)
{
    const int s = threadIdx.x;
    if (s >= T) return;
    int c = 1;
    for (int k = 0; k < nblk; ++k)
        for (int l = T - s; l <= T; ++l) c += hist[(size_t)k * (T + 1) + l];
    mcount[s] = c;
}

// [B+1 sorted rows, 2H] -> [B, 2H] in query order; an all-pad query's reverse half is the pad row's
// table row of GEMM row m at time t: the (clamped) token of query xrow[m]  ->  tokidx [T][R]
__global__ __launch_bounds__(256) void token_index_kernel(const int64_t *__restrict__ tok_ext, const int *__restrict__ xrow,
                                                          int64_t R, int T, int vocab, int *__restrict__ tokidx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * T) return;
    const int t = (int)(i / R);
    const int64_t m = i - (int64_t)t * R;
    int64_t tok = tok_ext[(int64_t)xrow[m] * T + t];
    tokidx[i] = (int)(tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok));
}

__global__ __launch_bounds__(256) void unsort_rows_kernel(const float *__restrict__ hs, const int *__restrict__ row_of,
                                                          const int *__restrict__ len, int64_t B, int H,
                                                          float *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 2 * H) return;
    const int64_t b = i / (2 * H);
    const int j = (int)(i - b * 2 * H);
    const int64_t src = (len && j >= H && len[b] == 0) ? 0 : (row_of ? (int64_t)row_of[b] : b);
    out[i] = hs[src * 2 * H + j];
}

// ---- a handful of queries (one serving request): the step as a vector chain ----------------------------------------
// With B <= 4 rows a 32-row MFMA tile is 3-13 % full and, worse, a wave walks its K = E + H chain 4 MFMAs (128 cycles) per
// 4 k's: 34 us per step whatever the batch.  Here a lane owns ONE gate column of one direction and runs the canonical chain
// itself -- acc = fma(x_k, W_ih[c][k], acc) over E, then fma(h_k, W_hh[c][k], acc) over H -- for all RB rows: [x_t | h]
// of the rows is staged in LDS once and read back as broadcasts (scalar loads cannot be kept in flight: every chunk paid
// their latency), the weights come from a chunk-major copy [(E + H) / 4][4H][4] made once per call (one coalesced 16-byte
// load per lane and 4 k's), fetched PF chunks ahead.  16 us per step at one query, 25 at two (default limit), 38 at four.  (hipcc drains all
// outstanding loads at the head of the chunk loop -- one latency per 32 chunks; a single launch for all T steps with an
// arrival counter between steps was built and measured: 20 us per step, slower than the launches, and removed.)  A wave = 4 gates x 16 units, so the four pre-activations of a
// unit meet through 1 KB of LDS and 16 lanes finish the cells.  Same chains, same bits (test: lstm_small = 0 vs 4).
struct SmallLstm {
    const float *X;                       // [B*T, E] embedded tokens
    const float *WT[2];                   // [(E + H) / 4][4H][4]: chunk c, column col, k = 4c..4c+3 (W_ih over E, then W_hh)
    const float *bih[2], *bhh[2];
    const float *hin, *cin;               // h [B, 2H] (direction d at column d*H), c [2][B][H]
    float *hout, *cout;
    int B, T, E, H, step, recurrent;      // recurrent = 0: first step, h_0 = 0 (no h terms in the chain)
};
// W [rows, K] row-major -> out [(K/4)][rows][4]: the four consecutive k's of a column adjacent, columns consecutive
__global__ __launch_bounds__(256) void pack_k4_kernel(const float *__restrict__ W, int64_t rows, int K, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (k4, row), row fastest: coalesced stores
    const int64_t nk4 = K / 4;
    if (i >= nk4 * rows) return;
    const int64_t k4 = i / rows, row = i - k4 * rows;
    reinterpret_cast<float4 *>(out)[i] = *reinterpret_cast<const float4 *>(W + row * K + 4 * k4);
}

template <int RB>
__global__ __launch_bounds__(64) void lstm_step_small_kernel(SmallLstm a)
{
    // chunks (of 4 k's, one 16-byte load per lane) in flight: the weights stream from L2 / Infinity Cache at ~0.6 us latency,
    // and with one wave per CU only what is in flight counts -- 8 chunks gave 25 us per step, the chip-wide 1 MB in flight
    constexpr int PF = 32, XF = 4;        // XF: LDS broadcast reads of [x | h] issued this many chunks ahead
    __shared__ float pre[RB][4][16];
    extern __shared__ __attribute__((aligned(16))) float xh[];          // [RB][E + H]: the rows' chain inputs
    const int lane = threadIdx.x, gate = lane >> 4, ul = lane & 15, d = blockIdx.y;
    const int unit = blockIdx.x * 16 + ul, H = a.H, E = a.E, G = 4 * H;
    const int uc = unit < H ? unit : H - 1;
    const int t = d ? a.T - 1 - a.step : a.step;
    const float4 *wt = reinterpret_cast<const float4 *>(a.WT[d]) + (size_t)gate * H + uc;
    const float *xr[RB], *hr[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int rr = r < a.B ? r : 0;
        xr[r] = a.X + ((size_t)rr * a.T + t) * E;
        hr[r] = a.hin + (size_t)rr * 2 * H + (size_t)d * H - E;          // indexed by the chain position k >= E
    }
    const int nchunk = (a.recurrent ? E + H : E) / 4;
    float acc[RB];
    float4 w[PF], xq[XF][RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.0f;
    auto fetch = [&](int slot, int c) {                                  // chunk index clamped: a redundant load, no branch
        const int cc = c < nchunk ? c : nchunk - 1;
        w[slot] = wt[(size_t)cc * G];
    };
    auto xfetch = [&](int slot, int c) {
        const int cc = c < nchunk ? c : nchunk - 1;
#pragma unroll
        for (int r = 0; r < RB; ++r) xq[slot][r] = *reinterpret_cast<const float4 *>(xh + r * (E + H) + 4 * cc);   // LDS broadcast
    };
    auto chain = [&](int slot, int xs) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float4 x = xq[xs][r];
            acc[r] = __builtin_fmaf(x.x, w[slot].x, acc[r]); acc[r] = __builtin_fmaf(x.y, w[slot].y, acc[r]);
            acc[r] = __builtin_fmaf(x.z, w[slot].z, acc[r]); acc[r] = __builtin_fmaf(x.w, w[slot].w, acc[r]);
        }
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) fetch(j, j);
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll 2
        for (int c = lane; c < nchunk; c += 64)
            *reinterpret_cast<float4 *>(xh + r * (E + H) + 4 * c) = *reinterpret_cast<const float4 *>((4 * c < E ? xr[r] : hr[r]) + 4 * c);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < XF; ++j) xfetch(j, j);
    int base = 0;
    for (; base + PF <= nchunk; base += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            chain(j, j % XF); fetch(j, base + j + PF); xfetch(j % XF, base + j + XF);
            __builtin_amdgcn_sched_barrier(0);          // keep every refill right behind its chunk (hipcc sinks them to the loop end)
        }
    }
#pragma unroll
    for (int j = 0; j < PF; ++j)
        if (base + j < nchunk) { chain(j, j % XF); xfetch(j % XF, base + j + XF); }
#pragma unroll
    for (int r = 0; r < RB; ++r) pre[r][gate][ul] = acc[r];
    __syncthreads();
    // cells: lane (row r = lane >> 4, unit ul); RB <= 4 rows fit one wave
    const int r = lane >> 4;
    if (r < RB && r < a.B && unit < H) {
        const float *bi = a.bih[d], *bh = a.bhh[d];
        const float ig = c_sigmoidf(pre[r][0][ul] + (bi[unit] + bh[unit]));
        const float fg = c_sigmoidf(pre[r][1][ul] + (bi[H + unit] + bh[H + unit]));
        const float gg = c_tanhf(pre[r][2][ul] + (bi[2 * H + unit] + bh[2 * H + unit]));
        const float og = c_sigmoidf(pre[r][3][ul] + (bi[3 * H + unit] + bh[3 * H + unit]));
        const size_t ci = ((size_t)d * a.B + r) * H + unit;
        const float cn = __builtin_fmaf(fg, a.cin[ci], ig * gg);
        a.cout[ci] = cn;
        a.hout[(size_t)r * 2 * H + (size_t)d * H + unit] = og * c_tanhf(cn);
    }
}

// Vocabularies up to this many entries get the input-projection table (P = emb x W_ih^T per entry, 2 directions):
// 2 x 32768 x 4096 floats = 1 GB at H = 1000 -- nothing on a 288 GB part, and far fewer rows than B*T once B is large.
constexpr int VOCAB_TABLE_MAX = 32768;
static bool use_vocab_table(int64_t B, int T, int vocab) { return vocab <= VOCAB_TABLE_MAX && (int64_t)vocab <= 4 * B * T; }

// The same step with the weight stream spread over four waves (one query, the model's shape).  The single-wave kernel above is
// bound by what ONE wave per CU can keep in flight: 32 chunks x 1 KB x 126 waves = 4 MB chip-wide against ~0.6 us of latency
// = 13 us for the 35 MB of weights a step reads.  Here waves 1-3 of a 256-thread workgroup only LOAD: chunk c of phase p goes to
// loader c mod 3, which keeps two phases of 16 chunks in registers (in flight) and hands a finished phase to wave 0 through a
// double-buffered 2 x 48 KB LDS ring; wave 0 runs the chain exactly as above (same order, same bits).  One __syncthreads per
// phase of 48 chunks, no polling: every wave reaches every barrier.  NCH (chunks of the chain) is a template constant so the
// phase loop is straight-line code (a loop head would drain the loads in flight).
#ifndef VFR_S4_NL
#define VFR_S4_NL 3
#define VFR_S4_PER 16
#define VFR_S4_AHEAD 2
#endif
constexpr int S4_NL = VFR_S4_NL, S4_PER = VFR_S4_PER, S4_AHEAD = VFR_S4_AHEAD;   // loader waves; chunks per loader and phase; phases in flight
constexpr int S4_PH = S4_NL * S4_PER;                                              // chunks per phase
template <int RB, int NCH>
__global__ __launch_bounds__(64 * (S4_NL + 1), 1) void lstm_step_small4_kernel(SmallLstm a)
{
    constexpr int NPH = (NCH + S4_PH - 1) / S4_PH, XF = 4;
    __shared__ float pre[RB][4][16];
    __shared__ __attribute__((aligned(16))) float4 ring[2][S4_PH][64];
    extern __shared__ __attribute__((aligned(16))) float xh[];          // [RB][E + H]: the rows' chain inputs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gate = lane >> 4, ul = lane & 15, d = blockIdx.y;
    const int unit = blockIdx.x * 16 + ul, H = a.H, E = a.E, G = 4 * H;
    const int uc = unit < H ? unit : H - 1;
    const int t = d ? a.T - 1 - a.step : a.step;
    const float4 *wt = reinterpret_cast<const float4 *>(a.WT[d]) + (size_t)gate * H + uc;
    // ---- stage [x_t | h] of the rows (all waves) ----
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int rr = r < a.B ? r : 0;
        const float *xr = a.X + ((size_t)rr * a.T + t) * E;
        const float *hr = a.hin + (size_t)rr * 2 * H + (size_t)d * H - E;
        for (int c = tid; c < NCH; c += 64 * (S4_NL + 1))
            *reinterpret_cast<float4 *>(xh + r * (E + H) + 4 * c) = *reinterpret_cast<const float4 *>((4 * c < E ? xr : hr) + 4 * c);
    }
    float acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.0f;
    if (wv > 0) {
        // ---- loaders: chunk c = S4_PH p + S4_NL j + (wv - 1), S4_AHEAD phases in flight (one register buffer per phase in flight) ----
        const int l = wv - 1;
        float4 bufs[S4_AHEAD][S4_PER];
        // (a running pointer, opaque to the compiler: with constant chunk indices it precomputes every address into a register
        // pair; chunks past the end re-read the last one: a redundant load, no branch)
        const float4 *wp = wt + (size_t)l * G;
        const float4 *const wlast = wt + (size_t)(NCH - 1) * G;
        auto next = [&]() -> float4 {
            asm volatile("" : "+v"(wp));
            const float4 v = *(wp < wlast ? wp : wlast);
            wp += S4_NL * (size_t)G;
            return v;
        };
        auto issue = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
#pragma unroll
            for (int j = 0; j < S4_PER; ++j) bufs[k][j] = next();
        };
        auto hand = [&](auto kc, auto hc) {
            constexpr int k = decltype(kc)::value, h = decltype(hc)::value;
#pragma unroll
            for (int j = 0; j < S4_PER; ++j) ring[h][S4_NL * j + l][lane] = bufs[k][j];
        };
        auto for_phase = [&](auto pc) {
            constexpr int p = decltype(pc)::value;
            hand(std::integral_constant<int, p % S4_AHEAD>{}, std::integral_constant<int, p & 1>{});
            if constexpr (p + S4_AHEAD < NPH) issue(std::integral_constant<int, p % S4_AHEAD>{});
            __syncthreads();                                             // phase p is in the ring (p = 0: the staged rows as well)
        };
        issue(std::integral_constant<int, 0>{});
        if constexpr (NPH > 1 && S4_AHEAD > 1) issue(std::integral_constant<int, 1>{});
        if constexpr (NPH > 2 && S4_AHEAD > 2) issue(std::integral_constant<int, 2>{});
        static_assert(NPH <= 12 && S4_AHEAD <= 3, "phase loop written out for up to 12 phases");
#define S4_P(N) if constexpr (N < NPH) for_phase(std::integral_constant<int, N>{});
        S4_P(0) S4_P(1) S4_P(2) S4_P(3) S4_P(4) S4_P(5) S4_P(6) S4_P(7) S4_P(8) S4_P(9) S4_P(10) S4_P(11)
#undef S4_P
        __syncthreads();                                                 // (matches the compute wave's barrier before the cells)
    } else {
        // ---- wave 0: the chain ----
        float4 xq[XF][RB];
        auto xfetch = [&](int slot, int c) {
            const int cc = c < NCH ? c : NCH - 1;
#pragma unroll
            for (int r = 0; r < RB; ++r) xq[slot][r] = *reinterpret_cast<const float4 *>(xh + r * (E + H) + 4 * cc);   // LDS broadcast
        };
#pragma unroll
        for (int p = 0; p < NPH; ++p) {
            __syncthreads();
            if (p == 0) {
#pragma unroll
                for (int j = 0; j < XF; ++j) xfetch(j, j);
            }
#pragma unroll
            for (int i = 0; i < S4_PH; ++i) {
                const int c = S4_PH * p + i;
                if (c < NCH) {
                    const float4 w = ring[p & 1][i][lane];
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        const float4 x = xq[c % XF][r];
                        acc[r] = __builtin_fmaf(x.x, w.x, acc[r]); acc[r] = __builtin_fmaf(x.y, w.y, acc[r]);
                        acc[r] = __builtin_fmaf(x.z, w.z, acc[r]); acc[r] = __builtin_fmaf(x.w, w.w, acc[r]);
                    }
                    xfetch(c % XF, c + XF);
                    if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // (eight ring reads in flight, not forty-eight)
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) pre[r][gate][ul] = acc[r];
        __syncthreads();
        // cells: lane (row r = lane >> 4, unit ul); RB <= 4 rows fit one wave
        const int r = lane >> 4;
        if (r < RB && r < a.B && unit < H) {
            const float *bi = a.bih[d], *bh = a.bhh[d];
            const float ig = c_sigmoidf(pre[r][0][ul] + (bi[unit] + bh[unit]));
            const float fg = c_sigmoidf(pre[r][1][ul] + (bi[H + unit] + bh[H + unit]));
            const float gg = c_tanhf(pre[r][2][ul] + (bi[2 * H + unit] + bh[2 * H + unit]));
            const float og = c_sigmoidf(pre[r][3][ul] + (bi[3 * H + unit] + bh[3 * H + unit]));
            const size_t ci = ((size_t)d * a.B + r) * H + unit;
            const float cn = __builtin_fmaf(fg, a.cin[ci], ig * gg);
            a.cout[ci] = cn;
            a.hout[(size_t)r * 2 * H + (size_t)d * H + unit] = og * c_tanhf(cn);
        }
    }
}

// ---- one or two queries, the whole sequence in ONE launch with the weights resident in LDS ---------------------------
// The per-step kernels above re-read all 35 MB of [W_ih | W_hh] (both directions) every step: 13 us per step whatever is done
// about the stream, 20 dependent launches = 0.27 ms of a 0.49 ms serving request.  Here a workgroup owns 8 hidden units of
// one direction = 32 gate columns and keeps their (E + H) x 32 weights in LDS for all T steps (chunk-major, 140.8 KB at the
// model's shape: 2 x 125 workgroups, one per CU); per step its first wave runs the 32 canonical chains -- lane = gate column,
// lanes 32..63 the second query on the same weights; the E-part before the previous step's h is needed --, finishes its 8
// cells (c stays in a register for the whole sequence) and publishes the 8 new h values as 8-byte {step tag, value} granules
// (agent-scope relaxed atomic stores: write-through, untorn); every workgroup of the direction then sweeps the H granules of
// the step with agent-scope atomic loads until every tag matches -- the data is the flag, no counter, no fence, no plain load
// of handed-off bytes (MI355X guide, Guideline 16 form R2).  Two granule buffers alternate: a workgroup can run at most one
// step ahead of the slowest one (it needs that one's h to go further).  Same chains in the same order as the step kernels:
// identical bits (tests: lstm_persist 0 vs 1, and the oracle).
// Every workgroup must be resident (the host checks grid <= CU count; 150 KB of LDS = one workgroup per CU); the sweeps are
// bounded: a workgroup that gives up raises a.err, poisons its outputs with NaN and leaves, and so do all the others -- and
// lstm_seq_rescue_kernel, enqueued behind every launch, then re-encodes the batch (below): the CALL never returns NaN.
typedef __attribute__((address_space(1))) unsigned long long seq_gu64;
struct SeqLstm {
    const float *X;                       // [B*T, E] embedded tokens
    const float *Wih[2], *Whh[2], *bih[2], *bhh[2];
    unsigned long long *hg;               // granules [2 buffers][2 directions][B][H], zeroed before the launch
    float *hout;                          // [B, 2H]: the final h (direction d at column d*H)
    unsigned *err;                        // raised when a sweep gave up (zeroed before the launch)
    int B, T, E, H;
    const float *Wfc, *bfc;               // lang_fc fused behind the last step (nullptr: the caller runs it): [D, 2H], [D]
    float *out;                           // [B, D]
    int D;
    int fault_block;                      // TEST HOOK (-1: off): this workgroup never publishes its h of step 1 -- the others must give up, not hang
};
constexpr int SEQ_WSTRIDE = 33 * 4;       // floats per chunk row: 32 columns x 4 k's + 4 of padding (staging writes conflict-free)
constexpr unsigned SEQ_MAX_SPINS = 1u << 17;

template <int RB>
__global__ __launch_bounds__(256, 1) void lstm_seq_small_kernel(SeqLstm a)
{
    constexpr int NG = 16 * RB;           // granule loads per lane and sweep: covers B x H <= 1024 x RB
    extern __shared__ __attribute__((aligned(16))) float seq_lds[];
    const int E = a.E, H = a.H, nce = E / 4, nch = (E + H) / 4;
    float *Wl = seq_lds;                  // [nch][33][4]
    float *xh = Wl + (size_t)nch * SEQ_WSTRIDE;      // [RB][E + H]: this step's embedded token | the previous step's h
    float *pre = xh + RB * (E + H);       // [RB][4][8] gate pre-activations
    const int tid = threadIdx.x, lane = tid & 63;
    const int nblk = (H + 7) / 8, d = blockIdx.x / nblk, u0 = (blockIdx.x % nblk) * 8;
    // ---- weights of the 32 columns -> LDS (all four waves; consecutive lanes = consecutive chunks of one column) ----
    {
        const int total = 32 * nch;
        for (int i0 = 0; i0 < total; i0 += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int i = i0 + 256 * j + tid;
                i = i < total ? i : total - 1;
                const int c = i / nch, ch = i - c * nch;
                int unit = u0 + (c & 7);
                unit = unit < H ? unit : H - 1;
                const size_t row = (size_t)(c >> 3) * H + unit;
                v[j] = ch < nce ? *reinterpret_cast<const float4 *>(a.Wih[d] + row * E + 4 * ch)
                                : *reinterpret_cast<const float4 *>(a.Whh[d] + row * H + 4 * (ch - nce));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int i = i0 + 256 * j + tid;
                i = i < total ? i : total - 1;
                const int c = i / nch, ch = i - c * nch;
                *reinterpret_cast<float4 *>(Wl + (size_t)ch * SEQ_WSTRIDE + c * 4) = v[j];
            }
        }
    }
    __syncthreads();
    if (tid >= 64) return;                // the sequence is one wave's work
    const int col = lane & 31, r = lane >> 5, gate = col >> 3, ul = col & 7;
    const bool rowok = r < RB && r < a.B;
    const int rr = rowok ? r : 0;
    const int unit = u0 + ul, uc = unit < H ? unit : H - 1;
    const float *wl = Wl + col * 4;
    const size_t gdir = (size_t)a.B * H;                                   // granules per (buffer, direction)
    // cell lanes: column lanes of gate 0 (col < 8) finish unit u0 + col of row r
    const bool cell = col < 8 && rowok;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f}, cst = 0.0f;
    if (cell) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bsum[g] = a.bih[d][g * H + uc] + a.bhh[d][g * H + uc];
    }
    bool dead = false;
    float acc = 0.0f;
    const float *xrow = xh + rr * (E + H);
    // chunks [c0, c1) of the lane's chain: groups of 5 chunks in two register sets, the next group's LDS reads (weights: one
    // 16-byte read per lane; inputs: broadcast) issued before the current group's 20 dependent fmas
    auto chain = [&](int c0, int c1) {
        constexpr int G = 5;
        float4 w0[G], x0[G], w1[G], x1[G];
        auto load = [&](float4 (&w)[G], float4 (&x)[G], int c) {
            c = c + G <= c1 ? c : c1 - G;                        // (the group past the end: re-read the last one, unused)
#pragma unroll
            for (int j = 0; j < G; ++j) {
                w[j] = *reinterpret_cast<const float4 *>(wl + (size_t)(c + j) * SEQ_WSTRIDE);
                x[j] = *reinterpret_cast<const float4 *>(xrow + 4 * (c + j));
            }
        };
        auto fma5 = [&](const float4 (&w)[G], const float4 (&x)[G]) {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                acc = __builtin_fmaf(x[j].x, w[j].x, acc); acc = __builtin_fmaf(x[j].y, w[j].y, acc);
                acc = __builtin_fmaf(x[j].z, w[j].z, acc); acc = __builtin_fmaf(x[j].w, w[j].w, acc);
            }
        };
        int ch = c0;
        if (c1 - c0 >= G) {
            load(w0, x0, ch);
            while (true) {
                load(w1, x1, ch + G);
                __builtin_amdgcn_sched_barrier(0);
                fma5(w0, x0);
                ch += G;
                if (ch + G > c1) break;
                load(w0, x0, ch + G);
                __builtin_amdgcn_sched_barrier(0);
                fma5(w1, x1);
                ch += G;
                if (ch + G > c1) break;
            }
        }
        for (; ch < c1; ++ch) {
            const float4 w = *reinterpret_cast<const float4 *>(wl + (size_t)ch * SEQ_WSTRIDE);
            const float4 x = *reinterpret_cast<const float4 *>(xrow + 4 * ch);
            acc = __builtin_fmaf(x.x, w.x, acc); acc = __builtin_fmaf(x.y, w.y, acc);
            acc = __builtin_fmaf(x.z, w.z, acc); acc = __builtin_fmaf(x.w, w.w, acc);
        }
    };
    // embedded tokens: the next step's are requested a step ahead (RB * E <= 256: four registers per lane)
    float xn[4];
    auto xload = [&](int step) {
        const int t = d ? a.T - 1 - step : step;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int i = lane + 64 * j;
            i = i < RB * E ? i : RB * E - 1;
            const int xr = i / E, k = i - xr * E;
            xn[j] = a.X[((size_t)(xr < a.B ? xr : 0) * a.T + (t < 0 ? 0 : (t >= a.T ? a.T - 1 : t))) * E + k];
        }
    };
    xload(0);
    for (int step = 0; step < a.T; ++step) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int i = lane + 64 * j;
            i = i < RB * E ? i : RB * E - 1;
            const int xr = i / E, k = i - xr * E;
            xh[xr * (E + H) + k] = xn[j];
        }
        xload(step + 1 < a.T ? step + 1 : step);
        acc = 0.0f;
        chain(0, nce);
        if (step > 0 && !dead) {
            // ---- sweep the H granules of step - 1 (tag = step) of this direction until all are there ----
            seq_gu64 *gp = (seq_gu64 *)(a.hg + ((size_t)((step - 1) & 1) * 2 + d) * gdir);
            const int n = a.B * H;
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
                unsigned long long gv[NG];
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const int g = lane + 64 * j;
                    gv[j] = __hip_atomic_load(gp + (g < n ? g : n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    int g = lane + 64 * j;                               // (past the end: the last granule again, same value)
                    g = g < n ? g : n - 1;
                    ok &= (unsigned)(gv[j] >> 32) == (unsigned)step;
                    const int gr = g / H;
                    xh[gr * (E + H) + E + (g - gr * H)] = __uint_as_float((unsigned)gv[j]);
                }
                if (__all(ok)) break;
                if (spins >= SEQ_MAX_SPINS) { dead = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (!dead) chain(nce, nch);
        }
        if (r < RB) pre[(r * 4 + gate) * 8 + ul] = acc;
        float hn = 0.0f;
        if (cell) {
            const float *pr = pre + r * 32 + ul;
            const float ig = c_sigmoidf(pr[0] + bsum[0]);
            const float fg = c_sigmoidf(pr[8] + bsum[1]);
            const float gg = c_tanhf(pr[16] + bsum[2]);
            const float og = c_sigmoidf(pr[24] + bsum[3]);
            cst = __builtin_fmaf(fg, cst, ig * gg);
            hn = og * c_tanhf(cst);
            if (dead) hn = __uint_as_float(0x7fc00000u);
            if (unit < H && !(step == 1 && (int)blockIdx.x == a.fault_block)) {
                if (step + 1 < a.T || a.Wfc) {
                    seq_gu64 *gq = (seq_gu64 *)(a.hg + ((size_t)(step & 1) * 2 + d) * gdir + (size_t)r * H + unit);
                    __hip_atomic_store(gq, ((unsigned long long)(unsigned)(step + 1) << 32) | __float_as_uint(hn),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (step + 1 == a.T) a.hout[(size_t)r * 2 * H + (size_t)d * H + unit] = hn;
            }
        }
    }
    // ---- lang_fc behind the last step: workgroup o < D gathers the final [h_fwd | h_bwd] of every row (the granules of step
    // T - 1, tag T, both directions) and ONE lane per row runs output o's chain over 2H (k ascending from zero, then + bias:
    // gemm_nt's order) -- 4 us instead of a launch + a 32-row MFMA tile (0.05 ms) ----
    if (a.Wfc && (int)blockIdx.x < a.D) {
        const int o = blockIdx.x, K2 = 2 * H;
        float *wrow = seq_lds;                           // [2H]        (the weight slices are no longer needed)
        float *hfin = seq_lds + K2;                      // [RB][2H]
        for (int i = lane; i < K2 / 4; i += 64)
            *reinterpret_cast<float4 *>(wrow + 4 * i) = *reinterpret_cast<const float4 *>(a.Wfc + (size_t)o * K2 + 4 * i);
        if (!dead) {
            const size_t gb = (size_t)((a.T - 1) & 1) * 2 * gdir;
            const int n = a.B * H;
            for (int dd = 0; dd < 2 && !dead; ++dd) {
                seq_gu64 *gp = (seq_gu64 *)(a.hg + gb + (size_t)dd * gdir);
                for (unsigned spins = 0;; ++spins) {
                    bool ok = true;
                    unsigned long long gv[NG];
#pragma unroll
                    for (int j = 0; j < NG; ++j) {
                        const int g = lane + 64 * j;
                        gv[j] = __hip_atomic_load(gp + (g < n ? g : n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int j = 0; j < NG; ++j) {
                        int g = lane + 64 * j;
                        g = g < n ? g : n - 1;
                        ok &= (unsigned)(gv[j] >> 32) == (unsigned)a.T;
                        const int gr = g / H;
                        hfin[gr * K2 + dd * H + (g - gr * H)] = __uint_as_float((unsigned)gv[j]);
                    }
                    if (__all(ok)) break;
                    if (spins >= SEQ_MAX_SPINS) { dead = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        if (lane < RB && lane < a.B) {
            const float *hp = hfin + lane * K2;
            float s = 0.0f;
#pragma unroll 8
            for (int k4 = 0; k4 < K2 / 4; ++k4) {
                const float4 w = *reinterpret_cast<const float4 *>(wrow + 4 * k4);
                const float4 x = *reinterpret_cast<const float4 *>(hp + 4 * k4);
                s = __builtin_fmaf(x.x, w.x, s); s = __builtin_fmaf(x.y, w.y, s);
                s = __builtin_fmaf(x.z, w.z, s); s = __builtin_fmaf(x.w, w.w, s);
            }
            s = s + a.bfc[o];
            a.out[(size_t)lane * a.D + o] = dead ? __uint_as_float(0x7fc00000u) : s;
        }
    }
    if (dead && lane == 0) atomicOr(a.err, 1u);
}

// ---- 3 .. 32 queries: the same single-launch sequence on the matrix pipe, the weights resident in REGISTERS ------------
// The MFMA tile step at these batch sizes is one lone workgroup per CU paying a load latency per K-tile: 31-34 us per step
// whatever the batch.  Here a workgroup again owns 8 hidden units of one direction (32 gate columns); wave (rt, ct) holds column
// tile ct (16 columns) of [W_ih | W_hh] as the B fragments of all 275 k-steps IN ITS REGISTERS for the whole sequence (275 of
// the wave's 512: one wave per SIMD) and row tile rt (16 queries) of [x_t | h] comes from LDS as A fragments (row stride 1100
// floats: 16 rows x 4 k's hit 64 distinct banks).  One v_mfma_f32_16x16x4_f32 per k-step on top of the accumulator = the
// canonical chain, bit for bit.  Gates meet through LDS, thread (query, unit) finishes its cell (c in a register for the whole
// sequence), h travels as tagged granules exactly as in lstm_seq_small_kernel; every wave sweeps a quarter of the B x H granules.
// NP > 1: the batch is run as NP parts of up to 16 RT queries that take turns on the same register-resident weights and the same
// LDS rows (33 .. 64 queries: two parts).  A part's step needs only THAT part's h of the previous step, which every workgroup
// published before it went on to the other part -- the hand-off of one part travels while the other computes.
template <int RT, int NP, int NCE, int NCH>
__global__ __launch_bounds__(256, 1) void lstm_seq_mfma_kernel(SeqLstm a)
{
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    constexpr int RB = 16 * RT, E = 4 * NCE, H = 4 * (NCH - NCE), KX = E + H, NGL = 16;
    extern __shared__ __attribute__((aligned(16))) float seq_lds[];
    float *xh = seq_lds;                  // [RB][KX]: this step's embedded token | the previous step's h, per query of the part
    float *pre = xh + RB * KX;            // [RB][32] gate pre-activations
    int &s_dead = *reinterpret_cast<int *>(pre + RB * 32);      // (dynamic too: a static word would push the total past what the attribute admits)
    float *wrow = pre + RB * 32 + 4;      // [2H] lang_fc row of output blockIdx.x (fused lang_fc only)
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wv & 1, rt = wv >> 1;
    const bool comp = rt < RT;
    const int nblk = (H + 7) / 8, d = blockIdx.x / nblk, u0 = (blockIdx.x % nblk) * 8;
    const int B = a.B;
    // ---- the wave's 16 columns of [W_ih | W_hh] -> registers, as the B fragments of the k-steps ----
    float wreg[NCH];
    {
        const int c = ct * 16 + l15;
        int unit = u0 + (c & 7);
        unit = unit < H ? unit : H - 1;
        const size_t row = (size_t)(c >> 3) * H + unit;
        const float *wi = a.Wih[d] + row * E + lq, *wh = a.Whh[d] + row * H + lq;
#pragma unroll
        for (int s = 0; s < NCH; ++s) wreg[s] = comp ? (s < NCE ? wi[4 * s] : wh[4 * (s - NCE)]) : 0.0f;
    }
    if (tid == 0) s_dead = 0;
    // cell of this thread: query (part row0 +) tid >> 3, unit u0 + (tid & 7)
    const int crow = tid >> 3, cu = tid & 7, cunit = u0 + cu;
    const int cuc = cunit < H ? cunit : H - 1;
    float bsum[4], cst[NP];
#pragma unroll
    for (int g = 0; g < 4; ++g) bsum[g] = a.bih[d][g * H + cuc] + a.bhh[d][g * H + cuc];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) cst[pp] = 0.0f;
    const size_t gdir = (size_t)B * H;
    auto rows_of = [&](int pp) { const int left = B - pp * RB; return left < RB ? (left > 0 ? left : 0) : RB; };
    // embedded tokens of the next step, requested a step ahead: RB * E / 256 values per thread and part
    constexpr int NX = (RB * E + 255) / 256;
    float xn[NP][NX];
    auto xload = [&](int step, auto pc) {
        constexpr int pp = decltype(pc)::value;
        const int t = d ? a.T - 1 - step : step, nr = rows_of(pp), lim = nr > 0 ? nr * E : 1;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            int i = tid + 256 * j;
            i = i < lim ? i : lim - 1;
            const int xr = i / E, k = i - xr * E;
            xn[pp][j] = a.X[((size_t)(pp * RB + xr < B ? pp * RB + xr : B - 1) * a.T + t) * E + k];
        }
    };
    bool dead = false;
    // every wave sweeps a quarter of the part's rows x H granules of direction dd written at step tag - 1 (tag = that step + 1)
    // into the h part of the LDS rows, 16 loads in flight per lane, a batch re-read until all its tags match
    auto sweep = [&](int dd, int tag, int row0, int nr) {
        seq_gu64 *gp = (seq_gu64 *)(a.hg + ((size_t)((tag - 1) & 1) * 2 + dd) * gdir + (size_t)row0 * H);
        const int n = nr * H, per = (n + 3) / 4, g0 = wv * per, g1 = g0 + per < n ? g0 + per : n;
        for (int base = g0; base < g1 && !dead; base += 64 * NGL) {
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
                unsigned long long gv[NGL];
#pragma unroll
                for (int j = 0; j < NGL; ++j) {
                    const int g = base + lane + 64 * j;
                    gv[j] = __hip_atomic_load(gp + (g < g1 ? g : g1 - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < NGL; ++j) {
                    int g = base + lane + 64 * j;
                    g = g < g1 ? g : g1 - 1;
                    ok &= (unsigned)(gv[j] >> 32) == (unsigned)tag;
                    const int gr = g / H;
                    xh[gr * KX + E + (g - gr * H)] = __uint_as_float((unsigned)gv[j]);
                }
                if (__all(ok)) break;
                if (spins >= SEQ_MAX_SPINS) { dead = true; s_dead = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
    };
    auto part_step = [&](int step, auto pc) {
        constexpr int pp = decltype(pc)::value;
        const int row0 = pp * RB, nr = rows_of(pp);
        if (nr <= 0) return;                                       // (uniform over the workgroup)
        {
            const int lim = nr * E;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                int i = tid + 256 * j;
                i = i < lim ? i : lim - 1;
                const int xr = i / E, k = i - xr * E;
                xh[xr * KX + k] = xn[pp][j];
            }
        }
        xload(step + 1 < a.T ? step + 1 : step, pc);
        __syncthreads();                                          // x_t staged (and the previous cells are done with `pre`)
        const int arow = (rt * 16 + l15) < nr ? rt * 16 + l15 : nr - 1;     // A rows past the part repeat its last query
        const float *ap = xh + arow * KX + lq;
        f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
        // k-steps [S0, S1) of the chain: the A fragments of the next group of 8 k-steps are requested (LDS) before the current
        // group's MFMAs are issued
        auto chain = [&](auto s0c, auto s1c) {
            constexpr int S0 = decltype(s0c)::value, S1 = decltype(s1c)::value, GK = 8, NGRP = (S1 - S0 + GK - 1) / GK;
            float ab[2][GK];
#pragma unroll
            for (int j = 0; j < GK; ++j) ab[0][j] = ap[4 * (S0 + j < S1 ? S0 + j : S1 - 1)];
#pragma unroll
            for (int g = 0; g < NGRP; ++g) {
                if (g + 1 < NGRP) {
#pragma unroll
                    for (int j = 0; j < GK; ++j) { const int sn = S0 + (g + 1) * GK + j; ab[(g + 1) & 1][j] = ap[4 * (sn < S1 ? sn : S1 - 1)]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < GK; ++j) {
                    const int sc = S0 + g * GK + j;
                    if (sc < S1) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[g & 1][j], wreg[sc], acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (comp) chain(std::integral_constant<int, 0>{}, std::integral_constant<int, NCE>{});
        if (step > 0 && !dead) sweep(d, step, row0, nr);
        __syncthreads();                                          // h of the previous step staged by all four waves
        dead = dead || s_dead != 0;
        if (comp && step > 0) chain(std::integral_constant<int, NCE>{}, std::integral_constant<int, NCH>{});
        if (comp) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pre[(rt * 16 + 4 * lq + r) * 32 + ct * 16 + l15] = acc[r];
        }
        __syncthreads();
        if (crow < nr) {
            const float *pr = pre + crow * 32 + cu;
            const float ig = c_sigmoidf(pr[0] + bsum[0]);
            const float fg = c_sigmoidf(pr[8] + bsum[1]);
            const float gg = c_tanhf(pr[16] + bsum[2]);
            const float og = c_sigmoidf(pr[24] + bsum[3]);
            cst[pp] = __builtin_fmaf(fg, cst[pp], ig * gg);
            float hn = og * c_tanhf(cst[pp]);
            if (dead) hn = __uint_as_float(0x7fc00000u);
            const int grow = row0 + crow;
            if (cunit < H && !(step == 1 && (int)blockIdx.x == a.fault_block)) {
                if (step + 1 < a.T || a.Wfc) {
                    seq_gu64 *gq = (seq_gu64 *)(a.hg + ((size_t)(step & 1) * 2 + d) * gdir + (size_t)grow * H + cunit);
                    __hip_atomic_store(gq, ((unsigned long long)(unsigned)(step + 1) << 32) | __float_as_uint(hn),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (step + 1 == a.T) a.hout[(size_t)grow * 2 * H + (size_t)d * H + cunit] = hn;
            }
        }
    };
    xload(0, std::integral_constant<int, 0>{});
    if constexpr (NP > 1) xload(0, std::integral_constant<int, 1>{});
    for (int step = 0; step < a.T; ++step) {
        part_step(step, std::integral_constant<int, 0>{});
        if constexpr (NP > 1) part_step(step, std::integral_constant<int, 1>{});
    }
    static_assert(NP <= 2, "parts written out for two");
    // ---- lang_fc behind the last step: workgroup o < D gathers the final h of every query, one direction at a time (forward
    // first: the chain runs over [h_fwd | h_bwd], k ascending), into the LDS rows' h part; lane r of wave 0 runs query r's
    // chain for output o against the row of Wfc staged in LDS, then + bias (gemm_nt's order) ----
    if (a.Wfc && (int)blockIdx.x < a.D) {
        const int o = blockIdx.x;
        for (int i = tid; i < 2 * H / 4; i += 256)
            *reinterpret_cast<float4 *>(wrow + 4 * i) = *reinterpret_cast<const float4 *>(a.Wfc + (size_t)o * 2 * H + 4 * i);
        for (int pp = 0; pp < NP; ++pp) {
            const int row0 = pp * RB, nr = rows_of(pp);
            if (nr <= 0) break;
            float sfc = 0.0f;
            for (int dd = 0; dd < 2; ++dd) {
                __syncthreads();                                  // (the rows' h part is free: the last chains / the previous gather are done)
                if (!dead) sweep(dd, a.T, row0, nr);
                __syncthreads();
                dead = dead || s_dead != 0;
                if (tid < nr) {
                    const float *hp = xh + tid * KX + E, *wp = wrow + dd * H;
#pragma unroll 8
                    for (int k4 = 0; k4 < H / 4; ++k4) {
                        const float4 w = *reinterpret_cast<const float4 *>(wp + 4 * k4);
                        const float4 x = *reinterpret_cast<const float4 *>(hp + 4 * k4);
                        sfc = __builtin_fmaf(x.x, w.x, sfc); sfc = __builtin_fmaf(x.y, w.y, sfc);
                        sfc = __builtin_fmaf(x.z, w.z, sfc); sfc = __builtin_fmaf(x.w, w.w, sfc);
                    }
                }
            }
            if (tid < nr) a.out[(size_t)(row0 + tid) * a.D + o] = dead ? __uint_as_float(0x7fc00000u) : sfc + a.bfc[o];
        }
    }
    if (dead && tid == 0) atomicOr(a.err, 1u);
}

// ---- rescue of a sequence kernel that gave up ----------------------------------------------------------------------------
// Enqueued behind EVERY launch of the two single-launch sequence kernels; returns at once when the error word is clear (the
// normal case: one empty launch).  When a sweep gave up (some workgroup of the grid was not resident: another stream or
// process held its CU) the outputs are NaN by construction, and this kernel re-encodes the batch with NO cross-workgroup
// dependency: one workgroup per query runs both directions and lang_fc by itself -- thread = hidden unit (H <= 1024), the
// four gate chains of its unit over [x_t | h] (k ascending from zero, the h part skipped at step 0, bias pair after the
// chain: the sequence kernels' order, hence the oracle's bits), h handed from step to step through LDS.  Slow (every
// workgroup streams all of [W_ih | W_hh] per step: milliseconds) and rare; it makes the call's result right instead of NaN,
// raises bit 1 of the error word and bit 0 of the host fault word (vfr_set_fault_word) so that the caller hears about it.
__global__ __launch_bounds__(1024) void lstm_seq_rescue_kernel(SeqLstm a, unsigned *fault_word_host)
{
    if (*a.err == 0u) return;
    extern __shared__ __attribute__((aligned(16))) float seq_lds[];
    const int E = a.E, H = a.H, T = a.T, K2 = 2 * H;
    float *xh = seq_lds;                  // [E + H]: this step's embedded token | the previous step's h
    float *hfin = xh + E + H;             // [2H]: final [h_fwd | h_bwd]
    const int b = blockIdx.x, u = threadIdx.x;
    const bool on = u < H;
    const int uc = on ? u : H - 1;
    for (int d = 0; d < 2; ++d) {
        float bsum[4], cst = 0.0f, hn = 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) bsum[g] = a.bih[d][g * H + uc] + a.bhh[d][g * H + uc];
        for (int step = 0; step < T; ++step) {
            const int t = d ? T - 1 - step : step;
            for (int k = u; k < E; k += 1024) xh[k] = a.X[((size_t)b * T + t) * E + k];
            __syncthreads();                                      // x_t and the previous step's h are in place
            if (on) {
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                for (int k4 = 0; k4 < E / 4; ++k4) {
                    const float4 x = *reinterpret_cast<const float4 *>(xh + 4 * k4);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 w = *reinterpret_cast<const float4 *>(a.Wih[d] + ((size_t)g * H + u) * E + 4 * k4);
                        acc[g] = __builtin_fmaf(x.x, w.x, acc[g]); acc[g] = __builtin_fmaf(x.y, w.y, acc[g]);
                        acc[g] = __builtin_fmaf(x.z, w.z, acc[g]); acc[g] = __builtin_fmaf(x.w, w.w, acc[g]);
                    }
                }
                if (step > 0)
                    for (int k4 = 0; k4 < H / 4; ++k4) {
                        const float4 x = *reinterpret_cast<const float4 *>(xh + E + 4 * k4);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 w = *reinterpret_cast<const float4 *>(a.Whh[d] + ((size_t)g * H + u) * H + 4 * k4);
                            acc[g] = __builtin_fmaf(x.x, w.x, acc[g]); acc[g] = __builtin_fmaf(x.y, w.y, acc[g]);
                            acc[g] = __builtin_fmaf(x.z, w.z, acc[g]); acc[g] = __builtin_fmaf(x.w, w.w, acc[g]);
                        }
                    }
                const float ig = c_sigmoidf(acc[0] + bsum[0]);
                const float fg = c_sigmoidf(acc[1] + bsum[1]);
                const float gg = c_tanhf(acc[2] + bsum[2]);
                const float og = c_sigmoidf(acc[3] + bsum[3]);
                cst = __builtin_fmaf(fg, cst, ig * gg);
                hn = og * c_tanhf(cst);
            }
            __syncthreads();                                      // every chain has read the old h
            if (on) xh[E + u] = hn;
        }
        if (on) {
            hfin[d * H + u] = hn;
            a.hout[(size_t)b * K2 + (size_t)d * H + u] = hn;
        }
        __syncthreads();
    }
    if (a.Wfc)
        for (int o = u; o < a.D; o += 1024) {
            const float *wr = a.Wfc + (size_t)o * K2;
            float s = 0.0f;
            for (int k4 = 0; k4 < K2 / 4; ++k4) {
                const float4 w = *reinterpret_cast<const float4 *>(wr + 4 * k4);
                const float4 x = *reinterpret_cast<const float4 *>(hfin + 4 * k4);
                s = __builtin_fmaf(x.x, w.x, s); s = __builtin_fmaf(x.y, w.y, s);
                s = __builtin_fmaf(x.z, w.z, s); s = __builtin_fmaf(x.w, w.w, s);
            }
            a.out[(size_t)b * a.D + o] = s + a.bfc[o];
        }
    if (b == 0 && u == 0) {
        // (the word only ever GAINS bits: a workgroup that starts after this one still sees it raised)
        atomicOr(a.err, 2u);
        if (fault_word_host) __hip_atomic_fetch_or(fault_word_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// ---- rescue of the multi-step kernel (gemm.hip: lstm_steps_mfma_kernel) ------------------------------------------------------
// Same contract as above for the big-batch form: one empty launch normally; when the error word is raised the final state of
// EVERY GEMM row is recomputed with no cross-workgroup dependency -- one workgroup per row, thread = hidden unit, the four
// gate chains of its unit started from the projection table (column of (gate, unit) in the fused step's tile order) and
// continued over h, k ascending, bias pair after the chain: the fused step's order, hence the oracle's bits.  A row of the
// reverse direction that the fused path lets ride on the all-pad row is simply stepped through its own pad tokens here (the
// same tokens, the same state).  Slow (every workgroup streams W_hh per step) and rare.
struct StepsRescue {
    const float *ptab; size_t pdir; int NP;          // projection tables [2][vocab, NP]
    const int *tokidx;                                // [T][R]
    const float *Whh[2], *bih[2], *bhh[2];
    float *hout;                                      // [R, 2H] final state (direction d at column d*H)
    unsigned *err;
    int R, T, H;
};
__global__ __launch_bounds__(1024) void lstm_steps_rescue_kernel(StepsRescue a, unsigned *fault_word_host)
{
    if (*a.err == 0u) return;
    extern __shared__ __attribute__((aligned(16))) float seq_lds[];
    float *hs = seq_lds;                              // [H] the previous step's h
    const int m = blockIdx.x, u = threadIdx.x, H = a.H;
    const bool on = u < H;
    const int uc = on ? u : H - 1;
    const int pcol = (uc >> 5) * 128 + ((uc >> 4) & 1) * 64 + (uc & 15);      // + 16 * gate
    for (int d = 0; d < 2; ++d) {
        float bsum[4], cst = 0.0f, hn = 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) bsum[g] = a.bih[d][g * H + uc] + a.bhh[d][g * H + uc];
        for (int step = 0; step < a.T; ++step) {
            const int t = d ? a.T - 1 - step : step;
            const float *prow = a.ptab + (size_t)d * a.pdir + (size_t)a.tokidx[(size_t)t * a.R + m] * a.NP + pcol;
            float acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = prow[16 * g];
            if (on && step > 0)
                for (int k4 = 0; k4 < H / 4; ++k4) {
                    const float4 x = *reinterpret_cast<const float4 *>(hs + 4 * k4);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 w = *reinterpret_cast<const float4 *>(a.Whh[d] + ((size_t)g * H + u) * H + 4 * k4);
                        acc[g] = __builtin_fmaf(x.x, w.x, acc[g]); acc[g] = __builtin_fmaf(x.y, w.y, acc[g]);
                        acc[g] = __builtin_fmaf(x.z, w.z, acc[g]); acc[g] = __builtin_fmaf(x.w, w.w, acc[g]);
                    }
                }
            const float ig = c_sigmoidf(acc[0] + bsum[0]);
            const float fg = c_sigmoidf(acc[1] + bsum[1]);
            const float gg = c_tanhf(acc[2] + bsum[2]);
            const float og = c_sigmoidf(acc[3] + bsum[3]);
            cst = __builtin_fmaf(fg, cst, ig * gg);
            hn = og * c_tanhf(cst);
            __syncthreads();                                      // every chain has read the old h
            if (on) hs[u] = hn;
            __syncthreads();
        }
        if (on) a.hout[(size_t)m * 2 * H + (size_t)d * H + u] = hn;
        __syncthreads();
    }
    if (m == 0 && u == 0) {
        atomicOr(a.err, 2u);
        if (fault_word_host) __hip_atomic_fetch_or(fault_word_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static int launch_seq_rescue(const SeqLstm &a, hipStream_t st)
{
    const size_t lds = ((size_t)a.E + 3 * (size_t)a.H) * sizeof(float);
    hipLaunchKernelGGL(lstm_seq_rescue_kernel, dim3((unsigned)a.B), dim3(1024), lds, st, a, fault_word());
    return hipGetLastError() == hipSuccess ? VFR_OK : fail(VFR_EHIP, "lstm_seq_rescue_kernel: launch failed");
}

// the sequence kernels declare up to 160 KB of dynamic LDS: asked for once per kernel AND DEVICE; a device / runtime that
// refuses keeps the per-step paths
template <typename K>
static bool seq_lds_admitted(K kernel)
{
    static std::atomic<signed char> state[VFR_MAX_DEVICES];           // 0: not asked yet, 1: admitted, -1: refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VFR_MAX_DEVICES) { (void)hipGetLastError(); return false; }
    signed char v = state[dev].load();
    if (v == 0) {
        v = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 1 : -1;
        if (v < 0) (void)hipGetLastError();
        state[dev].store(v);
    }
    return v == 1;
}

struct LstmWs {
    float *X, *gates, *c, *c2, *hcat, *hcat2, *hfinal, *xv, *wperm, *ptab, *wt;
    unsigned long long *hg;               // persistent sequence kernel: [16 B error word | 2 x 2 x B x H granules]
    int *tokidx;
    unsigned *sync;                       // multi-step kernel: tickets, error word, completion counters (lstm_steps_sync_words)
    int64_t *tok_ext;
    int *len, *row_of, *xrow, *mcount, *hist;
    size_t total;
};
static LstmWs carve(void *base, int64_t B, int T, int E, int H, int vocab)
{
    LstmWs w{};
    size_t off = 0;
    auto take_b = [&](size_t bytes) { char *p = static_cast<char *>(base) + off; off += align_up(bytes, 256); return p; };
    auto take = [&](size_t n) { return reinterpret_cast<float *>(take_b(n * sizeof(float))); };
    const size_t R = (size_t)B + 1;                       // + the virtual all-pad query
    w.X = take(R * T * E);
    w.gates = take((size_t)2 * R * 4 * H);
    w.c = take((size_t)2 * R * H);
    w.c2 = take((size_t)2 * R * H);
    w.hcat = take(R * 2 * H);
    w.hcat2 = take(R * 2 * H);
    w.hfinal = take((size_t)B * 2 * H);
    w.tok_ext = reinterpret_cast<int64_t *>(take_b(R * T * sizeof(int64_t)));
    w.len = reinterpret_cast<int *>(take_b(R * sizeof(int)));
    w.row_of = reinterpret_cast<int *>(take_b(R * sizeof(int)));
    w.xrow = reinterpret_cast<int *>(take_b(R * sizeof(int)));
    w.mcount = reinterpret_cast<int *>(take_b((size_t)(T + 1) * sizeof(int)));
    w.hist = reinterpret_cast<int *>(take_b((size_t)cdiv(R, SORT_BLOCK) * (T + 1) * sizeof(int)));
    if (use_vocab_table(B, T, vocab)) {
        const size_t np = (size_t)cdiv(H, 32) * 128;
        w.xv = take((size_t)vocab * E);
        w.wperm = take(2 * np * E);
        w.ptab = take(2 * (size_t)vocab * np);
        w.tokidx = reinterpret_cast<int *>(take_b(R * T * sizeof(int)));
        w.sync = reinterpret_cast<unsigned *>(take_b(lstm_steps_sync_words((int64_t)R, T) * sizeof(unsigned)));
    }
    if (B <= 4) w.wt = take((size_t)2 * (E + H) * 4 * H);       // k-major weights of the vector-chain step (a few queries)
    if (B <= 64) w.hg = reinterpret_cast<unsigned long long *>(take_b(16 + (size_t)4 * B * H * sizeof(unsigned long long)));
    w.total = off;
    return w;
}

}  // namespace vfr

extern "C" {

size_t vfr_bilstm_workspace_bytes(int64_t B, int T, int E, int H, int vocab)
{
    if (B < 0 || T < 0 || E < 0 || H < 0) return 0;
    return vfr::carve(nullptr, B, T, E, H, vocab).total;
}

int vfr_bilstm_final_f32(const int64_t *tokens, int64_t B, int T, const float *emb, int vocab, const float *len_tab,
                         const float *Wih_f, const float *Whh_f, const float *bih_f, const float *bhh_f,
                         const float *Wih_b, const float *Whh_b, const float *bih_b, const float *bhh_b, int E,
                         int H, const float *Wfc, const float *bfc, int D, float *out, void *workspace,
                         size_t workspace_bytes, vfr_stream_t stream)
{
    VFR_REQUIRE(tokens && emb && Wih_f && Whh_f && bih_f && bhh_f && Wih_b && Whh_b && bih_b && bhh_b && Wfc && bfc &&
                    out && B >= 0 && T > 0 && E > 0 && H > 0 && D > 0 && vocab > 0,
                VFR_EINVAL, "vfr_bilstm_final_f32: bad argument");
    if (B == 0) return VFR_OK;
    VFR_REQUIRE(workspace && workspace_bytes >= vfr_bilstm_workspace_bytes(B, T, E, H, vocab), VFR_EWORKSPACE,
                "vfr_bilstm_final_f32: workspace %zu < %zu bytes", workspace_bytes,
                vfr_bilstm_workspace_bytes(B, T, E, H, vocab));
    hipStream_t st = vfr::as_stream(stream);
    vfr::LstmWs w = vfr::carve(workspace, B, T, E, H, vocab);
    const float *Wih[2] = {Wih_f, Wih_b}, *Whh[2] = {Whh_f, Whh_b};
    const float *bih[2] = {bih_f, bih_b}, *bhh[2] = {bhh_f, bhh_b};
    const int G = 4 * H;

    VFR_REQUIRE(T <= 1024, VFR_EUNSUPPORTED, "vfr_bilstm_final_f32: T=%d > 1024", T);
    if (vfr::opt_lstm_persist() && B > vfr::opt_lstm_persist_min() && B <= vfr::opt_lstm_persist_max() && B <= 64 && E == 100 && H == 1000 && w.hg &&
        2 * (int)vfr::cdiv(H, 8) <= vfr::device_cu_count() &&
        (B <= 16 ? vfr::seq_lds_admitted(vfr::lstm_seq_mfma_kernel<1, 1, 25, 275>)
                 : B <= 32 ? vfr::seq_lds_admitted(vfr::lstm_seq_mfma_kernel<2, 1, 25, 275>) : vfr::seq_lds_admitted(vfr::lstm_seq_mfma_kernel<2, 2, 25, 275>))) {
        // 3 .. 64 queries at the model's shape: the whole sequence in one launch on the matrix pipe, weights in registers
        // (lstm_seq_mfma_kernel); every query steps through all T tokens (no sorting, no pad row)
        const size_t gbytes = 16 + (size_t)4 * B * H * sizeof(unsigned long long);
        if (int rc = vfr::fill_region(w.hg, 0u, gbytes, st)) return rc;
        {
        vfr::ProfScope prof(vfr::SITE_EMBED, st);
        vfr::launch_embed(tokens, B * T, vocab, emb, len_tab, E, w.X, st);
        }
        VFR_CHECK_LAUNCH("embed_kernel");
        const dim3 grid(2 * (unsigned)vfr::cdiv(H, 8));
        const bool fc_in = D <= (int)grid.x && (((uintptr_t)Wfc) & 15) == 0;        // lang_fc rides in the same launch
        vfr::SeqLstm a{w.X, {Wih_f, Wih_b}, {Whh_f, Whh_b}, {bih_f, bih_b}, {bhh_f, bhh_b}, w.hg + 2, w.hcat,
                       reinterpret_cast<unsigned *>(w.hg), (int)B, T, E, H, fc_in ? Wfc : nullptr, bfc, out, D, vfr::opt_lstm_persist_fault()};
        {
        vfr::ProfScope prof(vfr::SITE_GEMM_LSTM_REC, st);
        const int rtiles = B <= 16 ? 1 : 2;
        const size_t lds = ((size_t)16 * rtiles * (E + H + 32) + 4 + 2 * H) * sizeof(float);
        if (rtiles == 1)  hipLaunchKernelGGL((vfr::lstm_seq_mfma_kernel<1, 1, 25, 275>), grid, dim3(256), lds, st, a);
        else if (B <= 32) hipLaunchKernelGGL((vfr::lstm_seq_mfma_kernel<2, 1, 25, 275>), grid, dim3(256), lds, st, a);
        else              hipLaunchKernelGGL((vfr::lstm_seq_mfma_kernel<2, 2, 25, 275>), grid, dim3(256), lds, st, a);
        }
        VFR_CHECK_LAUNCH("lstm_seq_mfma_kernel");
        if (int rc = vfr::launch_seq_rescue(a, st)) return rc;         // no-op unless a sweep gave up: then the batch is re-encoded (never NaN)
        if (fc_in) return VFR_OK;
        vfr::GemmArgs g{};
        g.A = w.hcat; g.lda = 2 * H; g.W = Wfc; g.ldw = 2 * H; g.out = out; g.ldo = D; g.M = B; g.N = D; g.K = 2 * H;
        g.bias = bfc; g.epi = vfr::EPI_BIAS; g.site = vfr::SITE_GEMM_LANG_FC;
        return vfr::gemm_nt(g, st);
    }
    if (B <= vfr::opt_lstm_small() && B <= 4 && (E % 4) == 0 && (H % 4) == 0 && w.wt && (size_t)4 * (E + H) * 4 <= 48 * 1024) {
        // a handful of queries: every row steps through all T tokens with the vector-chain step (no sorting, no pad row)
        {
            const vfr::FillJob jobs[2] = {{w.c, nullptr, (size_t)2 * B * H * sizeof(float), 0u}, {w.hcat, nullptr, (size_t)B * 2 * H * sizeof(float), 0u}};
            if (int rc = vfr::fill_regions(jobs, 2, st)) return rc;
        }
        {
        vfr::ProfScope prof(vfr::SITE_EMBED, st);
        vfr::launch_embed(tokens, B * T, vocab, emb, len_tab, E, w.X, st);
        }
        VFR_CHECK_LAUNCH("embed_kernel");
        // one or two queries at a shape whose 32-column weight slices fit a CU's LDS: the whole sequence in one launch
        const size_t seq_lds = ((size_t)((E + H) / 4) * vfr::SEQ_WSTRIDE + (size_t)(B <= 1 ? 1 : 2) * (H + E + 32)) * sizeof(float);
        const int seq_grid = 2 * (int)vfr::cdiv(H, 8);
        if (vfr::opt_lstm_persist() && B <= 2 && w.hg && H <= 1024 && 2 * E <= 256 && (((uintptr_t)Wfc) & 15) == 0 && seq_lds <= 160 * 1024 && seq_grid <= vfr::device_cu_count() &&
            (B <= 1 ? vfr::seq_lds_admitted(vfr::lstm_seq_small_kernel<1>) : vfr::seq_lds_admitted(vfr::lstm_seq_small_kernel<2>))) {
            const size_t gbytes = 16 + (size_t)4 * B * H * sizeof(unsigned long long);
            if (int rc = vfr::fill_region(w.hg, 0u, gbytes, st)) return rc;
            // lang_fc rides in the same launch when every output gets a workgroup and the row fits beside the gathered h
            const bool fc_in = D <= seq_grid && (size_t)(1 + (B <= 1 ? 1 : 2)) * 2 * H * sizeof(float) <= seq_lds;
            vfr::SeqLstm a{w.X, {Wih_f, Wih_b}, {Whh_f, Whh_b}, {bih_f, bih_b}, {bhh_f, bhh_b}, w.hg + 2, w.hcat,
                           reinterpret_cast<unsigned *>(w.hg), (int)B, T, E, H, fc_in ? Wfc : nullptr, bfc, out, D, vfr::opt_lstm_persist_fault()};
            {
            vfr::ProfScope prof(vfr::SITE_GEMM_LSTM_REC, st);
            if (B <= 1) hipLaunchKernelGGL(vfr::lstm_seq_small_kernel<1>, dim3((unsigned)seq_grid), dim3(256), seq_lds, st, a);
            else        hipLaunchKernelGGL(vfr::lstm_seq_small_kernel<2>, dim3((unsigned)seq_grid), dim3(256), seq_lds, st, a);
            }
            VFR_CHECK_LAUNCH("lstm_seq_small_kernel");
            if (int rc = vfr::launch_seq_rescue(a, st)) return rc;
            if (fc_in) return VFR_OK;
            vfr::GemmArgs g{};
            g.A = w.hcat; g.lda = 2 * H; g.W = Wfc; g.ldw = 2 * H; g.out = out; g.ldo = D; g.M = B; g.N = D; g.K = 2 * H;
            g.bias = bfc; g.epi = vfr::EPI_BIAS; g.site = vfr::SITE_GEMM_LANG_FC;
            return vfr::gemm_nt(g, st);
        }
        float *wt[2] = {w.wt, w.wt + (size_t)(E + H) * 4 * H};
        {
        vfr::ProfScope prof(vfr::SITE_GEMM_LSTM_IN, st);
        for (int d = 0; d < 2; ++d) {                                   // [4H, E] and [4H, H] -> chunk-major [(E + H) / 4][4H][4]
            const int64_t n1 = (int64_t)(E / 4) * 4 * H, n2 = (int64_t)(H / 4) * 4 * H;
            hipLaunchKernelGGL(vfr::pack_k4_kernel, dim3((unsigned)vfr::cdiv(n1, 256)), dim3(256), 0, st, d ? Wih_b : Wih_f, 4 * (int64_t)H, E, wt[d]);
            hipLaunchKernelGGL(vfr::pack_k4_kernel, dim3((unsigned)vfr::cdiv(n2, 256)), dim3(256), 0, st, d ? Whh_b : Whh_f, 4 * (int64_t)H, H,
                               wt[d] + (size_t)E * 4 * H);
        }
        }
        VFR_CHECK_LAUNCH("pack_k4_kernel");
        float *hin = w.hcat, *hout = w.hcat2, *cin = w.c, *cout = w.c2;
        for (int step = 0; step < T; ++step) {
            vfr::SmallLstm a{w.X, {wt[0], wt[1]}, {bih_f, bih_b}, {bhh_f, bhh_b}, hin, cin, hout, cout,
                             (int)B, T, E, H, step, (step == 0 && vfr::opt_lstm_skip0()) ? 0 : 1};
            vfr::ProfScope prof(vfr::SITE_GEMM_LSTM_REC, st);
            const dim3 grid((unsigned)vfr::cdiv(H, 16), 2);
            const int nch = (a.recurrent ? E + H : E) / 4;
            if (B == 1 && nch == 275 && vfr::opt_lstm_small4()) {       // the model's shape: four-wave weight stream
                hipLaunchKernelGGL((vfr::lstm_step_small4_kernel<1, 275>), grid, dim3(64 * (vfr::S4_NL + 1)), (size_t)(E + H) * 4, st, a);
                float *tmp4 = hin; hin = hout; hout = tmp4;
                tmp4 = cin; cin = cout; cout = tmp4;
                continue;
            }
            switch ((int)B) {
            case 1: hipLaunchKernelGGL(vfr::lstm_step_small_kernel<1>, grid, dim3(64), (size_t)1 * (E + H) * 4, st, a); break;
            case 2: hipLaunchKernelGGL(vfr::lstm_step_small_kernel<2>, grid, dim3(64), (size_t)2 * (E + H) * 4, st, a); break;
            default: hipLaunchKernelGGL(vfr::lstm_step_small_kernel<4>, grid, dim3(64), (size_t)4 * (E + H) * 4, st, a); break;
            }
            float *tmp = hin; hin = hout; hout = tmp;
            tmp = cin; cin = cout; cout = tmp;
        }
        VFR_CHECK_LAUNCH("lstm_step_small_kernel");
        vfr::GemmArgs g{};
        g.A = hin; g.lda = 2 * H; g.W = Wfc; g.ldw = 2 * H; g.out = out; g.ldo = D; g.M = B; g.N = D; g.K = 2 * H;
        g.bias = bfc; g.epi = vfr::EPI_BIAS; g.site = vfr::SITE_GEMM_LANG_FC;
        return vfr::gemm_nt(g, st);
    }
    const int64_t R = B + 1;                                // GEMM rows: row 0 = all-pad query, then queries by length
    // tokens + one all-pad query -> embeddings of R queries
    {
        const vfr::FillJob jobs[5] = {{w.tok_ext, tokens, (size_t)B * T * sizeof(int64_t), 0u},
                                      {w.tok_ext + (size_t)B * T, nullptr, (size_t)T * sizeof(int64_t), 0u},
                                      {w.c, nullptr, (size_t)2 * R * H * sizeof(float), 0u},
                                      {w.hcat, nullptr, (size_t)R * 2 * H * sizeof(float), 0u},
                                      {w.sync, nullptr, w.sync ? vfr::lstm_steps_sync_words(R, T) * sizeof(unsigned) : 0, 0u}};
        if (int rc = vfr::fill_regions(jobs, w.sync ? 5 : 4, st)) return rc;        // one launch instead of a copy and three memsets
    }
    const bool fused = vfr::opt_gemm() != 0 && (E % 4) == 0 && (H % 4) == 0 &&
                       ((((uintptr_t)Wih_f) | ((uintptr_t)Whh_f) | ((uintptr_t)Wih_b) | ((uintptr_t)Whh_b)) & 15) == 0;
    const bool table = fused && vfr::use_vocab_table(B, T, vocab);
    {
    vfr::ProfScope prof(vfr::SITE_EMBED, st);
    if (table)          // one embedded row per VOCABULARY entry (the steps start their chains from the projection table)
        vfr::launch_embed(nullptr, (int64_t)vocab, vocab, emb, len_tab, E, w.xv, st);
    else
        vfr::launch_embed(w.tok_ext, R * T, vocab, emb, len_tab, E, w.X, st);
    hipLaunchKernelGGL(vfr::query_length_kernel, dim3((unsigned)vfr::cdiv(B, 256)), dim3(256), 0, st, tokens, B, T, w.len);
    hipLaunchKernelGGL(vfr::length_hist_kernel, dim3((unsigned)vfr::cdiv(B, vfr::SORT_BLOCK)), dim3(vfr::SORT_BLOCK), 0, st,
                       w.len, B, T, w.hist);
    hipLaunchKernelGGL(vfr::sort_rows_kernel, dim3((unsigned)vfr::cdiv(B, vfr::SORT_BLOCK)), dim3(vfr::SORT_BLOCK), 0, st,
                       w.len, w.hist, B, T, w.row_of, w.xrow);
    hipLaunchKernelGGL(vfr::active_rows_kernel, dim3(1), dim3(T < 64 ? 64 : T), 0, st, w.hist, (int)vfr::cdiv(B, vfr::SORT_BLOCK), T,
                       w.mcount);
    }
    VFR_CHECK_LAUNCH("bilstm row bookkeeping");
    float *h_sorted = w.hcat;
    const int64_t NP = vfr::cdiv(H, 32) * 128;              // tile-column count of the fused step (4 gates x 32 units per tile)
    if (table) {
        // P_d [vocab, NP] = emb_rows x W_ih,d^T (rows of W_ih permuted into the step's tile-column order): the first E
        // terms of every gate chain, once per vocabulary entry instead of once per (query, time)
        vfr::GemmArgs gp[2]{};
        for (int d = 0; d < 2; ++d) {
            if (int rc = vfr::lstm_permute_rows(Wih[d], H, E, w.wperm + (size_t)d * NP * E, st)) return rc;
            gp[d].A = w.xv; gp[d].lda = E; gp[d].W = w.wperm + (size_t)d * NP * E; gp[d].ldw = E;
            gp[d].out = w.ptab + (size_t)d * vocab * NP; gp[d].ldo = NP; gp[d].M = vocab; gp[d].N = (int)NP; gp[d].K = E;
            gp[d].site = vfr::SITE_GEMM_LSTM_IN;
        }
        if (int rc = vfr::gemm_nt_pair(gp[0], gp[1], st)) return rc;
        hipLaunchKernelGGL(vfr::token_index_kernel, dim3((unsigned)vfr::cdiv(R * T, 256)), dim3(256), 0, st, w.tok_ext, w.xrow, R, T,
                           vocab, w.tokidx);
        VFR_CHECK_LAUNCH("token_index_kernel");
    }
    if (fused && table && w.sync && H <= 1024 && vfr::lstm_steps_supported(R, H)) {
        // ALL T steps of both directions in one launch (gemm.hip: lstm_steps_mfma_kernel): no partial round of workgroups at
        // the end of every step, no launch gap between steps.  Three rotating state buffers (the third pair lives in the
        // generic path's gates array, unused here); the final state is in buffer T % 3.
        float *hbuf[3] = {w.hcat, w.hcat2, w.gates + (size_t)2 * R * H}, *cbuf[3] = {w.c, w.c2, w.gates};
        vfr::GemmArgs g[2]{};
        for (int d = 0; d < 2; ++d) {
            g[d].A2 = hbuf[0] + (size_t)d * H; g[d].lda2 = 2 * H; g[d].W2 = Whh[d]; g[d].ldw2 = H; g[d].K2 = H;
            g[d].bias = bih[d]; g[d].bias2 = bhh[d]; g[d].M = R; g[d].N = G;
            g[d].lstm_ldh = 2 * H; g[d].lstm_H = H; g[d].site = vfr::SITE_GEMM_LSTM_REC;
            g[d].lstm_xrow = nullptr; g[d].lstm_mcount = d ? w.mcount : nullptr;
            g[d].Cin = w.ptab + (size_t)d * vocab * NP; g[d].ldc = NP;
        }
        if (int rc = vfr::lstm_steps_run(g[0], g[1], hbuf, cbuf, w.tokidx, w.mcount, w.sync, T, st)) return rc;
        h_sorted = hbuf[T % 3];
        // no-op unless the launch gave up (a bug or a lost workgroup): then every row is re-encoded on its own (never NaN)
        vfr::StepsRescue ra{w.ptab, (size_t)vocab * NP, (int)NP, w.tokidx, {Whh[0], Whh[1]}, {bih[0], bih[1]}, {bhh[0], bhh[1]}, h_sorted, w.sync + 8, (int)R, T, H};
        hipLaunchKernelGGL(vfr::lstm_steps_rescue_kernel, dim3((unsigned)R), dim3(1024), (size_t)H * sizeof(float), st, ra, vfr::fault_word());
        VFR_CHECK_LAUNCH("lstm_steps_rescue_kernel");
    } else if (fused) {
        // one MFMA launch per time step for both directions: K = [x_t (E) | h (H)], gate epilogue fused; the reverse
        // direction only touches the rows that have reached a real token (lstm_mcount), the rest ride on row 0
        float *hin = w.hcat, *hout = w.hcat2, *cin = w.c, *cout = w.c2;
        for (int step = 0; step < T; ++step) {
            vfr::GemmArgs g[2]{};
            for (int d = 0; d < 2; ++d) {
                const int t = d ? T - 1 - step : step;
                g[d].A = w.X + (size_t)t * E; g[d].lda = (int64_t)T * E; g[d].W = Wih[d]; g[d].ldw = E; g[d].K = E;
                g[d].A2 = hin + (size_t)d * H; g[d].lda2 = 2 * H; g[d].W2 = Whh[d]; g[d].ldw2 = H;
                // h_0 = 0 (models.py:50-52): every recurrent term of the first step is fma(0, w, acc) == acc for finite
                // weights, so the chain is left where the x segment (or the projection table) put it and the step is its
                // gather + gate epilogue only.  (A non-finite W_hh would give NaN in the reference here; not reproduced.)
                g[d].K2 = (step == 0 && vfr::opt_lstm_skip0()) ? 0 : H;
                g[d].bias = bih[d]; g[d].bias2 = bhh[d]; g[d].M = R; g[d].N = G;
                g[d].lstm_c = cout + (size_t)d * R * H; g[d].lstm_cin = cin + (size_t)d * R * H;
                g[d].lstm_h = hout + (size_t)d * H; g[d].lstm_ldh = 2 * H;
                g[d].lstm_H = H; g[d].out = hout; g[d].site = vfr::SITE_GEMM_LSTM_REC;
                g[d].lstm_xrow = w.xrow; g[d].lstm_mcount = d ? w.mcount : nullptr; g[d].lstm_step = step;
                if (table) {
                    g[d].A = g[d].A2; g[d].W = g[d].W2; g[d].lda = g[d].lda2; g[d].ldw = g[d].ldw2; g[d].K = 0;   // no x segment
                    g[d].Cin = w.ptab + (size_t)d * vocab * NP; g[d].ldc = NP;
                    g[d].lstm_tok = w.tokidx + (size_t)t * R;
                }
            }
            if (int rc = vfr::lstm_step_pair(g[0], g[1], st)) return rc;
            // reverse-direction rows that have not joined yet keep (unused) stale values in the ping-pong buffers; they
            // are overwritten from row 0 when they join, so nothing needs copying between the buffers
            float *tmp = hin; hin = hout; hout = tmp;
            tmp = cin; cin = cout; cout = tmp;
        }
        h_sorted = hin;
    } else {
        // generic path (any E/H alignment, or the VALU cross-check build): same canonical order, unfused and without
        // the pad-prefix sharing -- x-part chain into gates, h-part chain continuing from it (C-in), then the pointwise
        // kernel adds the biases.  Rows stay in query order (row B is the unused all-pad query).
        for (int step = 0; step < T; ++step) {
            vfr::GemmArgs gx[2]{}, gh[2]{};
            for (int d = 0; d < 2; ++d) {
                const int t = d ? T - 1 - step : step;
                gx[d].A = w.X + (size_t)t * E; gx[d].lda = (int64_t)T * E; gx[d].W = Wih[d]; gx[d].ldw = E;
                gx[d].out = w.gates + (size_t)d * R * G; gx[d].ldo = G; gx[d].M = R; gx[d].N = G; gx[d].K = E;
                gx[d].site = vfr::SITE_GEMM_LSTM_IN;
                gh[d] = gx[d];
                gh[d].A = w.hcat + (size_t)d * H; gh[d].lda = 2 * H; gh[d].W = Whh[d]; gh[d].ldw = H; gh[d].K = H;
                gh[d].Cin = gx[d].out; gh[d].ldc = G; gh[d].site = vfr::SITE_GEMM_LSTM_REC;
            }
            if (int rc = vfr::gemm_nt_pair(gx[0], gx[1], st)) return rc;
            if (int rc = vfr::gemm_nt_pair(gh[0], gh[1], st)) return rc;
            {
            vfr::ProfScope prof(vfr::SITE_LSTM_POINTWISE, st);
            hipLaunchKernelGGL(vfr::lstm_pointwise_kernel, dim3((unsigned)vfr::cdiv(2 * R * H, 256)), dim3(256), 0, st,
                               w.gates, bih_f, bhh_f, bih_b, bhh_b, w.c, w.hcat, R, H);
            }
            VFR_CHECK_LAUNCH("lstm_pointwise_kernel");
        }
    }
    float *h_final = w.hfinal;
    hipLaunchKernelGGL(vfr::unsort_rows_kernel, dim3((unsigned)vfr::cdiv(B * 2 * H, 256)), dim3(256), 0, st, h_sorted,
                       fused ? w.row_of : nullptr, fused ? w.len : nullptr, B, H, h_final);
    VFR_CHECK_LAUNCH("unsort_rows_kernel");
    vfr::GemmArgs g{};
    g.A = h_final; g.lda = 2 * H; g.W = Wfc; g.ldw = 2 * H; g.out = out; g.ldo = D; g.M = B; g.N = D; g.K = 2 * H;
    g.bias = bfc; g.epi = vfr::EPI_BIAS; g.site = vfr::SITE_GEMM_LANG_FC;
    return vfr::gemm_nt(g, st);
}

}  // extern "C"
