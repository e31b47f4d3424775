// Chain GEMM "NT" for gfx950:  out[m][n] = epi( Cin[m][n] | 0  +  sum_k A[m][k] * W[n][k] ).
//
// Every output element is ONE k-ascending fp32 fma chain (oracle rule R1).  Two kernels compute
// exactly that chain and therefore the same bits:
//   * gemm_nt_valu  -- LDS-tiled v_fma_f32 kernel (64x64 tile, 4x4 per thread); the simple form,
//                      kept as the on-device cross-check (vfr_set_option("gemm", 0));
//   * gemm_nt_mfma  -- v_mfma_f32_16x16x4_f32 kernel (128x128 or 64x128 tile, 4 waves x 4x4 / 2x4 MFMA tiles),
//                      the production path: the MFMA's accumulate order over k IS the chain.
// Used by: visual MLP (model/models.py:21-26), BiLSTM input/recurrent projections and lang_fc
// (model/models.py:40-47), BERT-branch Linear (:31), VGG fc6/fc7 (get_rgb_features.py:126).
#include "vfr_common.h"
#include <type_traits>
#include <vector>
#include <cstdlib>
#include <cstdio>
#include "vfr_math.h"

namespace vfr {

// ------------------------------------------------------------------------------------------------
// VALU chain kernel
// ------------------------------------------------------------------------------------------------
constexpr int VBM = 64, VBN = 64, VBK = 16, VPAD = 4;

__global__ __launch_bounds__(256) void gemm_nt_valu(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) float As[VBK][VBM + VPAD];
    __shared__ __attribute__((aligned(16))) float Ws[VBK][VBN + VPAD];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * VBM;
    const int n0 = blockIdx.y * VBN;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t m = m0 + ty * 4 + i;
            int n = n0 + tx * 4 + j;
            acc[i][j] = (g.Cin && m < g.M && n < g.N) ? g.Cin[m * g.ldc + n] : 0.0f;
        }

    for (int k0 = 0; k0 < g.K; k0 += VBK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int e = tid + 256 * i, r = e >> 4, kk = e & 15;
            int64_t m = m0 + r;
            int n = n0 + r, k = k0 + kk;
            As[kk][r] = (m < g.M && k < g.K) ? g.A[m * g.lda + k] : 0.0f;
            Ws[kk][r] = (n < g.N && k < g.K) ? g.W[(int64_t)n * g.ldw + k] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < VBK; ++kk) {
            float4 a = *reinterpret_cast<const float4 *>(&As[kk][ty * 4]);
            float4 w = *reinterpret_cast<const float4 *>(&Ws[kk][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w}, wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(av[i], wv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t m = m0 + ty * 4 + i;
            int n = n0 + tx * 4 + j;
            if (m >= g.M || n >= g.N) continue;
            float v = acc[i][j];
            if (g.epi & EPI_BIAS2) v = v + (g.bias[n] + g.bias2[n]);
            else if (g.epi & EPI_BIAS) v = v + g.bias[n];
            if (g.epi & EPI_RES) v = v + g.res[m * g.ldr + n];
            if (g.epi & EPI_RELU) v = v > 0.0f ? v : 0.0f;
            g.out[m * g.ldo + n] = v;
        }
}


// ------------------------------------------------------------------------------------------------
// MFMA chain kernel: 128x128 (or 64x128) block tile, BK = 32, 4 waves as 2(M) x 2(N), each wave 4x4 (2x4) tiles of
// v_mfma_f32_16x16x4_f32 (64 / 32 accumulator VGPRs).  One MFMA consumes k .. k+3 in order on top of its accumulator
// (D = fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C))))), and the k-loop below walks k strictly ascending, so every
// output element is the same fp32 chain the VALU kernel and the oracle compute -- identical bits (the GPU tests compare with
// array_equal).
//
// Operand fetch: LDS tiles are row-major [row][k] with a 36-float row stride.  A fragment is ONE dword per lane --
// A[16*t + (lane & 15)][4*k4 + (lane >> 4)], exactly the 16x16x4 operand layout -- read with ds_read_b32
// (bank = 36*r + q mod 64: 64 distinct banks).  The first version of this kernel used 32x32x2 tiles fed by ds_read_b128
// fragments and a lane-half select; tools/ubench/mfma_rates.hip shows why it stalled at 121 TF on a 157 TF pipe: the 1 KB
// LDS returns compete with the 32x32x2 accumulator write-back (64 B/clk), whatever the prefetch distance or spreading, while
// 16x16x4 writes back half as much per flop.  fp32 MFMA runs at the fp32 vector rate (32 cycles per 16x16x4 per SIMD).
// ------------------------------------------------------------------------------------------------
#ifndef VFR_LSTM_NBUF
#define VFR_LSTM_NBUF 2      // the fused LSTM step's own choice (experiment switch)
#endif
#ifndef VFR_GLOAD_SLICE
#define VFR_GLOAD_SLICE 3      /* k-slice of a K-tile after which the global loads of K-tile kt+2 are issued (1, 2: within run-to-run noise) */
#endif
#ifndef VFR_GEMM_PIPE
#define VFR_GEMM_PIPE 1      // 1: K-tile loop software-pipelined across the tile boundary (see the main loop); 0: the plain loop
#endif
#ifndef VFR_LSTM_DEPTH
#ifndef VFR_LSTM_CELL_EARLY
#define VFR_LSTM_CELL_EARLY 1   // 1: the gate epilogue's loads (previous cell state, biases) are requested at the head of the tile; 0: at the head of the epilogue
#endif
#define VFR_LSTM_DEPTH 1     // fragment prefetch distance (k-slices) of the fused LSTM step; the other MFMA kernels use 2
#endif
#ifndef VFR_LSTM_WAVES
#define VFR_LSTM_WAVES 2
#endif
#ifndef VFR_GEMM_NBUF
#define VFR_GEMM_NBUF 2      // LDS tile buffers: 2 = double buffered (2 workgroups/CU), 1 = single (3 workgroups/CU)
#endif
#ifndef VFR_GEMM_SETPRIO
#define VFR_GEMM_SETPRIO 0
#endif
#ifndef VFR_GEMM_SPREAD
#define VFR_GEMM_SPREAD 2    // 2: staging stores interleaved 1:1 with the MFMAs of slices 1-2, order pinned (see the K-tile loop); 1: interleaved, order left to the scheduler; 0: one block behind slice 1
#endif
#ifndef VFR_GEMM_DEEP
#define VFR_GEMM_DEEP 0      // 1: two staging register sets, K-tiles kt + 2 and kt + 3 in flight (see DEEP in the kernel body); 0: one.
                             // EXPERIMENT (round 3): bit-identical, +20..32 VGPRs, and no faster anywhere -- the fused LSTM step at 64 / 200 /
                             // 625 / 1250 / 5000 queries 0.745 / 0.953 / 1.80 / 3.10 / 10.09 ms per pass against 0.730 / 0.955 / 1.81 / 3.09 /
                             // 10.16 with one set: the small-batch step is not waiting for its staging loads (nor for its fragment reads:
                             // VFR_LSTM_DEPTH 2 changes nothing either).  Off.
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MBM = 128, MBN = 128, MBK = 32, MLD_PAD = 36;

template <bool VEC>
__device__ __forceinline__ float4 load4_guard(const float *__restrict__ P, int64_t ld, int64_t row, int64_t nrows,
                                              int k, int K)
{
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row >= nrows) return r;
    const float *p = P + row * ld + k;
    if (VEC && k + 3 < K) return *reinterpret_cast<const float4 *>(p);
    if (k < K) r.x = p[0];
    if (k + 1 < K) r.y = p[1];
    if (k + 2 < K) r.z = p[2];
    if (k + 3 < K) r.w = p[3];
    return r;
}

struct GemmPair { GemmArgs p[2]; };

#ifdef VFR_GEMM_STAMPS
__device__ unsigned long long g_gemm_stamps[4];
#define GSTAMP(i) { if (LSTM) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); gst[i] += t_ - gt0; gt0 = t_; } }
#else
#define GSTAMP(i)
#endif

// MI = 32-row MFMA tiles per wave along M: 2 -> 128-row workgroup tile, 1 -> 64-row tile (half the work per workgroup, for
// launches whose 128-row grid would leave most of the chip idle or waiting on a second, nearly empty round)
//
// PP = ping-pong schedule (EXPERIMENT, off by default: vfr_set_option("gemm_pp", 1)).  A 512-thread workgroup holds two
// independent 4-wave tile groups (one wave of each on every SIMD) whose phases are interleaved by workgroup barriers -- group
// 0 runs the MFMA block of K-tile k while group 1 writes its K-tile k+1 to LDS and issues the loads of k+2, then they swap --
// so every SIMD always has exactly one MFMA stream and a group needs a single LDS buffer.  The idea: two free-running
// workgroups per CU pull each other into lock step (the one behind gets the whole matrix pipe while the other loads) and the
// pipe then idles through both load blocks (PMC: 77 % occupancy dense, 59 % LSTM step).  Measured: bit-identical results but
// SLOWER (dense 122 -> 98 TF, LSTM step 0.66 -> 0.79 ms): in-kernel stamps put 38 % of a wave's time in its MFMA blocks, 27 %
// in its (tiny) load blocks and 25 % at the barriers -- the load block of one group crawls beside the other group's MFMA
// block (raising that block's wave priority with s_setprio changes nothing).  Kept as a switchable variant for the next round's
// work on that stall.
// NARROW = 64-column workgroup tile, the four waves stacked along M (each wave still 16*TI x 64): for operands with N <= 64
// (VGG conv1_1 / conv1_2), where the 128-column tile would spend half of every MFMA on columns that do not exist.
// LFAST = the fused LSTM step in the shape the bench and the model run it: chains start from the projection table (K == 0: one
// segment) and the recurrent segment is whole K-tiles (+ a remainder of <= 8 taken as direct fragments) -- no segment select
// per staging load, no zero-fill select per staged A chunk: no vector ALU instruction left in the K-loop (14 v_cndmask per 64
// MFMAs in the general form, and VALU time is matrix-pipe time).  lstm_step_pair picks it when the launch qualifies.
// CHALO = the scalar-tap convolution loader on HALO-PADDED activations: input [B][H + 2][W + 2][C_in] with a zero border, output
// written into the interior of a tensor padded the same way (pooled size under EPI_POOL2).  Every tap of every output pixel then
// reads real memory holding the right value -- zero outside the image -- so the loader needs no tap mask, no address select and
// no zero-fill select: like the dense loop, no vector ALU instruction per K-tile (32 per 128 MFMAs in the masked form).
// MULTI = the LFAST step as ONE TASK of the multi-step kernel (lstm_steps_mfma_kernel below): the tile (task_by, task_m0) comes
// from the task decode, and every byte of recurrent state -- h (the staged A operand, the remainder fragments), the previous
// cell state, the stores of the new h / c -- moves with AGENT-scope accesses (sc1: write-through stores, loads that do not
// trust this XCD's L2), because the producer of a row's h ran on another XCD in the SAME launch.  The weights and the
// projection table are read-only and keep the plain (L2-resident) path.
typedef unsigned int vfr_u32x4 __attribute__((ext_vector_type(4)));
template <bool VEC, bool CONV, bool LSTM = false, int MI = 2, bool PP = false, bool NARROW = false, bool TRAIN = false, bool CFAST = false, bool LFAST = false, bool CHALO = false, bool MULTI = false>
__device__ __forceinline__ void gemm_nt_mfma_body(const GemmArgs &g, int task_by = 0, int64_t task_m0 = 0, int task_tid = 0)
{
    static_assert(!MULTI || (LFAST && !TRAIN), "MULTI: the table-start LSTM step inside the multi-step kernel");
    static_assert(!CFAST || CONV, "CFAST: the convolution loader for C_in a multiple of 32");
    static_assert(!CHALO || CFAST, "CHALO: the scalar-tap loader on halo-padded activations");
    static_assert(!LFAST || (LSTM && !PP), "LFAST: the table-start LSTM step");
    static_assert(!TRAIN || LSTM, "TRAIN: the fused LSTM step that also stores its gates");
    static_assert(!NARROW || (MI == 1 && !LSTM && !PP), "narrow tile: 128 x 64 only");
    constexpr int TBM = NARROW ? 128 * MI : (MI ? 64 * MI : 32);    // tile rows (MI = 0: the 32-row tile, one row tile per wave)
    constexpr int BN = NARROW ? 64 : MBN;               // tile columns
    constexpr int NA = TBM / 32, NW = BN / 32;          // float4 staging loads per thread of A / of W
    // LDS row layout.  128-row tiles: 32 floats + 4 of padding (row stride 36: the 16 rows x 4 k's of a fragment read fall in
    // 64 distinct banks).  Smaller tiles: blocks of 16 rows, row r of a block at 32*r + 4*(r >> 1) floats, block stride 544 =
    // 34 per row -- equally conflict-free (bank = 32*(r&1) + 4*(r>>1) + 4*k4 + q), LINEAR in the k-slice (every fragment read
    // of a K-tile is base + immediate; an XOR swizzle of the 16-byte chunks, the first version, cost three address VALU ops
    // per slice, and VALU time is matrix-pipe time on gfx950: tools/ubench/coissue.hip), every float4 16-byte aligned, and
    // the 192-row double buffer is 51 KB instead of 54 KB, so THREE workgroups fit the CU's 160 KB.
    constexpr bool SKEW = MI <= 1 && !PP;
    constexpr int RBLK = SKEW ? 544 : 16 * MLD_PAD;             // floats per block of 16 rows
    constexpr int A_FLOATS = (TBM / 16) * RBLK, BUF_FLOATS = ((TBM + BN) / 16) * RBLK;
    auto lds_off = [](int row, int k) {                         // float offset of (row, k) inside a tile
        return SKEW ? (row >> 4) * 544 + (row & 15) * 32 + 4 * ((row & 15) >> 1) + k : row * MLD_PAD + k;
    };
    // double-buffered tiles: [2][A | W] -> two (padded 128-row tiles) or three workgroups per CU
    constexpr int NBUF = PP ? 1 : (LSTM ? VFR_LSTM_NBUF : VFR_GEMM_NBUF);
    __shared__ __attribute__((aligned(16))) float lds_all[(PP ? 2 : 1) * NBUF * BUF_FLOATS];
    __shared__ int otab_s[CHALO ? TBM : 1];              // CHALO: padded output pixel of every tile row (pooled pixel under EPI_POOL2)
    const int grp = PP ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;      // tile group of this wave
    float *lds = lds_all + grp * NBUF * BUF_FLOATS;
    const int tid = PP ? (int)(threadIdx.x & 255) : MULTI ? task_tid : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = NARROW ? wave : wave >> 1, wn = NARROW ? 0 : wave & 1;
    int64_t m0 = (int64_t)(PP ? blockIdx.x * 2 + grp : blockIdx.x) * TBM;
    int n0 = blockIdx.y * BN;
    int by = blockIdx.y;                                 // column tile (LSTM step: 4 gates x 32 units)
    bool active = true;                                  // PP: a group without a tile still takes part in the barriers
    if constexpr (MULTI) { by = task_by; m0 = task_m0; }
    else
    if (LSTM && g.xcd_cols > 0) {
        // XCD-aware order of the fused step (1-D grid per direction; workgroup L runs on XCD L % 8): each XCD owns xcd_cols
        // adjacent column tiles -- its 2 MB slice of W_hh stays in its 4 MB L2 for the whole launch -- and walks the row
        // tiles with the column tile fastest, so the xcd_cols workgroups that share an h row tile run together.  In launch
        // order every XCD touched all 16 MB of W_hh: 1.3 GB of L2 misses per launch for 72 MB of operands.
        const unsigned L = blockIdx.x, j = L >> 3, cpx = (unsigned)g.xcd_cols;
        by = (int)((L & 7u) * cpx + j % cpx);
        m0 = (int64_t)(j / cpx) * TBM;
        if (by * 32 >= g.lstm_H) return;                 // grid padded to 8 * xcd_cols column tiles
    }
    if (!LSTM && g.xcd_cols > 0) {
        const unsigned L = blockIdx.x, per = 8u * (unsigned)g.xcd_cols, within = L % per;
        const int64_t rt = (int64_t)(L / per) * 8 + (within & 7u);
        m0 = (PP ? rt * 2 + grp : rt) * TBM;
        n0 = (int)(within >> 3) * BN;
        if (m0 >= g.M) { if (!PP) return; active = false; }     // grid padded to whole groups of 8 row tiles
    }
    if (PP && !LSTM && m0 >= g.M) active = false;
    // LSTM step with a shrinking / growing active prefix: rows past the active count do nothing this step
    const int64_t Mrows = (LSTM && g.lstm_mcount) ? (int64_t)g.lstm_mcount[g.lstm_step] : g.M;
    const int64_t Mprev = (LSTM && g.lstm_mcount) ? (g.lstm_step > 0 ? (int64_t)g.lstm_mcount[g.lstm_step - 1] : 1) : Mrows;
    if (LSTM && m0 >= Mrows) { if (!PP) return; active = false; }
    if (!active) m0 = 0;                                 // keeps every clamped address below in range; nothing is stored

#ifdef VFR_GEMM_STAMPS
    unsigned long long gst[4] = {0, 0, 0, 0}, gt0 = __builtin_amdgcn_s_memtime();
#endif
    // accumulators: the wave's 16*TI x 64 outputs as TI x 4 tiles of v_mfma_f32_16x16x4_f32 (4 registers each); lane l holds
    // rows 4*(l>>4) + r (r = 0..3) and column l & 15 of a tile
    constexpr int TI = MI ? 2 * MI : 1;
    const int l15 = lane & 15, lq = lane >> 4;
    f32x4 acc[TI][4];
    auto init_acc = [&]() {
#ifdef VFR_LSTM_NOGATHER      /* TIMING EXPERIMENT ONLY (wrong results): accumulators start from zero, no table gather */
        if (LSTM && g.lstm_tok) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
            return;
        }
#endif
        if (LSTM && g.lstm_tok) {
            // chains start from the vocabulary input-projection table: P[lstm_tok[row]][tile column].  Two batched load
            // rounds (all table-row indices, then all accumulators), not dependent pairs one after the other.
            int64_t prow[TI][4];
    #pragma unroll
            for (int ti = 0; ti < TI; ++ti)
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = m0 + wm * (16 * TI) + ti * 16 + 4 * lq + r;
                    prow[ti][r] = (int64_t)g.lstm_tok[row < Mrows ? row : Mrows - 1];
                }
            const float *pcol = g.Cin + (int64_t)by * MBN + wn * 64 + l15;
    #pragma unroll
            for (int ti = 0; ti < TI; ++ti)
    #pragma unroll
                for (int r = 0; r < 4; ++r)
    #pragma unroll
                    for (int tj = 0; tj < 4; ++tj) acc[ti][tj][r] = pcol[prow[ti][r] * g.ldc + tj * 16];
        } else {
    #pragma unroll
            for (int ti = 0; ti < TI; ++ti)
    #pragma unroll
                for (int tj = 0; tj < 4; ++tj)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t row = m0 + wm * (16 * TI) + ti * 16 + 4 * lq + r;
                        const int col = n0 + wn * 64 + tj * 16 + l15;
                        acc[ti][tj][r] = (g.Cin && row < g.M && col < g.N) ? g.Cin[row * g.ldc + col] : 0.0f;
                    }
        }
    };

    // staging registers: two sets in the software-pipelined loop (DEEP: K-tile kt + 2 is in flight in one set while kt + 3 is
    // requested into the other -- 1 3/4 tiles of cover for the load latency instead of 3/4: it is what a 32-row tile, whose MFMA
    // block is short, needs to keep enough bytes in flight), one set otherwise
    constexpr bool DEEP = VFR_GEMM_DEEP && !PP && (LSTM ? VFR_LSTM_NBUF : VFR_GEMM_NBUF) == 2 && VFR_GEMM_PIPE;
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    float4 ra[DEEP ? 2 : 1][NA], rw[DEEP ? 2 : 1][NW];
    // Zero-fill of staged elements outside the operand (conv padding taps, the tail of a segmented K) is decided when the
    // load is ISSUED but applied (to the A side only) when the registers are written to LDS: a select right after the load
    // would make hipcc wait for the load (vmcnt(0)) before the MFMA block and expose its latency in every K-tile.
    bool za[DEEP ? 2 : 1][NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) { za[0][i] = false; za[DEEP ? 1 : 0][i] = false; }
    // Staging loads.  Full K-tiles use UNCONDITIONAL loads (row index clamped into range; rows past M / N are
    // never stored) so the compiler can leave them in flight across the MFMA block -- a per-load bounds branch
    // makes hipcc drain vmcnt(0) right after issuing them.  Only the last, partial K-tile takes the guarded form.
    const float *arow[NA], *wrow[NW];
    // Fused LSTM step: the staged rows are addressed as UNIFORM base (segment pointer + the tile's k, scalar registers) +
    // a per-thread 32-bit byte offset that never changes (row * ld + the thread's k within a tile), so a K-tile costs one
    // select per load and no 64-bit vector arithmetic (lstm_step_pair checks that the offsets fit 32 bits).  A segment
    // shorter than one K-tile has a single tile whose out-of-range lanes re-read k = 0 (zeroed on the A side).
    unsigned aoff1[NA], aoff2[NA], woff1[NW], woff2[NW];
    const int kkl = (tid & 7) * 4;
    const int kk1 = (LSTM && g.K < MBK && kkl >= g.K) ? 0 : kkl, kk2 = (LSTM && g.K2 < MBK && kkl >= g.K2) ? 0 : kkl;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int f = tid + 256 * i, row = f >> 3, kk = (f & 7) * 4;
        int64_t nw = (int64_t)n0 + row < g.N ? (int64_t)n0 + row : g.N - 1;
        if (LSTM) {   // tile column c = (wn, tj, l15): gate = tj = (c >> 4) & 3, unit = 32*blockIdx.y + 16*wn + (c & 15)
            const int gate = (row >> 4) & 3;
            int unit = by * 32 + (row >> 6) * 16 + (row & 15);
            unit = unit < g.lstm_H ? unit : g.lstm_H - 1;
            nw = (int64_t)gate * g.lstm_H + unit;
            woff1[i] = (unsigned)((nw * g.ldw + kk1) * 4);
            woff2[i] = (unsigned)((nw * g.ldw2 + kk2) * 4);
        }
        wrow[i] = g.W + nw * g.ldw + kk;
        if (!LSTM) woff1[i] = (unsigned)(((nw - n0) * g.ldw + kk) * 4);     // dense: from the tile's first row (gload_full)
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = tid + 256 * i, row = f >> 3, kk = (f & 7) * 4;
        const int64_t ma = m0 + row < Mrows ? m0 + row : Mrows - 1;
        if (LSTM) {
            const int64_t hsrc = ma >= Mprev ? 0 : ma;             // a row joining now continues from the pad row's state
            aoff2[i] = (unsigned)((hsrc * g.lda2 + kk2) * 4);
            aoff1[i] = (unsigned)(((g.lstm_xrow ? (int64_t)g.lstm_xrow[ma] : ma) * g.lda + kk1) * 4);
        }
        arow[i] = CONV ? g.A : g.A + ma * g.lda + kk;
        if (!LSTM && !CONV) aoff1[i] = (unsigned)(((ma - m0) * g.lda + kk) * 4);
    }
    auto gload_full = [&](int k0, auto sc) {
        constexpr int S = decltype(sc)::value;
        // aligned operands: uniform tile base (scalar registers) + the thread's fixed 32-bit offset inside the tile's rows
        // (at most 127 rows of lda floats: gemm_nt checks that this fits 32 bits) -- no vector address arithmetic per K-tile
        const char *ba = reinterpret_cast<const char *>(g.A + m0 * g.lda + k0);
        const char *bw = reinterpret_cast<const char *>(g.W + (int64_t)n0 * g.ldw + k0);
        // (the empty asm keeps the 32-bit offset a 32-bit register inside the loop: hoisted as a zero-extended pair it would be
        // added to the base with two-pass 64-bit vector adds instead of going into the load's scalar-base address form)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            asm volatile("" : "+v"(aoff1[i]));
            if (VEC) ra[S][i] = *reinterpret_cast<const float4 *>(ba + aoff1[i]);
            else     ra[S][i] = make_float4(arow[i][k0], arow[i][k0 + 1], arow[i][k0 + 2], arow[i][k0 + 3]);
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            asm volatile("" : "+v"(woff1[i]));
            if (VEC) rw[S][i] = *reinterpret_cast<const float4 *>(bw + woff1[i]);
            else     rw[S][i] = make_float4(wrow[i][k0], wrow[i][k0 + 1], wrow[i][k0 + 2], wrow[i][k0 + 3]);
        }
    };
    auto gload_tail = [&](int k0, auto sc) {
        constexpr int S = decltype(sc)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + 256 * i, row = f >> 3, kk = (f & 7) * 4;
            ra[S][i] = load4_guard<false>(g.A, g.lda, m0 + row, g.M, k0 + kk, g.K);
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int f = tid + 256 * i, row = f >> 3, kk = (f & 7) * 4;
            rw[S][i] = load4_guard<false>(g.W, g.ldw, (int64_t)n0 + row, g.N, k0 + kk, g.K);
        }
    };
    // ---- implicit-GEMM conv loader: per staged row the output pixel is fixed, per K-tile the thread's 4 consecutive
    // k's share one tap (conv_cin % 4 == 0); loads are unconditional from a clamped address, then zero-selected.
    int cn[NA], coy[NA], cox[NA];
    bool crow_ok[NA];
    // CFAST (C_in % 32 == 0): a K-tile lies inside ONE filter tap, so the tap, its pixel displacement and the channel offset
    // are scalar per tile; per staged row only a 9-bit mask of the taps that fall inside the image and a fixed 32-bit offset
    // (from the tile's first pixel, biased by one image row + 1 so that it stays non-negative under every displacement) remain
    unsigned ctap[NA], coffb[NA];
    const int cbias = CFAST ? (g.conv_w + 1) * g.conv_cin : 0;          // floats
    // GEMM row -> output pixel (n, y, x) and its linear NHWC pixel index.  Plain order: row = pixel index.  EPI_POOL2: the rows
    // of a 2x2 pooling window are adjacent (row = window * 4 + (y & 1) * 2 + (x & 1), windows in pooled-NHWC order), so a lane's
    // accumulator quad is one window.  The linear index grows with the row in both orders (window starts ascending).
    const bool cpool = CONV && (g.epi & EPI_POOL2) != 0;
    auto conv_pixel = [&](int64_t pc, int &n_, int &y_, int &x_) -> int64_t {
        const int hw = g.conv_h * g.conv_w;
        if (g.M < (1ll << 31)) {
            // (the usual case in 32-bit arithmetic: a 64-bit division is ~100 instructions, and a K = 576 layer's tile is short)
            const unsigned pu = (unsigned)pc, w_ = (unsigned)g.conv_w;
            if (cpool) {
                const unsigned q = pu >> 2, sub = pu & 3u, hw4 = (unsigned)hw >> 2, w2 = w_ >> 1;
                const unsigned nn = q / hw4, rem = q - nn * hw4, y2 = rem / w2;
                n_ = (int)nn; y_ = (int)(2 * y2 + (sub >> 1)); x_ = (int)(2 * (rem - y2 * w2) + (sub & 1u));
            } else {
                const unsigned nn = pu / (unsigned)hw, rem = pu - nn * (unsigned)hw, yy = rem / w_;
                n_ = (int)nn; y_ = (int)yy; x_ = (int)(rem - yy * w_);
            }
            return ((int64_t)n_ * g.conv_h + y_) * g.conv_w + x_;
        }
        if (cpool) {
            const int64_t q = pc >> 2;
            const int sub = (int)(pc & 3), hw4 = hw >> 2, w2 = g.conv_w >> 1;
            n_ = (int)(q / hw4);
            const int rem = (int)(q - (int64_t)n_ * hw4), y2 = rem / w2;
            y_ = 2 * y2 + (sub >> 1);
            x_ = 2 * (rem - y2 * w2) + (sub & 1);
        } else {
            n_ = (int)(pc / hw);
            const int rem = (int)(pc - (int64_t)n_ * hw);
            y_ = rem / g.conv_w;
            x_ = rem - y_ * g.conv_w;
        }
        return ((int64_t)n_ * g.conv_h + y_) * g.conv_w + x_;
    };
    int64_t clin0 = m0;                                                 // linear pixel index of the tile's first row
    if (CONV && cpool) { int n0_, y0_, x0_; clin0 = conv_pixel(m0 < g.M ? m0 : 0, n0_, y0_, x0_); }
    int64_t chalo0 = 0;                                                 // CHALO: padded index of the tile's first pixel
    if constexpr (CHALO) {
        int n0_, y0_, x0_;
        (void)conv_pixel(m0 < g.M ? m0 : 0, n0_, y0_, x0_);
        chalo0 = ((int64_t)n0_ * (g.conv_h + 2) + y0_ + 1) * (g.conv_w + 2) + x0_ + 1;
    }
    if (CONV) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + 256 * i, row = f >> 3;
            const int64_t p = m0 + row;
            crow_ok[i] = p < g.M;
            const int64_t pc = crow_ok[i] ? p : 0;
            const int64_t plin = conv_pixel(pc, cn[i], coy[i], cox[i]);
            if constexpr (CHALO) {
                const int wp = g.conv_w + 2;
                const int64_t pin = ((int64_t)cn[i] * (g.conv_h + 2) + coy[i] + 1) * wp + cox[i] + 1;     // padded centre pixel
                coffb[i] = (unsigned)(((crow_ok[i] ? pin - chalo0 : 0) * g.conv_cin + (f & 7) * 4) * 4);
                ctap[i] = 0;
                if ((f & 7) == 0) {
                    const int oh = cpool ? (g.conv_h >> 1) : g.conv_h, ow = cpool ? (g.conv_w >> 1) : g.conv_w;
                    const int oy = cpool ? (coy[i] >> 1) : coy[i], ox = cpool ? (cox[i] >> 1) : cox[i];
                    otab_s[row] = (int)(((int64_t)cn[i] * (oh + 2) + oy + 1) * (ow + 2) + ox + 1);
                }
            } else
            if (CFAST) {
                unsigned mask = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = coy[i] + t / 3 - 1, ix = cox[i] + t % 3 - 1;
                    if (crow_ok[i] && iy >= 0 && iy < g.conv_h && ix >= 0 && ix < g.conv_w) mask |= 1u << t;
                }
                ctap[i] = mask;
                coffb[i] = (unsigned)(((crow_ok[i] ? plin - clin0 : 0) * g.conv_cin + (f & 7) * 4 + cbias) * 4);
            }
        }
    }
    const int ctpt = CFAST ? g.conv_cin / MBK : 1;                      // K-tiles per tap
    auto gload_conv_fast = [&](int k0, auto sc) {
        constexpr int S = decltype(sc)::value;
        const int kt = k0 / MBK, tap = kt / ctpt, ci0 = (kt - tap * ctpt) * MBK;
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;             // tap / 3 for tap < 9
        if constexpr (CHALO) {
            const int dp = (ky - 1) * (g.conv_w + 2) + (kx - 1);
            const char *ba = reinterpret_cast<const char *>(g.A + ((chalo0 + dp) * g.conv_cin + ci0));
            const char *bw = reinterpret_cast<const char *>(g.W + (int64_t)n0 * g.ldw + k0);
#pragma unroll
            for (int i = 0; i < NA; ++i) { asm volatile("" : "+v"(coffb[i])); ra[S][i] = *reinterpret_cast<const float4 *>(ba + coffb[i]); }
#pragma unroll
            for (int i = 0; i < NW; ++i) { asm volatile("" : "+v"(woff1[i])); rw[S][i] = *reinterpret_cast<const float4 *>(bw + woff1[i]); }
            return;
        }
        const int dpix = (ky - 1) * g.conv_w + (kx - 1);
        const unsigned dbytes = (unsigned)(dpix * g.conv_cin * 4), tbit = 1u << tap;
        // (m0 + dpix) * C_in + ci0 - bias may lie before the tensor: only ever dereferenced with an in-image offset added
        const char *ba = reinterpret_cast<const char *>(g.A + ((clin0 + dpix) * g.conv_cin + ci0 - cbias));
        const char *bw = reinterpret_cast<const char *>(g.W + (int64_t)n0 * g.ldw + k0);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool ok = (ctap[i] & tbit) != 0;
            // outside the image: the row's own pixel (always inside), zeroed when the registers go to LDS
            ra[S][i] = *reinterpret_cast<const float4 *>(ba + (ok ? coffb[i] : coffb[i] - dbytes));
            za[S][i] = !ok;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            asm volatile("" : "+v"(woff1[i]));
            rw[S][i] = *reinterpret_cast<const float4 *>(bw + woff1[i]);
        }
    };
    auto gload_conv = [&](int k0, auto sc) {
        constexpr int S = decltype(sc)::value;
        const int kk = (tid & 7) * 4, k = k0 + kk;
        const bool kok = k < g.K;
        const int kc = kok ? k : 0;
        const int tap = kc / g.conv_cin, ci = kc - tap * g.conv_cin, ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int iy = coy[i] + ky - 1, ix = cox[i] + kx - 1;
            const bool ok = kok && crow_ok[i] && iy >= 0 && iy < g.conv_h && ix >= 0 && ix < g.conv_w;
            const int64_t off = ok ? (((int64_t)cn[i] * g.conv_h + iy) * g.conv_w + ix) * g.conv_cin + ci : 0;
            ra[S][i] = *reinterpret_cast<const float4 *>(g.A + off);
            za[S][i] = !ok;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) rw[S][i] = *reinterpret_cast<const float4 *>(wrow[i] - kk + kc);    // row clamped, k clamped
    };
    // ---- segmented-K loader (LSTM step): tiles [0, nk1) walk [A | W] over K, tiles [nk1, nk1+nk2) walk [A2 | W2] over K2.
    // A partial last tile of a segment is shifted BACK to end exactly at the segment's end (nothing is read past a row);
    // its lanes below the previous tile's end are zeroed on the A side when the registers go to LDS (fma(0, w, acc) == acc for
    // the finite w re-read beside them), and so are the lanes past the end of a segment shorter than one tile.
    const int nk1 = (g.K + MBK - 1) / MBK;
    bool zq = false;
    // MULTI: buffer resource over the h operand (raw buffer, 32-bit byte offsets: lstm_steps_pair checks the 4 GB bound)
    __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(MULTI ? g.A2 : nullptr), 0, MULTI ? 0xFFFFFFFFu : 0u, 0x00020000);
    auto gload_seg = [&](int k0t, auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (LFAST) {           // one segment of whole K-tiles: scalar base + the thread's fixed offsets, nothing to select
            const char *ba = reinterpret_cast<const char *>(g.A2 + k0t);
            const char *bw = reinterpret_cast<const char *>(g.W2 + k0t);
#ifdef VFR_MULTI_PLAIN_A     /* TIMING EXPERIMENT ONLY (incoherent: h may be read stale): the A operand through the plain L2 path */
            if constexpr (false) {
#else
            if constexpr (MULTI) {       // h of the previous step, written in this launch by other XCDs: 16-byte loads at agent scope (sc1)
#endif
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    asm volatile("" : "+v"(aoff2[i]));
                    const vfr_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, (int)aoff2[i], k0t * 4, 16);
                    ra[S][i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
            } else
#pragma unroll
            for (int i = 0; i < NA; ++i) { asm volatile("" : "+v"(aoff2[i])); ra[S][i] = *reinterpret_cast<const float4 *>(ba + aoff2[i]); }
#pragma unroll
            for (int i = 0; i < NW; ++i) { asm volatile("" : "+v"(woff2[i])); rw[S][i] = *reinterpret_cast<const float4 *>(bw + woff2[i]); }
            return;
        }
        const int kt = k0t / MBK;
        const bool second = kt >= nk1;
        const int Kseg = second ? g.K2 : g.K, kseg = (second ? kt - nk1 : kt) * MBK;
        const int kb = (kseg + MBK > Kseg && Kseg >= MBK) ? Kseg - MBK : kseg;
        const int zlo = kseg - kb, zhi = Kseg < MBK ? Kseg : MBK;
        zq = kkl < zlo || kkl >= zhi;
        const char *ba = reinterpret_cast<const char *>((second ? g.A2 : g.A) + kb);
        const char *bw = reinterpret_cast<const char *>((second ? g.W2 : g.W) + kb);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            ra[S][i] = *reinterpret_cast<const float4 *>(ba + (second ? aoff2[i] : aoff1[i]));
            za[S][i] = zq;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) rw[S][i] = *reinterpret_cast<const float4 *>(bw + (second ? woff2[i] : woff1[i]));
    };
    // LSTM step on the projection table (K == 0) with a short remainder of the recurrent segment (H = 1000: 31 K-tiles + 8):
    // the remainder does not get a K-tile of its own (a shifted-back tile is 24 zero products in 32 -- 2.3 % of the launch's
    // MFMAs); its <= 2 k-slices are fetched as MFMA fragments straight from global memory at the head of the kernel and run
    // after the last full tile -- the same k-ascending chain, bit for bit.
    const int ktail = (LSTM && !PP && g.K == 0 && g.K2 > MBK && (g.K2 % MBK) != 0 && (g.K2 % MBK) <= 8) ? g.K2 % MBK : 0;
    constexpr int TI_ = MI ? 2 * MI : 1;
    float tfa[2][TI_], tfb[2][4];
    if (LSTM && ktail) {
        const int kb = g.K2 - ktail + (lane >> 4);
#pragma unroll
        for (int ti = 0; ti < TI_; ++ti) {
            const int64_t row = m0 + wm * (16 * TI_) + ti * 16 + (lane & 15);
            const int64_t ma = row < Mrows ? row : Mrows - 1;
            const float *pa = g.A2 + (ma >= Mprev ? 0 : ma) * g.lda2 + kb;
            if constexpr (MULTI) {
                tfa[0][ti] = __hip_atomic_load(pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tfa[1][ti] = ktail > 4 ? __hip_atomic_load(pa + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
            } else {
            tfa[0][ti] = pa[0];
            tfa[1][ti] = ktail > 4 ? pa[4] : 0.0f;
            }
        }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            int unit = by * 32 + wn * 16 + (lane & 15);
            unit = unit < g.lstm_H ? unit : g.lstm_H - 1;
            const float *pw = g.W2 + ((int64_t)tj * g.lstm_H + unit) * g.ldw2 + kb;
            tfb[0][tj] = pw[0];
            tfb[1][tj] = ktail > 4 ? pw[4] : 0.0f;
        }
    }
    auto tail_mfma = [&]() {
        if (LSTM && ktail) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (s == 1 && ktail <= 4) break;
#pragma unroll
                for (int ti = 0; ti < TI_; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(tfa[s][ti], tfb[s][tj], acc[ti][tj], 0, 0, 0);
            }
        }
    };
    const int nk_full = LSTM ? nk1 + (ktail ? g.K2 / MBK : (g.K2 + MBK - 1) / MBK)
                             : CONV ? (g.K + MBK - 1) / MBK : g.K / MBK;     // conv / lstm: every tile through a select loader
    auto swrite = [&](int b, auto sc) {
        constexpr int S = decltype(sc)::value;
        float *As = lds + b * BUF_FLOATS, *Ws = As + A_FLOATS;
#ifdef VFR_GEMM_NOSWRITE      /* TIMING EXPERIMENT ONLY (wrong results): staged registers are consumed but never written to LDS */
#pragma unroll
        for (int i = 0; i < NA; ++i) asm volatile("" :: "v"(ra[S][i].x), "v"(ra[S][i].y), "v"(ra[S][i].z), "v"(ra[S][i].w));
#pragma unroll
        for (int i = 0; i < NW; ++i) asm volatile("" :: "v"(rw[S][i].x), "v"(rw[S][i].y), "v"(rw[S][i].z), "v"(rw[S][i].w));
        return;
#endif
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + 256 * i, row = f >> 3, kk = (f & 7) * 4;
            float4 v = ra[S][i];
            if ((CONV && !CHALO) || (LSTM && !LFAST)) { const bool z = za[S][i]; v.x = z ? 0.f : v.x; v.y = z ? 0.f : v.y; v.z = z ? 0.f : v.z; v.w = z ? 0.f : v.w; }
            *reinterpret_cast<float4 *>(&As[lds_off(row, kk)]) = v;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int f = tid + 256 * i, row = f >> 3, kk = (f & 7) * 4;
            float4 v = rw[S][i];
            // (W is not zero-filled: wherever a k lies outside the operand the A side is zero, and the weight re-read from a
            // clamped address beside it is finite, so the product leaves the chain as it is)
            *reinterpret_cast<float4 *>(&Ws[lds_off(row, kk)]) = v;
        }
    };
    // one staged float4 (piece j < NA: of A, else of W) -> LDS: the pieces of swrite, for the interleaved schedule
    auto swrite_piece = [&](int b, auto sc, int j) {
        constexpr int S = decltype(sc)::value;
        float *As = lds + b * BUF_FLOATS, *Ws = As + A_FLOATS;
        if (j < NA) {
            const int f = tid + 256 * j, row = f >> 3, kk = (f & 7) * 4;
            float4 v = ra[S][j];
            if ((CONV && !CHALO) || (LSTM && !LFAST)) { const bool z = za[S][j]; v.x = z ? 0.f : v.x; v.y = z ? 0.f : v.y; v.z = z ? 0.f : v.z; v.w = z ? 0.f : v.w; }
            *reinterpret_cast<float4 *>(&As[lds_off(row, kk)]) = v;
        } else {
            const int i = j - NA, f = tid + 256 * i, row = f >> 3, kk = (f & 7) * 4;
            *reinterpret_cast<float4 *>(&Ws[lds_off(row, kk)]) = rw[S][i];
        }
    };
    auto compute = [&](int b) {
        const float *As = lds + b * BUF_FLOATS, *Ws = As + A_FLOATS;
        // fragment of tile row block ti at k-slice k4: ONE dword per lane, A[16*ti + (lane & 15)][4*k4 + (lane >> 4)] -- exactly
        // the 16x16x4 operand layout, read with ds_read_b32 (64 distinct banks, conflict-free, in both layouts)
        const float *ap = &As[lds_off(wm * (16 * TI) + l15, lq)];
        const float *wp = &Ws[lds_off(wn * 64 + l15, lq)];
        auto kcol = [&](int k4) { return k4 * 4; };
        // fragment ring: the reads of slice k4+DEPTH are issued (and pinned) BEFORE the MFMAs of slice k4, so their
        // latency hides under the matrix pipe.  DEPTH 2 matters when a workgroup is ALONE on its CU (partial last round, small
        // launches): with one slice of cover a lone wave's stream stalled on LDS latency and a half-filled round cost as much
        // as a full one; with two it costs half.  The LSTM step (64-row tiles, always several rounds) is 1.5 % faster with 1.
        constexpr int DEPTH = LSTM ? VFR_LSTM_DEPTH : 2, RING = DEPTH + 1;
        float fa[RING][TI], fb[RING][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) fa[d][ti] = ap[ti * RBLK + kcol(d)];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) fb[d][tj] = wp[tj * RBLK + kcol(d)];
        }
#pragma unroll
        for (int k4 = 0; k4 < MBK / 4; ++k4) {
            const int cur = k4 % RING, nxt = (k4 + DEPTH) % RING;
            if (k4 + DEPTH < MBK / 4) {
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) fa[nxt][ti] = ap[ti * RBLK + kcol(k4 + DEPTH)];
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) fb[nxt][tj] = wp[tj * RBLK + kcol(k4 + DEPTH)];
            }
            __builtin_amdgcn_sched_barrier(0);
            // one MFMA = k, k+1, k+2, k+3 in order on top of the accumulator: the oracle's chain
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][ti], fb[cur][tj], acc[ti][tj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // main loop over the FULL K-tiles, branch-free: iteration kt prefetches tile min(kt+1, last) (the final
    // iteration re-reads its own tile, which is harmless) so no control-flow join sits between the loads and the
    // MFMA block -- a join there makes the compiler drain vmcnt(0) before the first MFMA.
    // Fused LSTM step: what the gate epilogue needs from memory -- the previous cell state of the lane's 4*TI cells and the four
    // bias pairs of its unit -- is requested at the HEAD of the tile, beside the projection-table gather (one latency window,
    // 12 registers held across the K-loop at 64 rows), not at the head of the epilogue, where nothing else of this wave is in
    // flight to cover it.
    constexpr int TIc = MI ? 2 * MI : 1;
    float cprev[LSTM ? TIc : 1][4];
    float bi = 0.f, bf = 0.f, bg = 0.f, bo = 0.f;
    auto load_cell_inputs = [&]() {
        if constexpr (LSTM) {
            const int H = g.lstm_H, unit = by * 32 + wn * 16 + (lane & 15);
            const int uc = unit < H ? unit : H - 1;
            bi = g.bias[uc] + g.bias2[uc]; bf = g.bias[H + uc] + g.bias2[H + uc];
            bg = g.bias[2 * H + uc] + g.bias2[2 * H + uc]; bo = g.bias[3 * H + uc] + g.bias2[3 * H + uc];
#pragma unroll
            for (int ti = 0; ti < TIc; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = m0 + wm * (16 * TIc) + ti * 16 + 4 * (lane >> 4) + r;
                    const int64_t rc = row < Mrows ? row : Mrows - 1;
                    if constexpr (MULTI) cprev[ti][r] = __hip_atomic_load(g.lstm_cin + (rc >= Mprev ? 0 : rc) * H + uc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else cprev[ti][r] = g.lstm_cin[(rc >= Mprev ? 0 : rc) * H + uc];
                }
        }
    };
    constexpr bool CELL_EARLY = LSTM && !PP && NBUF == 2 && VFR_GEMM_PIPE && VFR_LSTM_CELL_EARLY;
    auto gload_main = [&](int k0, auto sc) { if (LSTM) gload_seg(k0, sc); else if (CFAST) gload_conv_fast(k0, sc); else if (CONV) gload_conv(k0, sc); else gload_full(k0, sc); };
    if (PP) {
        // group g runs  C0 L1 C1 L2 ... C(nk-1)  delayed by g phases;  Ck = MFMA block on K-tile k (LDS), Lk = K-tile k from
        // registers to LDS + issue the loads of K-tile k+1.  One workgroup barrier per phase.
        const int nk = nk_full;
        init_acc();
        if (active && nk > 0) { gload_main(0, S0); swrite(0, S0); if (nk > 1) gload_main(MBK, S0); }
        __syncthreads();
        GSTAMP(0)
        for (int p = 0; p < 2 * nk; ++p) {
            const int q = p - grp;
            if (active && q >= 0 && q < 2 * nk - 1) {
                const int k = q >> 1;
                if (q & 1) { swrite(0, S0); if (k + 2 < nk) gload_main((k + 2) * MBK, S0); }
                else compute(0);
            }
            __syncthreads();
        }
    } else if (NBUF == 2 && VFR_GEMM_PIPE) {
        // K-tile loop, software-pipelined so that NOTHING but the barrier itself sits between two MFMA blocks.  Per K-tile kt
        // (8 k-slices of 4, TI*4 MFMAs each):
        //   slice 1   registers (K-tile kt+1, loaded during the previous tile) -> the other LDS buffer: it was last read for
        //             K-tile kt-1 and every wave passed that tile's barrier after issuing its last fragment reads
        //   slice 3   issue the global loads of K-tile kt+2 (unconditional, index clamped: no control-flow join that would
        //             make hipcc drain vmcnt before the MFMAs)
        //   slice 8-DEPTH  barrier (K-tile kt+1 visible), then the fragment reads continue seamlessly into K-tile kt+1
        // so the LDS writes, the load issue and the first fragment reads of the next tile all run under MFMAs of this one.
        constexpr int DEPTH = LSTM ? VFR_LSTM_DEPTH : 2, RING = DEPTH == 1 ? 2 : 4, NS = MBK / 4;
        const int nk = nk_full;
        float fa[RING][TI], fb[RING][4];
        auto frag_read = [&](int buf, int slice, int slot) {
            const float *As = lds + buf * BUF_FLOATS, *Ws = As + A_FLOATS;
            const float *ap = &As[lds_off(wm * (16 * TI) + l15, lq) + slice * 4];
            const float *wp = &Ws[lds_off(wn * 64 + l15, lq) + slice * 4];
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) fa[slot][ti] = ap[ti * RBLK];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) fb[slot][tj] = wp[tj * RBLK];
        };
        // the first K-tile's loads are in flight while the accumulators are initialised (C-in / the LSTM table gather: two
        // dependent load rounds whose latency would otherwise stand alone at the head of every tile)
        if (nk > 0) gload_main(0, S0);
        init_acc();
        if constexpr (CELL_EARLY) load_cell_inputs();
        // Accumulators that start from LOADED values (C-in, the LSTM projection-table gather) must have landed before the loop is
        // entered: otherwise the waitcnt pass, merging the loop-entry state into the loop header, guards the first MFMAs of EVERY
        // trip with vmcnt(3) .. vmcnt(0) -- which on the back edge drains the staging loads requested five slices earlier, i.e. the
        // prefetch is thrown away every second K-tile (seen in the ISA of lstm_step_mfma_pair).  The values are needed by the
        // first MFMA anyway; waiting here, before K-tile 1 is requested, costs nothing.
        // (unconditional: behind a run-time test the wait would itself sit on one arm of a join)
        __builtin_amdgcn_s_waitcnt(0x0F70);                                                    // vmcnt(0)
        if (nk > 0) {
            swrite(0, S0);
            if constexpr (DEEP) {        // K-tile 1 into set 1, K-tile 2 into set 0 (free again): two tiles in flight from here on
                gload_main((nk > 1 ? 1 : 0) * MBK, S1);
                gload_main((nk > 2 ? 2 : nk - 1) * MBK, S0);
            } else {
                gload_main((nk > 1 ? 1 : 0) * MBK, S0);
            }
        }
        __syncthreads();
        GSTAMP(0)
        if (nk > 0) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) frag_read(0, d, d % RING);
        }
        // two K-tiles per trip, so the LDS buffer of every read and write is a compile-time constant and folds into the
        // instruction's immediate offset (a run-time buffer index cost 10-15 vector adds per K-tile)
        auto ktile = [&](int kt, auto cbc) {
            constexpr int cb = decltype(cbc)::value, nb = cb ^ 1;
#pragma unroll
            for (int k4 = 0; k4 < NS; ++k4) {
                const int sn = k4 + DEPTH;                   // slice whose fragments are fetched now
#ifndef VFR_GEMM_NOSYNC
                if (sn == NS) __syncthreads();               // K-tile kt+1 (written at slice 1) is visible from here on
#endif
                if (sn < NS) frag_read(cb, sn, sn % RING);
                else         frag_read(nb, sn - NS, sn % RING);    // (past the last K-tile: stale data, never used)
                __builtin_amdgcn_sched_barrier(0);
                // VFR_GEMM_SPREAD: the staging stores of K-tile kt + 1 go out ONE behind each MFMA of slices 1 and 2 instead of as a
                // block behind slice 1 (a wave alone on its SIMD issues no MFMA while it moves 5-8 float4 to the LDS: 254 cycles
                // per K-tile of a 32-row tile)
                constexpr int NM = TI * 4, NPIECE = NA + NW;
                static_assert(!VFR_GEMM_SPREAD || NPIECE <= 2 * NM, "staging pieces must fit behind the MFMAs of two slices");
#pragma unroll
                for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj) {
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[k4 % RING][ti], fb[k4 % RING][tj], acc[ti][tj], 0, 0, 0);
                        if (VFR_GEMM_SPREAD && (k4 == 1 || k4 == 2)) {
                            const int piece = (k4 - 1) * NM + ti * 4 + tj;
                            if (piece < NPIECE) swrite_piece(nb, std::integral_constant<int, DEEP ? nb : 0>{}, piece);
                        }
                    }
#if VFR_GEMM_SPREAD == 2
                // (pin the order: MFMA, [the piece's zero-selects], its LDS store -- left alone the scheduler hangs four of the stores
                // behind the first MFMA)
                if (k4 == 1 || k4 == 2) {
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        const int piece = (k4 - 1) * NM + m;
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (piece < NPIECE) {
                            if (((CONV && !CHALO) || (LSTM && !LFAST)) && piece < NA) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                        }
                    }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                // DEEP: K-tile kt + 1 sits in register set nb (= its parity); once it is in LDS that set takes K-tile kt + 3,
                // while set cb still holds K-tile kt + 2 in flight
                if (!VFR_GEMM_SPREAD && k4 == 1) swrite(nb, std::integral_constant<int, DEEP ? nb : 0>{});
                if (k4 == VFR_GLOAD_SLICE) {
                    const int ahead = DEEP ? 3 : 2;
                    const int t2 = kt + ahead < nk ? kt + ahead : nk - 1;
                    gload_main(t2 * MBK, std::integral_constant<int, DEEP ? nb : 0>{});
                }
            }
        };
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            ktile(kt, std::integral_constant<int, 0>{});
            ktile(kt + 1, std::integral_constant<int, 1>{});
        }
        if (kt < nk) ktile(kt, std::integral_constant<int, 0>{});
        if (!CONV && !LSTM && (g.K % MBK)) {   // partial last tile: guarded loads, zero padded (fma(0,0,acc) == acc)
            __syncthreads();
            gload_tail(nk_full * MBK, S0);
            swrite(0, S0);
            __syncthreads();
            compute(0);
        }
    } else {
    init_acc();
    if (nk_full > 0) { gload_main(0, S0); swrite(0, S0); }
    __syncthreads();
    GSTAMP(0)
    for (int kt = 0; kt < nk_full; ++kt) {
        const int nxt = kt + 1 < nk_full ? kt + 1 : nk_full - 1;
        gload_main(nxt * MBK, S0);
        __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ahead of the MFMA block (hipcc sinks it otherwise)
        if (VFR_GEMM_SETPRIO) __builtin_amdgcn_s_setprio(1);
        compute(kt % NBUF);
        if (VFR_GEMM_SETPRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // the other buffer was last read in iteration kt-1 and every wave has passed that iteration's barrier:
        // refill it now (overlapping the other waves' MFMAs); one barrier per K-tile
        if (NBUF == 1) __syncthreads();     // single buffer: everyone must finish reading before the refill
        swrite((kt + 1) % NBUF, S0);
#ifndef VFR_GEMM_NOSYNC
        __syncthreads();
#endif
    }
    if (!CONV && !LSTM && (g.K % MBK)) {   // partial last tile: guarded loads, zero padded (fma(0,0,acc) == acc)
        gload_tail(nk_full * MBK, S0);
        swrite(nk_full % NBUF, S0);
        __syncthreads();
        compute(nk_full % NBUF);
    }
    }
    if (PP && !active) return;
    tail_mfma();

    GSTAMP(1)
    if (LSTM) {
        // tile columns are (gate, unit): a lane holds i, f, g, o of unit 32*blockIdx.y + 16*wn + (lane & 15) in its four column
        // tiles, for 4*TI rows -- every cell is finished by ONE lane, no exchange
        const int H = g.lstm_H, unit = by * 32 + wn * 16 + l15;
        const bool valid = unit < H;
        const int uc = valid ? unit : H - 1;
        if constexpr (!CELL_EARLY) load_cell_inputs();
        // batched and branch-free: the previous-cell loads first (clamped addresses), then 4*TI independent gate evaluations
        // in ONE basic block (the asm pins the results ahead of the predicated stores; hipcc otherwise sinks each cell's ~190
        // instructions into its own store predicate, one serial dependency chain after the other)
        float cnew[TI][4], hnew[TI][4];
        float gsave[TRAIN ? TI : 1][4][4];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ig = c_sigmoidf(acc[ti][0][r] + bi);
                const float fg = c_sigmoidf(acc[ti][1][r] + bf);
                const float gg = c_tanhf(acc[ti][2][r] + bg);
                const float og = c_sigmoidf(acc[ti][3][r] + bo);
                cnew[ti][r] = __builtin_fmaf(fg, cprev[ti][r], ig * gg);
                hnew[ti][r] = og * c_tanhf(cnew[ti][r]);
                asm volatile("" : "+v"(cnew[ti][r]), "+v"(hnew[ti][r]));
                if constexpr (TRAIN) { gsave[ti][r][0] = ig; gsave[ti][r][1] = fg; gsave[ti][r][2] = gg; gsave[ti][r][3] = og; }
            }
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + wm * (16 * TI) + ti * 16 + 4 * lq + r;
                if (valid && row < Mrows) {
                    if constexpr (MULTI) {
                        __hip_atomic_store(g.lstm_c + row * H + unit, cnew[ti][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(g.lstm_h + row * g.lstm_ldh + unit, hnew[ti][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                    g.lstm_c[row * H + unit] = cnew[ti][r];
                    g.lstm_h[row * g.lstm_ldh + unit] = hnew[ti][r];
                    }
                    if constexpr (TRAIN) {
                        float *g4 = g.lstm_gates + row * 4 * H + unit;
                        g4[0] = gsave[ti][r][0]; g4[H] = gsave[ti][r][1]; g4[2 * H] = gsave[ti][r][2]; g4[3 * H] = gsave[ti][r][3];
                    }
                }
            }
        GSTAMP(2)
#ifdef VFR_GEMM_STAMPS
        if (lane == 0) for (int i = 0; i < 3; ++i) atomicAdd(&g_gemm_stamps[i], gst[i]);
        if (tid == 0) atomicAdd(&g_gemm_stamps[3], 1ull);
#endif
        return;
    }
    if (CONV && (g.epi & EPI_POOL2)) {
        // 2x2 max-pool in place of the store of four rows: the quad (rows 4*lq .. 4*lq + 3 of a 16-row block) is one window
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                const int col = n0 + wn * 64 + tj * 16 + l15;
                const int64_t row = m0 + wm * (16 * TI) + ti * 16 + 4 * lq;
                if (col >= g.N || row >= g.M) continue;
                const float badd = (g.epi & EPI_BIAS) ? g.bias[col] : 0.0f;
                float v = acc[ti][tj][0] + badd;
#pragma unroll
                for (int r = 1; r < 4; ++r) { const float x = acc[ti][tj][r] + badd; v = x > v ? x : v; }
                if (g.epi & EPI_RELU) v = v > 0.0f ? v : 0.0f;
                if constexpr (CHALO) g.out[(int64_t)otab_s[row - m0] * g.ldo + col] = v;
                else                 g.out[(row >> 2) * g.ldo + col] = v;
            }
        return;
    }
    // Plain epilogues (bias / residual / ReLU) of 16-byte friendly outputs go through LDS: a lane's accumulator registers are
    // one column x four rows, so direct stores (and residual loads) are 64-byte row segments -- a 1x1 convolution with a short
    // K (ResNet's expansion convolutions: K = 64 .. 512, an output and a residual of 4x the input each) then runs at ~1 TB/s on
    // its epilogue alone.  The tile is parked in the (now idle) staging buffers and leaves as whole rows, 16 bytes per lane:
    // same values, same order of the additions.
    if constexpr (!LSTM && !CHALO && !PP) {
        const bool res = (g.epi & EPI_RES) != 0;
        const bool vec_epi = !(g.epi & (EPI_POOL2 | EPI_VIS)) && ((g.N | (int)g.ldo | (res ? (int)g.ldr : 0)) & 3) == 0 &&
                             ((((uintptr_t)g.out) | (res ? (uintptr_t)g.res : (uintptr_t)0)) & 15) == 0;
        if (vec_epi) {
            constexpr int CLD = BN + 4, C4 = BN / 4;                      // padded row stride (floats); float4 per tile row
            static_assert(TBM * CLD <= NBUF * BUF_FLOATS, "the output tile must fit the staging buffers");
            __syncthreads();                                             // every wave has read its last fragments
            float *ct = lds;
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ct[(wm * (16 * TI) + ti * 16 + 4 * lq + r) * CLD + wn * 64 + tj * 16 + l15] = acc[ti][tj][r];
            __syncthreads();
            for (int idx = tid; idx < TBM * C4; idx += 256) {
                const int row = idx / C4, c4 = idx - row * C4;
                const int64_t grow = m0 + row;
                const int col = n0 + 4 * c4;
                if (grow >= g.M || col >= g.N) continue;
                float4 v = *reinterpret_cast<const float4 *>(ct + row * CLD + 4 * c4);
                if (g.epi & EPI_BIAS2) {
                    v.x = v.x + (g.bias[col] + g.bias2[col]); v.y = v.y + (g.bias[col + 1] + g.bias2[col + 1]);
                    v.z = v.z + (g.bias[col + 2] + g.bias2[col + 2]); v.w = v.w + (g.bias[col + 3] + g.bias2[col + 3]);
                } else if (g.epi & EPI_BIAS) {
                    v.x = v.x + g.bias[col]; v.y = v.y + g.bias[col + 1]; v.z = v.z + g.bias[col + 2]; v.w = v.w + g.bias[col + 3];
                }
                if (res) {
                    const float4 rr = *reinterpret_cast<const float4 *>(g.res + grow * g.ldr + col);
                    v.x = v.x + rr.x; v.y = v.y + rr.y; v.z = v.z + rr.z; v.w = v.w + rr.w;
                }
                if (g.epi & EPI_RELU) { v.x = v.x > 0.0f ? v.x : 0.0f; v.y = v.y > 0.0f ? v.y : 0.0f; v.z = v.z > 0.0f ? v.z : 0.0f; v.w = v.w > 0.0f ? v.w : 0.0f; }
                *reinterpret_cast<float4 *>(g.out + grow * g.ldo + col) = v;
            }
            return;
        }
    }
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            const int col = n0 + wn * 64 + tj * 16 + l15;
            if (col >= g.N) continue;
            float badd = 0.0f;
            if (g.epi & EPI_BIAS2) badd = g.bias[col] + g.bias2[col];
            else if (g.epi & EPI_BIAS) badd = g.bias[col];
            if (g.epi & EPI_VIS) {
                // hidden layer of the clip encoder fused here: the S array never exists in HBM.  Loads first (clamped
                // rows, no branches), then the arithmetic, then predicated stores.
                const float w0 = g.vis_w0[col], w1 = g.vis_w1[col], b = g.bias[col];
                float cx[4], t0[4], t1[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = m0 + wm * (16 * TI) + ti * 16 + 4 * lq + r;
                    const int64_t rc = row < g.M ? row : g.M - 1;
                    t0[r] = g.vis_te[2 * rc]; t1[r] = g.vis_te[2 * rc + 1];
                    cx[r] = g.vis_cx[(int64_t)g.vis_row[rc] * g.N + col];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = m0 + wm * (16 * TI) + ti * 16 + 4 * lq + r;
                    const float te = __builtin_fmaf(t1[r], w1, __builtin_fmaf(t0[r], w0, 0.0f));
                    const float x = ((acc[ti][tj][r] + cx[r]) + te) + b;
                    if (row < g.M) g.out[row * g.ldo + col] = x > 0.0f ? x : 0.0f;
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + wm * (16 * TI) + ti * 16 + 4 * lq + r;
                if (row >= g.M) continue;
                float v = acc[ti][tj][r];
                if (g.epi & (EPI_BIAS | EPI_BIAS2)) v = v + badd;
                if (g.epi & EPI_RES) v = v + g.res[row * g.ldr + col];
                if (g.epi & EPI_RELU) v = v > 0.0f ? v : 0.0f;
                if constexpr (CHALO) g.out[(int64_t)otab_s[row - m0] * g.ldo + col] = v;
                else                 g.out[row * g.ldo + col] = v;
            }
        }
}

template <bool VEC, int MI = 2>
__global__ __launch_bounds__(256, (VFR_GEMM_NBUF == 1 || MI <= 1) ? 3 : 2) void gemm_nt_mfma(GemmArgs g) { gemm_nt_mfma_body<VEC, false, false, MI>(g); }

// dense operands with N <= 64 (ResNet's stem and layer1 1x1 convolutions): 128 x 64 tiles, no MFMA spent on absent columns
__global__ __launch_bounds__(256, 3) void gemm_nt_mfma_narrow(GemmArgs g) { gemm_nt_mfma_body<true, false, false, 1, false, true>(g); }

template <bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_nt_mfma_pair(GemmPair gp)
{
    const GemmArgs g = gp.p[blockIdx.z];      // a private copy: fields read inside the K loop must not be re-fetched from the
    gemm_nt_mfma_body<VEC, false>(g);         // kernarg array there (an s_load + lgkmcnt(0) would also drain the LDS reads)
}

template <int MI = 2, bool NARROW = false, bool CFAST = false, bool CHALO = false>
__global__ __launch_bounds__(256, MI <= 1 ? 3 : 2) void conv3x3_nhwc_mfma(GemmArgs g) { gemm_nt_mfma_body<true, true, false, MI, false, NARROW, false, CFAST, false, CHALO>(g); }

template <int MI, bool LFAST = false>
__global__ __launch_bounds__(256, MI <= 1 ? 3 : VFR_LSTM_WAVES) void lstm_step_mfma_pair(GemmPair gp)
{
    const GemmArgs g = gp.p[blockIdx.z];      // private copy, see gemm_nt_mfma_pair
    gemm_nt_mfma_body<true, false, true, MI, false, false, false, false, LFAST>(g);
}
// ---- all T steps of both directions in ONE launch ------------------------------------------------------------------------
// One launch per time step ends every step with a partial round of workgroups (3.3 .. 6.6 rounds of the chip's 768 slots per
// step at 5 000 queries: +-5 % per step in profiles/r4t_lstm_steps.txt; 1.1 rounds at a rank's 626 rows: half the chip idle
// half the time) and starts the next one 6 us later with every CU in its prologue at once.  Here the (step, direction, row
// tile, column tile) tasks of the whole sequence are ONE ordered list per XCD group (step-major; inside a step the order of
// the per-step launch: forward row tiles, then the reverse direction's active ones, the group's column tiles fastest -- its
// slice of W_hh stays in its L2), and 3 workgroups per CU draw tickets from their group's counter until the list is empty.
// A task of step s needs h / c of step s - 1 for ITS rows (all column tiles of the same row tile: the K dimension) and, in the
// reverse direction, for row tile 0 (the shared all-pad row); it waits until the completion counters of those row tiles
// read `ncol`.  The state rotates through THREE buffers (step s reads s % 3, writes (s + 1) % 3), and a task also waits for
// ALL of step s - 2 (whose reads are the last users of the buffer it writes).
// No deadlock, whatever is resident: every dependency points to an earlier step, every group's list is sorted by step, and
// a ticket is only ever held by a workgroup that is running -- so the earliest unfinished task of the whole launch is always
// either running or next in line for a running workgroup.  (Nothing here needs the grid to be co-resident: unlike the
// register-resident sequence kernels a workgroup never waits for one that has not started.)  The waits are bounded all
// the same (a bug or a lost workgroup must not hang the device): on expiry the error word is raised, every workgroup leaves,
// and the rescue kernel behind the launch (bilstm.hip) re-encodes the batch.
// Coherence: the producer of a row's h ran on another XCD, whose L2 this one does not snoop -- see MULTI in the kernel body.
struct LstmSteps {
    GemmArgs p[2];                 // step-invariant arguments per direction (W2 / ldw2 / biases / Cin = the projection table / M = rows ...)
    float *hbuf[3], *cbuf[3];      // rotating state: h [R, 2H] (direction d at column d*H), c [2][R, H]; buffer 0 zeroed
    const int *tokidx;             // [T][R] projection-table row of GEMM row m at time t
    const int *mcount;             // reverse direction: active rows per step
    unsigned *sync;                // [8 tickets | error word | pad to 16 | T step counters | T x 2 x rt_max row-tile counters], zeroed
    int T, cpx, ncol, rt_max;
    unsigned max_spins;
    int fault_task;                // TEST HOOK (-1: off): this ticket of group 0 never signals its completion
};
#ifdef VFR_STEPS_STAMPS
__device__ unsigned long long g_steps_stamps[8];
__device__ unsigned long long g_steps_waits[32];
#define SSTAMP(i) { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sst[i] += t_ - st0; st0 = t_; } }
#else
#define SSTAMP(i)
#endif
template <int MI>
__global__ __launch_bounds__(256, MI <= 1 ? 3 : 2) void lstm_steps_mfma_kernel(LstmSteps a)
{
#ifdef VFR_STEPS_STAMPS
    unsigned long long sst[6] = {0, 0, 0, 0, 0, 0}, st0 = __builtin_amdgcn_s_memtime();
#endif
    constexpr int TBM = MI ? 64 * MI : 32;
    __shared__ unsigned s_next;
    __shared__ int s_dead;
    const int tid = threadIdx.x;
    const unsigned x = blockIdx.x & 7u;
    unsigned *tick = a.sync + x, *err = a.sync + 8, *stepdone = a.sync + 16, *done = a.sync + 16 + a.T;
    const int R = (int)a.p[0].M, rtf = (R + TBM - 1) / TBM, cpx = a.cpx, T = a.T;
    // (mcount is written by an earlier kernel; loaded through the vector path here, its values are pinned to SGPRs: everything
    // below -- the step, the tile, every pointer of the task -- must stay scalar, or the body's buffer loads are wrapped in
    // waterfall loops and its K-tile counter lives in a VGPR)
    auto rtr_of = [&](int s) { return __builtin_amdgcn_readfirstlane((a.mcount[s] + TBM - 1) / TBM); };
    // position in the group's list: step `cur`, whose tasks are tickets [base, base + ncur)
    int cur = 0;
    unsigned base = 0, ncur = (unsigned)((rtf + rtr_of(0)) * cpx);
    auto advance = [&](unsigned n, int &c, unsigned &b, unsigned &nc) {      // -> the step ticket n belongs to (c == T: past the end)
        while (c < T && n >= b + nc) {
            b += nc; ++c;
            if (c < T) nc = (unsigned)((rtf + rtr_of(c)) * cpx);
        }
        c = __builtin_amdgcn_readfirstlane(c);
        b = __builtin_amdgcn_readfirstlane(b);
        nc = __builtin_amdgcn_readfirstlane(nc);
    };
    struct Deps { const unsigned *p[3]; unsigned want[3]; };
    // what ticket n of step c waits for: its row tile and (reverse direction) row tile 0 at step c - 1, all of step c - 2
    auto deps_of = [&](unsigned n, int c, unsigned b) -> Deps {
        Deps q{{nullptr, nullptr, nullptr}, {0u, 0u, 0u}};
        unsigned j = n - b;
        const int d = j >= (unsigned)(rtf * cpx) ? 1 : 0;
        if (d) j -= (unsigned)(rtf * cpx);
        const int rt = (int)(j / (unsigned)cpx), by = (int)(x * (unsigned)cpx + j % (unsigned)cpx);
        if (by >= a.ncol || c == 0) return q;
        const unsigned *dprev = done + ((size_t)(c - 1) * 2 + d) * a.rt_max;
        if (rt < (d ? rtr_of(c - 1) : rtf)) { q.p[0] = dprev + rt; q.want[0] = (unsigned)a.ncol; }
        if (d && rt != 0) { q.p[1] = dprev; q.want[1] = (unsigned)a.ncol; }
        if (c >= 2) { q.p[2] = stepdone + (c - 2); q.want[2] = (unsigned)((rtf + rtr_of(c - 2)) * a.ncol); }
        return q;
    };
    auto wait_ge = [&](const unsigned *ptr, unsigned target) -> bool {
        unsigned spins = 0;
        while (__hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 63u) == 0u && (spins > a.max_spins || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) return false;
        }
        return true;
    };
    if (tid == 0) {
        // the first ticket, its dependencies the slow way
        const unsigned n0 = atomicAdd(tick, 1u);
        int c = 0; unsigned b = 0, nc = ncur;
        advance(n0, c, b, nc);
        bool ok = true;
        if (c < T) {
            const Deps q = deps_of(n0, c, b);
            for (int i = 0; i < 3; ++i) if (ok && q.p[i]) ok = wait_ge(q.p[i], q.want[i]);
        }
        if (!ok) atomicOr(err, 1u);
        s_dead = ok ? 0 : 1;
        s_next = n0;
    }
    __syncthreads();
    unsigned n = __builtin_amdgcn_readfirstlane(s_next);
    for (;;) {
        SSTAMP(0)
        if (__builtin_amdgcn_readfirstlane(s_dead)) break;
        advance(n, cur, base, ncur);
        if (cur >= T) break;
        unsigned j = n - base;
        const int d = j >= (unsigned)(rtf * cpx) ? 1 : 0;
        if (d) j -= (unsigned)(rtf * cpx);
        const int rt = (int)(j / (unsigned)cpx), by = (int)(x * (unsigned)cpx + j % (unsigned)cpx);
        const bool valid = by < a.ncol;
        SSTAMP(1)
        if (valid) {
            GemmArgs g = a.p[d];
            const int t = d ? T - 1 - cur : cur;
            float *hin = a.hbuf[cur % 3], *hout = a.hbuf[(cur + 1) % 3];
            const int H = g.lstm_H;
            g.A2 = hin + (size_t)d * H; g.A = g.A2;
            g.lstm_h = hout + (size_t)d * H;
            g.lstm_cin = a.cbuf[cur % 3] + (size_t)d * R * H;
            g.lstm_c = a.cbuf[(cur + 1) % 3] + (size_t)d * R * H;
            g.lstm_tok = a.tokidx + (size_t)t * R;
            g.lstm_step = cur;
            g.K2 = cur == 0 ? 0 : H;                 // h_0 = 0: the first step is its table gather + gate epilogue (see bilstm.hip)
            // (the thread index is made opaque per task: everything the body derives from it would otherwise be hoisted out of the
            // task loop and held in registers across it -- 40 more VGPRs than the per-step kernel, spilled at three workgroups per CU)
            int tid_task = tid;
            asm volatile("" : "+v"(tid_task));
            gemm_nt_mfma_body<true, false, true, MI, false, false, false, false, true, false, true>(g, by, (int64_t)rt * TBM, tid_task);
        }
        SSTAMP(2)
        // The next ticket is drawn only now (a ticket drawn before the task would sit idle for the task's whole duration, and the
        // same row tile of the next step, on all eight groups, with it), and the next task's dependency counters are requested
        // BEFORE this task's stores are drained: both round trips ride on the drain (read one after the other behind the
        // barrier, three dependent round trips cost 7 % of a 5 000-query pass).
        Deps q{{nullptr, nullptr, nullptr}, {0u, 0u, 0u}};
        unsigned seen[3] = {0u, 0u, 0u};
        int c2 = cur;
        unsigned nxt = 0;
        if (tid == 0) {
            nxt = atomicAdd(tick, 1u);
            unsigned b2 = base, nc2 = ncur;
            advance(nxt, c2, b2, nc2);
            if (c2 < T) {
                q = deps_of(nxt, c2, b2);
#pragma unroll
                for (int i = 0; i < 3; ++i) if (q.p[i]) seen[i] = __hip_atomic_load(q.p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): this wave's write-through stores of h / c have been acknowledged
        __syncthreads();
        SSTAMP(3)
        if (tid == 0) {
            if (valid && !(a.fault_task >= 0 && x == 0u && n == (unsigned)a.fault_task)) {
                __hip_atomic_fetch_add(done + ((size_t)cur * 2 + d) * a.rt_max + rt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(stepdone + cur, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            bool ok = true;
            if (c2 < T) {
#pragma unroll
                for (int i = 0; i < 3; ++i) if (ok && q.p[i] && seen[i] < q.want[i]) {
#ifdef VFR_STEPS_STAMPS
                    const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
                    ok = wait_ge(q.p[i], q.want[i]);
#ifdef VFR_STEPS_STAMPS
                    atomicAdd(&g_steps_waits[i], 1ull); atomicAdd(&g_steps_waits[3 + i], __builtin_amdgcn_s_memtime() - w0);
                    if (c2 < 20) atomicAdd(&g_steps_waits[8 + c2], 1ull);
#endif
                }
                if (!ok) atomicOr(err, 1u);
            }
            s_dead = ok ? 0 : 1;
            s_next = nxt;
        }
        SSTAMP(4)
        __syncthreads();                             // the next task's dependencies are met (and every wave has left this task's LDS tiles)
        n = __builtin_amdgcn_readfirstlane(s_next);
        SSTAMP(5)
    }
#ifdef VFR_STEPS_STAMPS
    if (tid == 0) { for (int i = 0; i < 6; ++i) atomicAdd(&g_steps_stamps[i], sst[i]); atomicAdd(&g_steps_stamps[6], 1ull); }
#endif
}

template <int MI>
__global__ __launch_bounds__(256, MI <= 1 ? 3 : 2) void lstm_step_train_mfma_pair(GemmPair gp)
{
    const GemmArgs g = gp.p[blockIdx.z];
    gemm_nt_mfma_body<true, false, true, MI, false, false, true>(g);
}
// blockIdx.z = problem * nsplit + K range: 32-row tiles over one K range of one problem (gemm_nt_splitk_pair)
__global__ __launch_bounds__(256, 3) void gemm_nt_mfma_splitk_pair(GemmPair gp, int nsplit, int64_t out_stride)
{
    const int z = blockIdx.z, d = z / nsplit, j = z - d * nsplit;
    GemmArgs g = gp.p[d];
    const int kc = g.K / nsplit;
    g.A += (int64_t)j * kc; g.W += (int64_t)j * kc; g.out += (int64_t)j * out_stride; g.K = kc;
    gemm_nt_mfma_body<true, false, false, 0>(g);
}
__global__ __launch_bounds__(512, 2) void lstm_step_mfma_pair_pp(GemmPair gp)
{
    const GemmArgs g = gp.p[blockIdx.z];
    gemm_nt_mfma_body<true, false, true, 2, true>(g);
}
template <bool VEC>
__global__ __launch_bounds__(512, 2) void gemm_nt_mfma_pp(GemmArgs g) { gemm_nt_mfma_body<VEC, false, false, 2, true>(g); }

__global__ __launch_bounds__(256) void repack_rows_kernel(const float *__restrict__ src, int64_t ld_src, int rows, int cols,
                                                          float *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    dst[i] = src[r * ld_src + c];
}

// W [4H, E] -> the fused step's tile-column order [128 * ceil(H/32), E] (zero rows for units >= H): tile column
// c = 128*tile + rr holds gate (rr>>4)&3 of unit 32*tile + 16*(rr>>6) + (rr&15)
__global__ __launch_bounds__(256) void lstm_permute_rows_kernel(const float *__restrict__ W, int H, int E, int ncols,
                                                                float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)ncols * E) return;
    const int c = (int)(i / E), k = (int)(i - (int64_t)c * E);
    const int tile = c >> 7, rr = c & 127;
    const int gate = (rr >> 4) & 3, unit = tile * 32 + (rr >> 6) * 16 + (rr & 15);
    out[i] = unit < H ? W[((int64_t)gate * H + unit) * E + k] : 0.0f;
}

int lstm_permute_rows(const float *W, int H, int E, float *out, hipStream_t st)
{
    const int ncols = (int)cdiv(H, 32) * MBN;
    ProfScope prof(SITE_REPACK, st);
    hipLaunchKernelGGL(lstm_permute_rows_kernel, dim3((unsigned)cdiv((int64_t)ncols * E, 256)), dim3(256), 0, st, W, H, E, ncols, out);
    VFR_CHECK_LAUNCH("lstm_permute_rows_kernel");
    return VFR_OK;
}

int repack_rows(const float *src, int64_t ld_src, int rows, int cols, float *dst, hipStream_t st)
{
    if (rows <= 0 || cols <= 0) return VFR_OK;
    ProfScope prof(SITE_REPACK, st);
    hipLaunchKernelGGL(repack_rows_kernel, dim3((unsigned)cdiv((int64_t)rows * cols, 256)), dim3(256), 0, st, src, ld_src, rows,
                       cols, dst);
    VFR_CHECK_LAUNCH("repack_rows_kernel");
    return VFR_OK;
}

static bool gemm_vec_ok(const GemmArgs &g)
{
    return ((g.lda | g.ldw) & 3) == 0 && ((((uintptr_t)g.A) | ((uintptr_t)g.W)) & 15) == 0;
}

int lstm_step_pair(const GemmArgs &g0, const GemmArgs &g1, hipStream_t st)
{
    if (g0.M == 0) return VFR_OK;
    for (const GemmArgs *g : {&g0, &g1}) {
        VFR_REQUIRE(g->A && g->W && g->A2 && g->W2 && g->bias && g->bias2 && g->lstm_c && g->lstm_cin && g->lstm_h && g->lstm_H > 0, VFR_EINVAL,
                    "lstm_step_pair: bad argument");
        VFR_REQUIRE(!g->lstm_tok || (g->Cin && g->K == 0 && g->ldc >= cdiv(g->lstm_H, 32) * MBN), VFR_EINVAL,
                    "lstm_step_pair: the input-projection table needs Cin [vocab, >= 128*ceil(H/32)] and K == 0");
        VFR_REQUIRE(((g->lda | g->ldw | g->lda2 | g->ldw2 | g->K | g->K2) & 3) == 0 &&
                        ((((uintptr_t)g->A) | ((uintptr_t)g->W) | ((uintptr_t)g->A2) | ((uintptr_t)g->W2)) & 15) == 0,
                    VFR_EUNSUPPORTED, "lstm_step_pair: E, H and the operand strides must be multiples of 4 floats, 16-byte aligned");
        // the kernel addresses staged rows with 32-bit byte offsets from the segment pointers
        const int64_t lda_max = g->lda > g->lda2 ? g->lda : g->lda2, ldw_max = g->ldw > g->ldw2 ? g->ldw : g->ldw2;
        VFR_REQUIRE((g->M * lda_max + MBK) * 4 < (1ll << 32) && (4ll * g->lstm_H * ldw_max + MBK) * 4 < (1ll << 32), VFR_EUNSUPPORTED,
                    "lstm_step_pair: operand larger than 4 GB (rows x leading dimension); split the batch");
    }
    ProfScope prof(g0.site, st);
    GemmPair gp{{g0, g1}};
    // 64-row tiles by default: a step is only a few rounds of the chip's 512 workgroup slots, a partial last round of
    // 128-row tiles costs as much as a full one (a workgroup alone on a CU is not twice as fast), and half-size tiles halve that
    // loss -- measured faster at every batch size from 625 to 5000 queries (lstm_tile = 2 forces 128 rows).  Below ~700
    // rows (a rank's share of the batch on 8 GPUs) the step is hardly more than one round and 32-row tiles at three
    // workgroups per CU balance the CUs better still (tools/lstm_tile_sweep.py: -1 % at 625 rows, -11 % at 313; alone the
    // encoder also gains 1-2 % at 1250 / 2500 rows, but beside the clip encoder on the other stream it does not; +5 % at 5000).
    const int tile = opt_lstm_tile() ? opt_lstm_tile() : (g0.M <= 700 ? 3 : 1);
    const int ncol = (int)cdiv(g0.lstm_H, 32), cpx = (int)cdiv(ncol, 8);
    const bool xcd = opt_lstm_xcd() && !(tile == 2 && opt_gemm_pp());
    if (xcd) gp.p[0].xcd_cols = gp.p[1].xcd_cols = cpx;
    auto grid_for = [&](int rows) {
        const unsigned rt = (unsigned)cdiv(g0.M, rows);
        return xcd ? dim3(rt * (unsigned)cpx * 8u, 1, 2) : dim3(rt, (unsigned)ncol, 2);
    };
    // the table-start step with a recurrent segment of whole K-tiles (+ a remainder the kernel takes as direct fragments): the
    // instantiation without selects in its K-loop (same chains)
    auto fast_ok = [&](const GemmArgs &g) {
        const int rem = g.K2 % MBK;
        return g.lstm_tok && g.K == 0 && g.K2 >= MBK && (rem == 0 || (g.K2 > MBK && rem <= 8));
    };
    const bool lfast = opt_lstm_fast() && fast_ok(g0) && fast_ok(g1) && VFR_GEMM_PIPE && VFR_LSTM_NBUF == 2;
    if (g0.lstm_gates) {                     // training forward: the same step, gates stored as well (32- or 64-row tiles)
        VFR_REQUIRE(g1.lstm_gates, VFR_EINVAL, "lstm_step_pair: gates buffer for one direction only");
        if (tile == 3) hipLaunchKernelGGL(lstm_step_train_mfma_pair<0>, grid_for(32), dim3(256), 0, st, gp);
        else           hipLaunchKernelGGL(lstm_step_train_mfma_pair<1>, grid_for(64), dim3(256), 0, st, gp);
    } else if (tile == 3) {                  // 32-row tiles, three workgroups per CU
        if (lfast) hipLaunchKernelGGL((lstm_step_mfma_pair<0, true>), grid_for(32), dim3(256), 0, st, gp);
        else       hipLaunchKernelGGL(lstm_step_mfma_pair<0>, grid_for(32), dim3(256), 0, st, gp);
    } else if (tile != 2) {
        if (lfast) hipLaunchKernelGGL((lstm_step_mfma_pair<1, true>), grid_for(64), dim3(256), 0, st, gp);
        else       hipLaunchKernelGGL(lstm_step_mfma_pair<1>, grid_for(64), dim3(256), 0, st, gp);
    } else if (opt_gemm_pp()) {
        dim3 grid((unsigned)cdiv(cdiv(g0.M, MBM), 2), (unsigned)cdiv(g0.lstm_H, 32), 2);
        hipLaunchKernelGGL(lstm_step_mfma_pair_pp, grid, dim3(512), 0, st, gp);
    } else {
        hipLaunchKernelGGL(lstm_step_mfma_pair<2>, grid_for(MBM), dim3(256), 0, st, gp);
    }
    VFR_CHECK_LAUNCH("lstm_step_mfma_pair");
#ifdef VFR_GEMM_STAMPS
    if (g0.lstm_step == 19 || g0.lstm_step == 0) {
        unsigned long long h[4];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_stamps), sizeof h);
        const double tot = (double)h[0] + h[1] + h[2];
        fprintf(stderr, "[gemm stamps] step %d: tiles so far %llu  prologue %.1f%%  main loop %.1f%%  epilogue %.1f%%  (%.0f ticks per wave-tile)\n",
                g0.lstm_step, h[3], 100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, tot / (4.0 * h[3]));
    }
#endif
    return VFR_OK;
}

// all T steps in one launch (lstm_steps_mfma_kernel); g0 / g1 carry the step-invariant arguments of the two directions
bool lstm_steps_supported(int64_t R, int H)
{
    const int rem = H % MBK;
    // (the cross-check switches of the per-step launch -- general K-loop, first step with its recurrent part, launch order, 128-row
    // tiles -- select the per-step path)
    return opt_lstm_multi() && opt_lstm_fast() && opt_lstm_skip0() && opt_lstm_xcd() && bool(VFR_GEMM_PIPE) && VFR_LSTM_NBUF == 2 && (H % 4) == 0 && H >= MBK && (rem == 0 || (H > MBK && rem <= 8)) &&
           ((R + 1) * 2 * (int64_t)H + MBK) * 4 < (1ll << 32);
}
size_t lstm_steps_sync_words(int64_t R, int T) { return 16 + (size_t)T + (size_t)T * 2 * (size_t)cdiv(R, 32); }
int lstm_steps_run(const GemmArgs &g0, const GemmArgs &g1, float *const hbuf[3], float *const cbuf[3], const int *tokidx, const int *mcount,
                   unsigned *sync, int T, hipStream_t st)
{
    if (g0.M == 0 || T == 0) return VFR_OK;
    for (const GemmArgs *g : {&g0, &g1}) {
        VFR_REQUIRE(g->W2 && g->bias && g->bias2 && g->Cin && g->lstm_H > 0 && tokidx && mcount && sync, VFR_EINVAL, "lstm_steps_run: bad argument");
        VFR_REQUIRE(lstm_steps_supported(g->M, g->lstm_H), VFR_EUNSUPPORTED, "lstm_steps_run: shape not supported (use the per-step launches)");
        VFR_REQUIRE(((g->lda2 | g->ldw2) & 3) == 0 && (((uintptr_t)g->W2) & 15) == 0 && (4ll * g->lstm_H * g->ldw2 + MBK) * 4 < (1ll << 32), VFR_EUNSUPPORTED,
                    "lstm_steps_run: operand strides must be multiples of 4 floats, 16-byte aligned, under 4 GB");
    }
    ProfScope prof(g0.site, st);
    LstmSteps a{};
    a.p[0] = g0; a.p[1] = g1;
    for (int d = 0; d < 2; ++d) { a.p[d].K = 0; a.p[d].W = a.p[d].W2; a.p[d].ldw = a.p[d].ldw2; a.p[d].lda = a.p[d].lda2; a.p[d].xcd_cols = 0; }
    for (int i = 0; i < 3; ++i) { a.hbuf[i] = hbuf[i]; a.cbuf[i] = cbuf[i]; }
    a.tokidx = tokidx; a.mcount = mcount; a.sync = sync; a.T = T;
    a.ncol = (int)cdiv(g0.lstm_H, 32); a.cpx = (int)cdiv(a.ncol, 8);
    const int tile = opt_lstm_tile() == 3 || (opt_lstm_tile() == 0 && g0.M <= 700) ? 3 : opt_lstm_tile() == 2 ? 2 : 1;      // 32-row tiles for small batches (as lstm_step_pair)
    a.rt_max = (int)cdiv(g0.M, 32);
    a.fault_task = opt_lstm_persist_fault();
    a.max_spins = a.fault_task >= 0 ? 1u << 15 : 1u << 22;       // x (~1 us per poll): seconds -- only ever reached through a bug or a lost workgroup
    const int64_t rows = tile == 3 ? 32 : tile == 2 ? 128 : 64;
    const int64_t per_step_max = 2 * cdiv(g0.M, rows) * a.cpx * 8;
    int64_t slots = (int64_t)device_cu_count() * (tile == 2 ? 2 : 3) / 8 * 8;
    if (slots > per_step_max * T) slots = per_step_max * T;
    if (slots < 8) slots = 8;
    if (tile == 3)      hipLaunchKernelGGL(lstm_steps_mfma_kernel<0>, dim3((unsigned)slots), dim3(256), 0, st, a);
    else if (tile == 2) hipLaunchKernelGGL(lstm_steps_mfma_kernel<2>, dim3((unsigned)slots), dim3(256), 0, st, a);
    else                hipLaunchKernelGGL(lstm_steps_mfma_kernel<1>, dim3((unsigned)slots), dim3(256), 0, st, a);
    VFR_CHECK_LAUNCH("lstm_steps_mfma_kernel");
#if defined(VFR_STEPS_STAMPS) || defined(VFR_GEMM_STAMPS)      /* timing builds only (tools/build_variant.sh): counters and phase stamps of the launch */
    if (getenv("VFR_STEPS_DEBUG")) {
        const size_t nw = lstm_steps_sync_words(g0.M, T);
        std::vector<unsigned> h(nw);
        hipError_t e = hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), sync, nw * 4, hipMemcpyDeviceToHost);
        fprintf(stderr, "[steps debug] sync err %d (%s) grid %lld tile %d R %lld T %d ncol %d cpx %d rt_max %d\n  tickets:", (int)e, hipGetErrorString(e), (long long)slots, tile, (long long)g0.M, T, a.ncol, a.cpx, a.rt_max);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %u", h[i]);
        fprintf(stderr, "\n  err %u\n  stepdone:", h[8]);
        for (int i = 0; i < T; ++i) fprintf(stderr, " %u", h[16 + i]);
        fprintf(stderr, "\n  done[0][0][..8]:");
        for (int i = 0; i < 8 && i < a.rt_max; ++i) fprintf(stderr, " %u", h[16 + T + i]);
        fprintf(stderr, "\n");
#ifdef VFR_GEMM_STAMPS
        {
            unsigned long long h4[4], z4[4] = {0, 0, 0, 0};
            (void)hipMemcpyFromSymbol(h4, HIP_SYMBOL(g_gemm_stamps), sizeof h4);
            const double tot4 = (double)h4[0] + h4[1] + h4[2];
            fprintf(stderr, "  body stamps: tiles %llu  prologue %.1f%%  main loop %.1f%%  epilogue %.1f%%  (%.0f ticks per wave-tile)\n",
                    h4[3], 100 * h4[0] / tot4, 100 * h4[1] / tot4, 100 * h4[2] / tot4, tot4 / (4.0 * h4[3]));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), z4, sizeof z4);
        }
#endif
#ifdef VFR_STEPS_STAMPS
        unsigned long long hs[8];
        (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_steps_stamps), sizeof hs);
        double tot = 0; for (int i = 0; i < 6; ++i) tot += (double)hs[i];
        fprintf(stderr, "  stamps (thread 0 of %llu workgroups, 100 MHz ticks): loop top %.1f%%  decode %.1f%%  body %.1f%%  dep request + drain + barrier %.1f%%  signal + dep wait %.1f%%  barrier %.1f%%   total %.0f ticks per workgroup\n",
                hs[6], 100 * hs[0] / tot, 100 * hs[1] / tot, 100 * hs[2] / tot, 100 * hs[3] / tot, 100 * hs[4] / tot, 100 * hs[5] / tot, tot / (double)hs[6]);
        unsigned long long hw[32], zw[32] = {};
        (void)hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_steps_waits), sizeof hw);
        fprintf(stderr, "  slow-path waits: row tile %llu (%.0f ticks each)  row tile 0 %llu (%.0f)  step-2 %llu (%.0f); by step:", hw[0], hw[0] ? (double)hw[3] / hw[0] : 0.0,
                hw[1], hw[1] ? (double)hw[4] / hw[1] : 0.0, hw[2], hw[2] ? (double)hw[5] / hw[2] : 0.0);
        for (int i = 0; i < 20; ++i) fprintf(stderr, " %llu", hw[8 + i]);
        fprintf(stderr, "\n");
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_steps_waits), zw, sizeof zw);
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_steps_stamps), z, sizeof z);
#endif
    }
#endif
    return VFR_OK;
}

int gemm_nt_pair(const GemmArgs &g0, const GemmArgs &g1, hipStream_t st)
{
    if (opt_gemm() == 0 || g0.M != g1.M || g0.N != g1.N || g0.K != g1.K || g0.epi != g1.epi) {
        if (int rc = gemm_nt(g0, st)) return rc;
        return gemm_nt(g1, st);
    }
    if (g0.M == 0 || g0.N == 0) return VFR_OK;
    VFR_REQUIRE(g0.A && g0.W && g0.out && g1.A && g1.W && g1.out, VFR_EINVAL, "gemm_nt_pair: bad argument");
    VFR_REQUIRE(g0.lda < (1ll << 22) && g0.ldw < (1ll << 22) && g1.lda < (1ll << 22) && g1.ldw < (1ll << 22), VFR_EUNSUPPORTED,
                "gemm_nt_pair: leading dimension of 4M floats or more");
    ProfScope prof(g0.site, st);
    GemmPair gp{{g0, g1}};
    dim3 grid((unsigned)cdiv(g0.M, MBM), (unsigned)cdiv(g0.N, MBN), 2);
    if (gemm_vec_ok(g0) && gemm_vec_ok(g1)) hipLaunchKernelGGL(gemm_nt_mfma_pair<true>, grid, dim3(256), 0, st, gp);
    else                                    hipLaunchKernelGGL(gemm_nt_mfma_pair<false>, grid, dim3(256), 0, st, gp);
    VFR_CHECK_LAUNCH("gemm_nt_mfma_pair");
    return VFR_OK;
}

int gemm_nt_splitk_pair(const GemmArgs &g0, const GemmArgs &g1, int nsplit, int64_t out_stride, hipStream_t st)
{
    if (g0.M == 0 || g0.N == 0) return VFR_OK;
    VFR_REQUIRE(g0.A && g0.W && g0.out && g1.A && g1.W && g1.out && g0.M == g1.M && g0.N == g1.N && g0.K == g1.K && nsplit >= 1 &&
                    nsplit <= 16 && g0.epi == EPI_NONE && g1.epi == EPI_NONE && !g0.Cin && !g1.Cin, VFR_EINVAL,
                "gemm_nt_splitk_pair: bad argument");
    VFR_REQUIRE(g0.lda < (1ll << 22) && g0.ldw < (1ll << 22) && g1.lda < (1ll << 22) && g1.ldw < (1ll << 22), VFR_EUNSUPPORTED,
                "gemm_nt_splitk_pair: leading dimension of 4M floats or more");
    VFR_REQUIRE(g0.K % (4 * nsplit) == 0 && gemm_vec_ok(g0) && gemm_vec_ok(g1), VFR_EUNSUPPORTED,
                "gemm_nt_splitk_pair: K must be a multiple of 4 * nsplit, operands 16-byte aligned with strides in whole float4");
    ProfScope prof(g0.site, st);
    GemmPair gp{{g0, g1}};
    dim3 grid((unsigned)cdiv(g0.M, 32), (unsigned)cdiv(g0.N, MBN), (unsigned)(2 * nsplit));
    hipLaunchKernelGGL(gemm_nt_mfma_splitk_pair, grid, dim3(256), 0, st, gp, nsplit, out_stride);
    VFR_CHECK_LAUNCH("gemm_nt_mfma_splitk_pair");
    return VFR_OK;
}

int gemm_nt(const GemmArgs &g, hipStream_t st)
{
    if (g.M == 0 || g.N == 0) return VFR_OK;
    VFR_REQUIRE(g.A && g.W && g.out && g.M > 0 && g.N > 0 && g.K >= 0, VFR_EINVAL, "gemm_nt: bad argument");
    VFR_REQUIRE(!(g.epi & (EPI_BIAS | EPI_BIAS2)) || g.bias, VFR_EINVAL, "gemm_nt: bias flag without bias");
    VFR_REQUIRE(!(g.epi & EPI_BIAS2) || g.bias2, VFR_EINVAL, "gemm_nt: bias2 flag without bias2");
    VFR_REQUIRE(!(g.epi & EPI_RES) || g.res, VFR_EINVAL, "gemm_nt: residual flag without res");
    VFR_REQUIRE(g.lda < (1ll << 22) && g.ldw < (1ll << 22), VFR_EUNSUPPORTED, "gemm_nt: leading dimension of 4M floats or more");
    ProfScope prof(g.site, st);
    dim3 grid((unsigned)cdiv(g.M, MBM), (unsigned)cdiv(g.N, MBN));
    if (g.conv_cin > 0) {                      // implicit-GEMM convolution: MFMA kernel only
        VFR_REQUIRE((g.conv_cin & 3) == 0 && g.K == 9 * g.conv_cin && (g.ldw & 3) == 0 &&
                        ((((uintptr_t)g.A) | ((uintptr_t)g.W)) & 15) == 0,
                    VFR_EINVAL, "gemm_nt(conv): needs Cin %% 4 == 0, K = 9*Cin and 16-byte aligned operands");
        // C_in a multiple of 32 (every VGG layer but the first): the loader with scalar tap arithmetic
        VFR_REQUIRE(!(g.epi & EPI_POOL2) || ((g.conv_h | g.conv_w) & 1) == 0, VFR_EINVAL, "gemm_nt(conv): the fused 2x2 pool needs even height and width");
        VFR_REQUIRE(!(g.epi & EPI_POOL2) || !(g.epi & (EPI_RES | EPI_VIS | EPI_BIAS2)), VFR_EINVAL, "gemm_nt(conv): fused pool with bias / ReLU only");
        // (the offsets of a tile's rows from its first pixel: 128 consecutive pixels, or with the fused pool 32 windows that may
        // wrap over window rows -- under 2 * 128 + 4 * w pixels)
        const bool cf = (g.conv_cin % MBK) == 0 && (int64_t)(g.conv_w + 1 + 2 * MBM + 4 * g.conv_w + 2 * g.conv_w + 2) * g.conv_cin * 4 < (1ll << 31);
        VFR_REQUIRE(!g.conv_halo || (cf && !(g.epi & EPI_RES)), VFR_EINVAL, "gemm_nt(conv): halo-padded activations need C_in %% 32 == 0 (and no residual)");
#define VFR_CONV(MI_, NARROW_, GRID_, ARGS_)                                                                            \
        do {                                                                                                            \
            if (g.conv_halo) hipLaunchKernelGGL((conv3x3_nhwc_mfma<MI_, NARROW_, true, true>), GRID_, dim3(256), 0, st, ARGS_); \
            else if (cf) hipLaunchKernelGGL((conv3x3_nhwc_mfma<MI_, NARROW_, true>), GRID_, dim3(256), 0, st, ARGS_);    \
            else    hipLaunchKernelGGL((conv3x3_nhwc_mfma<MI_, NARROW_, false>), GRID_, dim3(256), 0, st, ARGS_);        \
        } while (0)
        if (g.N <= 64) {                                      // conv1_x: 128 x 64 tiles, no MFMA spent on absent columns
            VFR_CONV(1, true, dim3(grid.x, 1), g);
        } else if ((int64_t)grid.x * grid.y < 384) {          // under 1.5 workgroups per CU: 64-row tiles (as the dense GEMM)
            VFR_CONV(1, false, dim3((unsigned)cdiv(g.M, 64), grid.y), g);
        } else if (grid.y > 1 && grid.y <= 16 && grid.x >= 64) {     // XCD-aware tile order (see xcd_cols): activations stream once
            GemmArgs gx = g;
            gx.xcd_cols = (int)grid.y;
            VFR_CONV(2, false, dim3((unsigned)(cdiv(grid.x, 8) * 8 * grid.y)), gx);
        } else {
            VFR_CONV(2, false, grid, g);
        }
#undef VFR_CONV
        VFR_CHECK_LAUNCH("conv3x3_nhwc_mfma");
        return VFR_OK;
    }
    if (opt_gemm() == 0) {
        dim3 vgrid((unsigned)cdiv(g.M, VBM), (unsigned)cdiv(g.N, VBN));
        hipLaunchKernelGGL(gemm_nt_valu, vgrid, dim3(256), 0, st, g);
        VFR_CHECK_LAUNCH("gemm_nt_valu");
        return VFR_OK;
    }
    const bool vec = gemm_vec_ok(g);
    if (opt_gemm_small() != 64 && (int64_t)grid.x * grid.y < (opt_gemm_small() > 64 ? opt_gemm_small() : 384)) {
        // fewer 128-row workgroups than 1.5 per CU (context rows, output / query projections, VGG fc6-fc7): 32-row tiles at
        // three workgroups per CU -- four times the workgroups, so the CUs are evenly loaded (tools/small_gemm.py: 10-40 %
        // faster than 64-row tiles on every such GEMM of the pass; alone also 5 % at 824 tiles, a rank's clip encoder on 8 GPUs,
        // but not beside the query encoder on the other stream; slower from ~1600 tiles on.  "gemm_small" 64 forces the 64-row
        // tiles, a value above 64 moves the threshold)
        dim3 grid32((unsigned)cdiv(g.M, 32), grid.y);
        if (vec) hipLaunchKernelGGL((gemm_nt_mfma<true, 0>), grid32, dim3(256), 0, st, g);
        else     hipLaunchKernelGGL((gemm_nt_mfma<false, 0>), grid32, dim3(256), 0, st, g);
        VFR_CHECK_LAUNCH("gemm_nt_mfma<32>");
        return VFR_OK;
    }
    if ((int64_t)grid.x * grid.y < 384) {
        // (cross-check build of the same class: 64-row tiles)
        dim3 grid64((unsigned)cdiv(g.M, 64), grid.y);
        if (vec) hipLaunchKernelGGL((gemm_nt_mfma<true, 1>), grid64, dim3(256), 0, st, g);
        else     hipLaunchKernelGGL((gemm_nt_mfma<false, 1>), grid64, dim3(256), 0, st, g);
        VFR_CHECK_LAUNCH("gemm_nt_mfma<64>");
        return VFR_OK;
    }
    if (vec && g.N <= 64) {
        hipLaunchKernelGGL(gemm_nt_mfma_narrow, dim3(grid.x, 1), dim3(256), 0, st, g);
        VFR_CHECK_LAUNCH("gemm_nt_mfma_narrow");
        return VFR_OK;
    }
    const bool pp = opt_gemm_pp() && (g.K % MBK) == 0 && g.K >= 2 * MBK;
    // Short chains (K <= 2048: ResNet's 1x1 convolutions, the clip encoder's output layer): a 128-row tile's prologue and
    // epilogue weigh as much as its few K-tiles, and three 64-row workgroups per CU cover them better than two 128-row ones
    // (tools/resnet_layers.py: the expansion convolutions 179 -> 148 us, the whole ResNet-152 stack 33.9 -> 31.6 ms).  Long
    // chains (the clip encoder's K = 4096) keep the 128-row tiles (123-127 TF against 117-120).
    if (opt_gemm_small() == 1 || (opt_gemm_small() == 0 && g.K <= 2048)) {
        const unsigned gx64 = (unsigned)cdiv(g.M, 64);
        GemmArgs gx = g;
        const bool xcd = grid.y > 1 && grid.y <= 16 && gx64 >= 64;
        gx.xcd_cols = xcd ? (int)grid.y : 0;
        dim3 gr = xcd ? dim3((unsigned)(cdiv(gx64, 8) * 8 * grid.y)) : dim3(gx64, grid.y);
        if (vec) hipLaunchKernelGGL((gemm_nt_mfma<true, 1>), gr, dim3(256), 0, st, gx);
        else     hipLaunchKernelGGL((gemm_nt_mfma<false, 1>), gr, dim3(256), 0, st, gx);
        VFR_CHECK_LAUNCH("gemm_nt_mfma<64>(large)");
        return VFR_OK;
    }
    if (grid.y > 1 && grid.y <= 16 && grid.x >= 64) {
        // tall GEMM with a few column tiles (the clip encoder's seg x W1: 1641 x 4): XCD-aware tile order, see xcd_cols
        GemmArgs gx = g;
        gx.xcd_cols = (int)grid.y;
        if (pp) {
            dim3 grid1((unsigned)(cdiv(cdiv(grid.x, 2), 8) * 8 * grid.y));
            if (vec) hipLaunchKernelGGL(gemm_nt_mfma_pp<true>, grid1, dim3(512), 0, st, gx);
            else     hipLaunchKernelGGL(gemm_nt_mfma_pp<false>, grid1, dim3(512), 0, st, gx);
        } else {
            dim3 grid1((unsigned)(cdiv(grid.x, 8) * 8 * grid.y));
            if (vec) hipLaunchKernelGGL(gemm_nt_mfma<true>, grid1, dim3(256), 0, st, gx);
            else     hipLaunchKernelGGL(gemm_nt_mfma<false>, grid1, dim3(256), 0, st, gx);
        }
        VFR_CHECK_LAUNCH("gemm_nt_mfma(xcd)");
        return VFR_OK;
    }
    if (pp) {
        dim3 grid2((unsigned)cdiv(grid.x, 2), grid.y);
        if (vec) hipLaunchKernelGGL(gemm_nt_mfma_pp<true>, grid2, dim3(512), 0, st, g);
        else     hipLaunchKernelGGL(gemm_nt_mfma_pp<false>, grid2, dim3(512), 0, st, g);
        VFR_CHECK_LAUNCH("gemm_nt_mfma_pp");
        return VFR_OK;
    }
    if (vec) hipLaunchKernelGGL(gemm_nt_mfma<true>, grid, dim3(256), 0, st, g);
    else     hipLaunchKernelGGL(gemm_nt_mfma<false>, grid, dim3(256), 0, st, g);
    VFR_CHECK_LAUNCH("gemm_nt_mfma");
    return VFR_OK;
}

}  // namespace vfr

extern "C" int vfr_linear_f32(const float *A, int64_t M, int K, const float *W, const float *b, int N, int relu,
                              float *out, vfr_stream_t stream)
{
    VFR_REQUIRE(A && W && out && M >= 0 && K > 0 && N > 0, VFR_EINVAL, "vfr_linear_f32: bad argument");
    vfr::GemmArgs g{};
    g.A = A; g.lda = K; g.W = W; g.ldw = K; g.out = out; g.ldo = N; g.M = M; g.N = N; g.K = K;
    g.bias = b;
    g.epi = (b ? vfr::EPI_BIAS : 0) | (relu ? vfr::EPI_RELU : 0);
    g.site = vfr::SITE_LINEAR;
    return vfr::gemm_nt(g, vfr::as_stream(stream));
}
