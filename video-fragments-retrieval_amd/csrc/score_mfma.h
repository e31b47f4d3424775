// MFMA pre-filter for the fused scorer (included by score.hip, inside namespace vfr).
//
// The exact scorer spends two thirds of its time on the direct-difference distance ((v - q) + eps)^2: three VALU issues
// per (query, clip, dimension), and not a GEMM.  Here the distances come from the GEMM form on the matrix cores,
//
//     d~^2(q, c) = a_c + b_q - 2 v_c.q,     a_c = |v_c|^2 + 2 eps sum(v_c) + D eps^2,     b_q = |q|^2 - 2 eps sum(q)
//
// (a_c, b_q accumulated in fp64 by a pre-pass, one MFMA fma per (query, clip, dimension)), and every decision the pass
// makes with them carries a RIGOROUS error margin, so the emitted results stay those of the exact chain:
//
//   rank counts   per (rank key, span length L) the exact bounds LO/HI on the fp32 SUM of the L clip distances
//                 (score < x <=> S <= LO, score <= x <=> S <= HI) are widened by Delta_L >= |S~ - S|: a moment is counted when
//                 S~ <= LO - Delta_L, not counted when S~ > HI + Delta_L, and a (query, video) pair that holds a moment in
//                 between -- or a clip closer than the distance floor the margin was derived for -- contributes nothing
//                 here and is marked in a (video, query group) bitmap; score_pairs_video_kernel re-scores the marked pairs
//                 video by video with the canonical chain (lane = query) and counts them exactly (ties by moment id).
//   top-k         the selection machinery of score.hip runs unchanged on APPROXIMATE keys for k' = k + MF_EXTRA; the
//                 finisher re-scores those k' candidates exactly and emits the k best.  The exact top-k is contained in
//                 {s~ <= s~_(k) + 2 delta}, which is inside the k' list whenever s~_(k') > s~_(k) + 2 delta (or fewer than
//                 k' candidates exist); the finisher checks that per query.
//   fallback      a query group for which a check fails (more than MF_EXTRA candidates inside the margin -- duplicated
//                 videos --, a key distance too small for any margin) is flagged on the device and answered
//                 by the exact kernels of score.hip, launched with a group mask; everything else returns at once there.
//
// Error margin.  With R = max_c |v_c|, G = (R + |q|)^2 (+ the eps terms):
//   * MFMA side: the K = 100 chain is cut into 6 blocks of 16 products and one of 4, each block summed by
//     v_mfma_f32_16x16x4_f32 from zero (a k-ordered fp32 fma chain: error <= 16 u sum|2 q_k v_k| <= 8 u G over the blocks),
//     the 7 block sums, a_c and b_q added by 8 fp32 adds of partial sums <= G, a_c and b_q each rounded once from fp64:
//     |d~^2 - d*^2| <= E2 := 20 u G  (17 u G by the count above).  Hence |d~ - d*| <= E2 / d~ <= E2 / dfl whenever d~ >= dfl.
//   * oracle side: t_k = fl(fl(v_k - q_k) + eps), S = fma chain of t_k^2: |S - S*| <= 105 u S*, so the exact-path distance d
//     obeys |d - d*| <= 53.5 u d*;  v_sqrt_f32 on the approximate side adds 2 u d~.
//   * sums of L distances add (L - 1) u S on either side.
//   => |S~ - S| <= L E2 / dfl + (56 + 2 L) u S  =: Delta_L   for every moment of a pair whose clips all have d~ >= dfl.
// Centring.  The distance is translation invariant and E2 grows with the squared NORMS, so the GEMM runs on v - mu, q - mu
// (mu = the mean clip embedding, rounded once per element): embeddings that share a large common offset lose it, R and |q|
// above are the norms of the centred rows, and the two roundings move a distance by at most u (R + |q|) -- added to Delta_L.
// The exact paths read the original rows.
// dfl = 3/4 of the smaller rank-key distance of the query (per-query constant); for the top-k check the floor is the best
// approximate score itself (no clip is closer to the query than the best single-clip moment).
//
// bf16 mode (BASELINE config 5): operands rounded to bf16, fp32 accumulate (v_mfma_f32_16x16x32_bf16), a_c / b_q from the
// fp32 data.  Approximate by design: rank counts use the approximate distances as they are, the top-k is the exact re-rank
// of the k' best bf16 candidates.  No margins, no queue, no fallback.
#pragma once

constexpr int MF_EXTRA = 28;                 // k' = k + MF_EXTRA candidates by approximate key
constexpr float MF_U = 5.9604645e-8f;        // 2^-24
constexpr unsigned MF_WIDTH_MAX = 1u << 28;  // window widths (ulps of the sum) the kernel's shifted 10-bit fields hold
constexpr int MF_HBINS = 32;                 // bins of the top-k threshold histogram (below)
constexpr int MF_TAB = 4;                    // table words per (query, key, L): LO, HI (exact, exclusive bit bounds), LO wide, width
#define MF_RING_ROWS(NT) ((NT) > 6 ? 38 : 32)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct MfmaArgs {
    const float *va;                         // [total_clips] a_c
    const float4 *qmeta;                     // [Nq] {b_q, dfl, E2, R + |q|} (centred norms)
    const unsigned *tab;                     // [Nq][NR][NT][MF_TAB]
    unsigned *wmax;                          // [Nq] widest window of the query over keys and span lengths (zeroed per call, atomicMax by the table pre-pass)
    unsigned long long *cnt_ws;              // [NR][Nq] rank counts of this call (committed to count_lt at the end)
    ulonglong2 *amb;                         // [Nv][groups] ambiguous pairs: bit l of .x / .y = query 64 g + l needs key 0 / 1 re-counted
    unsigned long long *pairs_total;         // [1] marked pairs re-scored exactly (statistics)
    int *fallback;                           // [groups] != 0: answered by the exact kernels
    const unsigned short *vb;                // bf16 mode: V as bf16 [total_clips][128] (zero padded)
    const float *rv;                         // [1] max norm of the CENTRED clip rows (pre-pass)
    const float *vc, *qc;                    // V - mu [total_clips, D], Q - mu [Nq, D]: the operands of the approximate GEMM
    unsigned *qbound;                        // [Nq][2] whole-video bounds of the rank keys (zeroed per call, atomicMax by the table pre-pass):
                                             //   [0] bits of HAB: dmin >= HAB => no moment of the video is counted or ambiguous, for either key
                                             //   [1] ~bits of BBL: dmax <  BBL => every moment of the video is counted, for both keys
    unsigned *hist;                          // [Nq][MF_HBINS] top-k threshold histogram (nullable; zeroed per call): counts of candidate keys by score
    uint2 *hrange;                           // [Nq] {score bits of bin 0's lower edge, 1 + log2(bin width in bits)}; .y == 0: histogram not in use for the query
    const int *diff, *perm;                  // (nullable) diff[q] >> 24 = difficulty (0 .. SORT_SAMPLE) of query q; perm[p] = the query at sorted position p (sorted pass)
    int defer_max;                           // whole-video early-out: the rank half of the triangle is skipped when at most this many
                                             // lanes of the wave are left undecided by HAB / BBL; those lanes are marked ambiguous
                                             // (re-counted exactly by score_pairs_video_kernel).  < 0: early-out off
};

// ---------------------------------------------------------------------------------------------------------------------
// pre-passes
// ---------------------------------------------------------------------------------------------------------------------
// Guard of VFR_MFMA_BANK_READY.  The bank-side products in the workspace (mu, centred rows, a_c, max norm, bf16 copy) are
// only as good as the bank they were computed from.  Every pre-filter call hashes what the products depend on -- every
// 32-bit word of V (each multiplied by an odd weight derived from its position, summed modulo 2^64: order-independent, so
// the block partials can be added with atomics) and of the clip offsets -- and a one-thread kernel compares
// {hash, total_clips, Nv, D, eps, bf16 copy present} with the signature the previous call left behind the products:
//   * BANK_READY and equal      -> `stale` = 0: the bank pre-passes, which are always launched, return at once;
//   * anything else             -> `stale` = 1: they recompute, and the new signature is stored.
// No host decision, no synchronisation; 84 MB of V hash in ~20 us.  What is NOT covered: the caller's promise that the
// product region of the workspace itself was not overwritten since that call.
struct BankSig { unsigned long long hash; int total_clips, Nv, D; unsigned eps_bits; int has_bf16; int pad; };
__global__ __launch_bounds__(256) void mfma_bank_hash_kernel(const unsigned *__restrict__ Vw, int64_t nwords,
                                                             const unsigned *__restrict__ off_w, int64_t noff,
                                                             unsigned long long *__restrict__ hash)
{
    __shared__ unsigned long long red[4];
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    unsigned long long h = 0ull;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < nwords; i += stride) {
        if (i + 3 < nwords) {
            const uint4 w = *reinterpret_cast<const uint4 *>(Vw + i);
            const unsigned long long p = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
            h += w.x * (p | 1ull) + w.y * ((p + 0x2545F4914F6CDD1Dull) | 1ull) + w.z * ((p ^ 0xD6E8FEB86659FD93ull) | 1ull) +
                 w.w * ((p * 3ull + 0x632BE59BD9B4E019ull) | 1ull);
        } else {
            for (int64_t j2 = i; j2 < nwords; ++j2) h += Vw[j2] * ((((unsigned long long)j2 + 7ull) * 0xC2B2AE3D27D4EB4Full) | 1ull);
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < noff; i += (int64_t)gridDim.x * 256)
        h += off_w[i] * ((((unsigned long long)i + 11ull) * 0x165667B19E3779F9ull) | 1ull);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) h += __shfl_xor(h, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(hash, red[0] + red[1] + red[2] + red[3]);
}
__global__ void mfma_bank_check_kernel(const unsigned long long *__restrict__ hash_now, BankSig now, int ready_claimed,
                                       BankSig *__restrict__ stored, int *__restrict__ stale, float *__restrict__ rv)
{
    now.hash = *hash_now;
    const BankSig old = *stored;
    const bool same = ready_claimed && old.hash == now.hash && old.total_clips == now.total_clips && old.Nv == now.Nv && old.D == now.D &&
                      old.eps_bits == now.eps_bits && old.has_bf16 >= now.has_bf16;
    *stale = same ? 0 : 1;
    if (!same) { *stored = now; *rv = 0.0f; }                             // (rv: the max-norm accumulator of mfma_prep_v_kernel)
}


// mean clip embedding mu [D] (D <= 128), two deterministic stages: 1024-thread blocks sum a slice of rows (thread = column x
// row phase, fp64), then one 1024-thread block adds the per-block partials (8 slices per column)
__global__ __launch_bounds__(1024) void mfma_mean_partial_kernel(const float *__restrict__ V, int total_clips, int D, int rows_per_block,
                                                                 double *__restrict__ partial, const int *__restrict__ stale)
{
    __shared__ double red[1024];
    if (!*stale) return;                                                  // the products in the workspace are this bank's
    const int k = threadIdx.x & 127, ph = threadIdx.x >> 7;              // column, row phase (8)
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    r1 = r1 < total_clips ? r1 : total_clips;
    double acc = 0.0;
    if (k < D)
        for (int64_t r = r0 + ph; r < r1; r += 8) acc += V[r * D + k];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (ph == 0 && k < D) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i * 128 + k];
        partial[(int64_t)blockIdx.x * D + k] = t;
    }
}
__global__ __launch_bounds__(1024) void mfma_mean_final_kernel(const double *__restrict__ partial, int nblocks, int D, int total_clips,
                                                               float *__restrict__ mu, const int *__restrict__ stale)
{
    __shared__ double red[1024];
    if (!*stale) return;
    const int k = threadIdx.x & 127, sl = threadIdx.x >> 7;
    double acc = 0.0;
    if (k < D)
        for (int b = sl; b < nblocks; b += 8) acc += partial[(int64_t)b * D + k];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0 && k < D) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i * 128 + k];
        mu[k] = (float)(t / (double)(total_clips > 0 ? total_clips : 1));
    }
}

// one wave per clip row: the centred row v - mu (fp32, what the MFMAs read), its a_c in fp64, the largest centred norm
// (float atomicMax on the bits: norms are >= 0), optional bf16 copy
__global__ __launch_bounds__(256) void mfma_prep_v_kernel(const float *__restrict__ V, int total_clips, int D, float eps,
                                                          const float *__restrict__ mu, float *__restrict__ vc,
                                                          float *__restrict__ va, float *__restrict__ rv,
                                                          unsigned short *__restrict__ vb, const int *__restrict__ stale)
{
    if (!*stale) return;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_clips) return;
    const float *v = V + row * D;
    double s2 = 0.0, s1 = 0.0;
    for (int k = lane; k < D; k += 64) {
        const float c = v[k] - mu[k];
        vc[row * D + k] = c;
        const double x = c;
        s2 += x * x; s1 += x;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s2 += __shfl_xor(s2, o, 64); s1 += __shfl_xor(s1, o, 64); }
    if (lane == 0) {
        va[row] = (float)(s2 + 2.0 * (double)eps * s1 + (double)D * (double)eps * (double)eps);
        // (one contended atomic per row would serialise the launch: only rows that can raise the maximum issue one)
        const unsigned nb = __float_as_uint((float)__builtin_sqrt(s2) * 1.0000002f);
        if (nb > __hip_atomic_load(reinterpret_cast<unsigned *>(rv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(reinterpret_cast<unsigned *>(rv), nb);
    }
    if (vb)
        for (int k = lane; k < 128; k += 64) {
            const __bf16 b = (__bf16)(k < D ? v[k] - mu[k] : 0.0f);
            vb[row * 128 + k] = __builtin_bit_cast(unsigned short, b);
        }
}

// 16 lanes per query (coalesced row reads, fp64 shuffle reduction): the centred row q - mu, b_q, E2, the distance floor
__global__ __launch_bounds__(256) void mfma_prep_q_kernel(const float *__restrict__ Q, int64_t Nq, int D, float eps,
                                                          const float *__restrict__ mu, float *__restrict__ qc,
                                                          const float *__restrict__ rv, int NR,
                                                          const float *__restrict__ rank_dist, float4 *__restrict__ qmeta)
{
    const int sub = threadIdx.x & 15;
    const int64_t q = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool on = q < Nq;
    const float *p = Q + (on ? q : 0) * D;
    double s2 = 0.0, s1 = 0.0;
    if (on)
        for (int k = sub; k < D; k += 16) {
            const float c = p[k] - mu[k];
            qc[q * D + k] = c;
            const double x = c;
            s2 += x * x; s1 += x;
        }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { s2 += __shfl_xor(s2, o, 64); s1 += __shfl_xor(s1, o, 64); }
    if (!on || sub != 0) return;
    const float bq = (float)(s2 - 2.0 * (double)eps * s1);
    const float qn = (float)__builtin_sqrt(s2) * 1.0000002f;
    const float R = *rv;
    // G bounds every partial sum of the approximate chain: (R + |q|)^2 + 2 eps sqrt(D) (R + |q|) + D eps^2, rounded up
    const float rs = (R + qn) * 1.000001f;
    const float G = (rs * rs + 2.0f * eps * __builtin_sqrtf((float)D) * rs + (float)D * eps * eps) * 1.00001f;
    // E2 bounds |d~^2 - d'^2| for the CENTRED rows; the centring roundings themselves move a distance by <= u (R + |q|),
    // which the table kernel and the finisher add per clip (qmeta.w = R + |q|)
    const float E2 = 20.0f * MF_U * G;
    float xmin = __builtin_inff();
    for (int r = 0; r < NR; ++r) { const float x = rank_dist[r * Nq + q]; xmin = x < xmin ? x : xmin; }
    const float dfl = NR > 0 && xmin < __builtin_inff() ? 0.75f * xmin : 0.0f;
    qmeta[q] = make_float4(bq, dfl, E2, rs);
}

// one thread per (query, rank key, span length): the exact bit bounds of the sum and their widened forms
template <int NT>
__global__ __launch_bounds__(256) void mfma_prep_tab_kernel(int64_t Nq, int NR, const float *__restrict__ rank_dist,
                                                            const float4 *__restrict__ qmeta, unsigned *__restrict__ tab,
                                                            int bf16_mode, int *__restrict__ fallback, unsigned *__restrict__ wmax,
                                                            unsigned *__restrict__ qbound)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Nq * NR * NT) return;
    const int L = (int)(i % NT) + 1, r = (int)((i / NT) % NR);
    const int64_t q = i / ((int64_t)NT * NR);
    const float4 meta = qmeta[q];
    const float dfl = meta.y, E2 = meta.z;
    const float x = rank_dist[r * Nq + q];
    unsigned *t = tab + i * MF_TAB;
    const float lo = sum_bound<true>(x, L), hi = sum_bound<false>(x, L);
    const unsigned LOX = min(excl_bound(lo), 0x7F800000u), HIX = min(excl_bound(hi), 0x7F800000u);
    unsigned LOW = LOX, HIW = HIX;
    if (!bf16_mode && x < __builtin_inff()) {
        // Delta_L: see the header.  S is at most hi + Delta_L near the window; (56 + 2L) u S with 2 % head-room.
        const float Sref = (hi > 0.0f ? hi : 0.0f) + (float)L * 1e-3f;
        const float delta = dfl > 0.0f ? ((float)L * (E2 / dfl + MF_U * meta.w * 1.01f) + (float)(56 + 2 * L) * MF_U * Sref * 1.02f) * 1.0001f
                                       : __builtin_inff();
        float low = lo - delta;                                           // lo = -1: no sum is below the key
        low = low > 0.0f ? next_down(low) : -1.0f;
        float hiw = (hi > 0.0f ? hi : 0.0f) + delta;
        hiw = hiw < __builtin_inff() ? next_up(hiw) : hiw;
        LOW = min(excl_bound(low), 0x7F800000u);
        HIW = min(excl_bound(hiw), 0x7F800000u);
        if (HIW < HIX) HIW = HIX;
        if (LOW > LOX) LOW = LOX;
    }
    *reinterpret_cast<uint4 *>(t) = make_uint4(LOX, HIX, LOW, HIW - LOW);
    // Whole-video bounds (score_mfma_kernel's early-out).  A sum of L approximate distances, each >= dmin, is at least
    // L dmin (1 - u)^(L - 1) >= L dmin (1 - 21 u); so dmin >= (H / L)(1 + 66 u) with H = the float whose bits are HIW puts every sum of
    // the level at or above HIW: not counted, not in the window.  Likewise dmax < (Lo / L)(1 - 66 u), Lo = the float of LOW, puts
    // every sum strictly below LOW: counted.  HAB = max over keys and levels of the first, BBL = min of the second (one pair per
    // query: "above both keys" / "below both keys"; a video between the keys stays undecided).  The division and the product
    // round by <= 2 u, the next_up / next_down absorb the rest.
    if (qbound) {
        const float H = __uint_as_float(HIW), Lo = __uint_as_float(LOW);
        float ta = (H / (float)L) * (1.0f + 66.0f * MF_U);
        ta = ta < __builtin_inff() ? next_up(ta) : ta;
        float tb = (Lo / (float)L) * (1.0f - 66.0f * MF_U);
        tb = tb < __builtin_inff() ? (tb > 0.0f ? next_down(tb) : 0.0f) : 3.4028235e38f;
        atomicMax(qbound + 2 * q, __float_as_uint(ta));
        atomicMax(qbound + 2 * q + 1, ~__float_as_uint(tb));
    }
    // The kernel keeps the widths as 10-bit mantissas under one per-query shift (rounded up: a window is never narrowed), so
    // a key deep in the tail -- norms far above the key distance, windows of thousands of ulps -- stays on this path: only
    // its ambiguous pairs cost more.  A key distance so small that no margin exists (delta = inf: the window is the whole
    // float range) sends the query's group to the exact kernels right away; the pre-filter kernel returns at once for it.
    if (!bf16_mode && HIW - LOW >= MF_WIDTH_MAX) fallback[q >> 6] = 1;
    if (!bf16_mode && HIW - LOW > 1023u) atomicMax(wmax + q, HIW - LOW);   // (narrower windows need no shift: no atomic in the common case)
}

// ---------------------------------------------------------------------------------------------------------------------
// Query order.  The whole-video early-out of score_mfma_kernel is a WAVE decision (64 queries), and query batches are
// mixtures: with rank keys in the near tail half the queries leave (almost) no video undecided while a quarter leave most
// of them -- one such query per wave and the wave runs every triangle.  So the pass runs on a permutation of the batch:
// queries sorted by DIFFICULTY = how many of SORT_SAMPLE sample videos (evenly spaced through the bank) the query's smaller rank key
// leaves undecided (smallest clip distance below the key, largest not), then the key distance itself.  A heuristic on plain arithmetic (no margins): it
// only chooses which queries share a wave; every result is computed as before and scattered back to the caller's order.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SORT_MAX_QUERIES = 12288;      // (the rank computation keeps every key in LDS: 48 KB)
constexpr int SORT_SAMPLE = 8;               // sample videos per query (64 were measured first: the sample pass then cost 70 us, as much as the
                                             // sort saves on a corpus whose every query is hard; 8 videos + the key distance order nearly as well)
// wave-task (query group, sample video): lane = query (its row in registers), the video's clip rows are wave-uniform (scalar
// loads); a lane for which the video is undecided adds one to its query's count
__global__ __launch_bounds__(64) void mfma_difficulty_kernel(const float *__restrict__ Q, int64_t Nq, const float *__restrict__ V,
                                                             const int32_t *__restrict__ clip_off, int Nv, int NR,
                                                             const float *__restrict__ rank_dist, int *__restrict__ diff)
{
    const int lane = threadIdx.x, group = blockIdx.x / SORT_SAMPLE, sv = blockIdx.x % SORT_SAMPLE;
    const int64_t q = (int64_t)group * 64 + lane;
    const bool active = q < Nq;
    const float4 *q4 = reinterpret_cast<const float4 *>(Q + (active ? q : Nq - 1) * FAST_D);
    float4 qr[FAST_D / 4];
#pragma unroll
    for (int j4 = 0; j4 < FAST_D / 4; ++j4) qr[j4] = q4[j4];
    float x = __builtin_inff();
    for (int r = 0; r < NR; ++r) { const float y = rank_dist[r * Nq + (active ? q : Nq - 1)]; x = y < x ? y : x; }
    const int v = (int)((int64_t)sv * Nv / SORT_SAMPLE);
    const int c0 = clip_off[v], c1 = clip_off[v + 1];
    float dmin2 = __builtin_inff(), dmax2 = 0.0f;
    for (int c = c0; c < c1; ++c) {
        const float *vr = V + (int64_t)c * FAST_D;                       // wave-uniform
        float acc = 0.0f;
#pragma unroll
        for (int j4 = 0; j4 < FAST_D / 4; ++j4) {
            float t = vr[4 * j4] - qr[j4].x; acc = __builtin_fmaf(t, t, acc);
            t = vr[4 * j4 + 1] - qr[j4].y; acc = __builtin_fmaf(t, t, acc);
            t = vr[4 * j4 + 2] - qr[j4].z; acc = __builtin_fmaf(t, t, acc);
            t = vr[4 * j4 + 3] - qr[j4].w; acc = __builtin_fmaf(t, t, acc);
        }
        dmin2 = acc < dmin2 ? acc : dmin2;
        dmax2 = acc > dmax2 ? acc : dmax2;
    }
    const float x2 = x * x;
    // sort key: (undecided sample videos) << 24 | the top 24 bits of the key distance -- among queries that leave the same
    // number of sample videos undecided the one with the smaller key is the easier (its key sits deeper in the tail)
    if (active && c1 > c0 && dmin2 < x2 * 1.0002f && !(dmax2 < x2 * 0.9998f)) atomicAdd(diff + q, 1 << 24);
    if (active && sv == 0) atomicAdd(diff + q, (int)((__float_as_uint(x < 0.0f ? 0.0f : x) >> 7) & 0xFFFFFFu));
}
// perm[rank] = q with rank = #{j : diff[j] < diff[q]} + #{j < q : diff[j] == diff[q]} (a stable sort, deterministic); 16 lanes
// share a query's scan over all keys (LDS)
__global__ __launch_bounds__(256) void mfma_sort_perm_kernel(const int *__restrict__ diff, int Nq, int *__restrict__ perm)
{
    __shared__ int ds[SORT_MAX_QUERIES];
    for (int i = threadIdx.x; i < Nq; i += 256) ds[i] = diff[i];
    __syncthreads();
    const int sub = threadIdx.x & 15, q = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int qq = q < Nq ? q : Nq - 1;
    const int mine = ds[qq];
    int rank = 0;
    for (int j = sub; j < Nq; j += 16) {
        const int o = ds[j];
        rank += (o < mine || (o == mine && j < qq)) ? 1 : 0;
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) rank += __shfl_xor(rank, o, 64);
    if (q < Nq && sub == 0) perm[rank] = q;
}
// sorted copies of the per-query inputs: Qs[p] = Q[perm[p]], rank keys, threshold seeds
__global__ __launch_bounds__(256) void mfma_gather_queries_kernel(const int *__restrict__ perm, int64_t Nq, const float *__restrict__ Q,
                                                                  int NR, const float *__restrict__ rank_dist,
                                                                  const int64_t *__restrict__ rank_idx, const int64_t *__restrict__ seed,
                                                                  float *__restrict__ Qs, float *__restrict__ rds,
                                                                  int64_t *__restrict__ ris, int64_t *__restrict__ seeds)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Nq * FAST_D) return;
    const int64_t p = i / FAST_D;
    const int kk = (int)(i - p * FAST_D);
    const int64_t src = perm[p];
    Qs[i] = Q[src * FAST_D + kk];
    if (kk < NR) { rds[kk * Nq + p] = rank_dist[kk * Nq + src]; ris[kk * Nq + p] = rank_idx[kk * Nq + src]; }
    if (kk == MAX_RANK && seed) seeds[p] = seed[src];
}
// results back to the caller's order: lists copied, counts ADDED (count_lt accumulates by contract)
__global__ __launch_bounds__(256) void mfma_scatter_results_kernel(const int *__restrict__ perm, int64_t Nq, int k, int NR,
                                                                   const float *__restrict__ ods, const int64_t *__restrict__ ois,
                                                                   const int64_t *__restrict__ cnts, float *__restrict__ out_dist,
                                                                   int64_t *__restrict__ out_idx, int64_t *__restrict__ count_lt)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per = k > NR ? k : NR;
    if (i >= Nq * per) return;
    const int64_t p = i / per;
    const int j = (int)(i - p * per);
    const int64_t dst = perm[p];
    if (j < k) { out_dist[dst * k + j] = ods[p * k + j]; out_idx[dst * k + j] = ois[p * k + j]; }
    if (j < NR) count_lt[j * Nq + dst] += cnts[j * Nq + p];
}

// ---------------------------------------------------------------------------------------------------------------------
// approximate fused kernel.  thread = query for the moment triangle (as score_fast_kernel); the distances of 16-clip
// column tiles x the wave's 64 queries come from 100 (f32) / 16 (bf16) MFMAs per tile, the accumulator tiles go through a
// per-wave LDS ring (clip-major rows of 64 queries, 16-byte chunks XOR-swizzled by row so that both the b128 tile writes
// and the per-query reads are bank-conflict free) and a video is processed as soon as its last clip has been written.
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, int KPL, int NR, bool TOPK, bool BF16>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8)))
void score_mfma_kernel(const float *__restrict__ Qp, const float *__restrict__ Vp, const int32_t *__restrict__ clip_off,
                       const int64_t *__restrict__ mom_off, ScoreArgs a, MfmaArgs m)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // ring rows: a tile of 16 clips lands while at most NT - 1 rows of unfinished videos are still live -> NT + 15 rows are
    // enough; 38 rows (9.5 KB) + the 10.5 KB bound table = 20 KB per wave = eight waves per CU (48 rows left room for seven:
    // one SIMD of every CU ran a single wave with nothing to cover its operand latency)
    constexpr int CAP = KPL * 64, RR = MF_RING_ROWS(NT), NRR = NR > 0 ? NR : 1, NW = (NT + 2) / 3;
    constexpr int KS = BF16 ? 4 : 25;                                   // MFMA k-steps per tile
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    const int task = blockIdx.x;
    const int chunk = task / a.num_groups, group = task - chunk * a.num_groups;
    float *ring = smem;                                                  // [RR][64] approximate d^2 - b_q
    unsigned *lox_lds = reinterpret_cast<unsigned *>(smem + RR * 64);    // [NR][NT][64] widened lower bounds

    const int64_t qi = (int64_t)group * 64 + lane;
    const bool active = qi < a.Nq;
    if (!BF16 && m.fallback[group]) {                                     // this group is answered by the exact kernels
        if (TOPK) a.buf_cnt[(size_t)task * 64 + lane] = 0;
        return;
    }
    const float4 meta = m.qmeta[active ? qi : a.Nq - 1];
    const float bq = meta.x, dfl = (active && !BF16) ? meta.y : 0.0f;

    // ---- A operand: -2 q, queries 16t + j of the group, k-slots of this lane's quarter g ----
    float Af[BF16 ? 1 : 4][BF16 ? 1 : 25];
    bf16x8 Ab[BF16 ? 4 : 1][BF16 ? 4 : 1];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int64_t row = (int64_t)group * 64 + 16 * t + j;
        row = row < a.Nq ? row : a.Nq - 1;
        const float *p = m.qc + row * FAST_D;
        if constexpr (!BF16) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float4 x = *reinterpret_cast<const float4 *>(p + 24 * g + 4 * i);
                Af[t][4 * i] = -2.0f * x.x; Af[t][4 * i + 1] = -2.0f * x.y; Af[t][4 * i + 2] = -2.0f * x.z; Af[t][4 * i + 3] = -2.0f * x.w;
            }
            Af[t][24] = -2.0f * p[96 + g];
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = 32 * s + 8 * g + i;
                    Ab[t][s][i] = (__bf16)(e < FAST_D ? -2.0f * p[e < FAST_D ? e : 0] : 0.0f);
                }
            }
        }
    }

    const int v0 = a.v_lo + (int)((int64_t)(a.v_hi - a.v_lo) * chunk / a.num_chunks);
    const int v1 = a.v_lo + (int)((int64_t)(a.v_hi - a.v_lo) * (chunk + 1) / a.num_chunks);

    // ---- thresholds ----
    unsigned long long thr = KEY_MAX;
    int cnt = 0, nlt[NRR] = {0};
    unsigned long long *col = TOPK ? a.buf + (size_t)task * 64 * CAP : nullptr;
    float thrf = __builtin_inff();                                       // top-k filter: a moment of L clips may enter when sum <= thrf * L (* 1 + 8u)
    unsigned wpk[NW];                                                    // window width per span length (max over keys): 10-bit mantissas, 3 per register ...
    unsigned wsh = 0u;                                                   // ... under one per-lane shift: width = mantissa << wsh, rounded UP
    bool wide = false;
#pragma unroll
    for (int w = 0; w < NW; ++w) wpk[w] = 0u;
    if (NR > 0 && !BF16) {
        const unsigned wmax = active ? m.wmax[qi] : 0u;                   // 0: every window of the query fits 10 bits
        wide = wmax >= MF_WIDTH_MAX;                                      // (its group is flagged by the table pre-pass)
        const int bits = 32 - __builtin_clz(wmax | 1u);
        wsh = bits > 10 ? (unsigned)(bits - 10) : 0u;
    }
#pragma unroll
    for (int L = 1; L <= NT; ++L) {
        unsigned wd = 0u;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const unsigned *t = m.tab + (((active ? qi : 0) * NR + r) * NT + (L - 1)) * MF_TAB;
            const uint4 e = *reinterpret_cast<const uint4 *>(t);
            lox_lds[(r * NT + (L - 1)) * 64 + lane] = active ? (BF16 ? e.x : e.z) : 0u;
            const unsigned w_ = (active && !BF16) ? e.w : 0u;
            wd = w_ > wd ? w_ : wd;
        }
        // mantissa: exact when wsh == 0; otherwise floor + 1 (< 1024 because wmax < 2^(10 + wsh))
        const unsigned mant = wsh ? (wd >> wsh) + 1u : wd;
        wpk[(L - 1) / 3] |= (mant & 1023u) << (10 * ((L - 1) % 3));
        asm volatile("" ::: "memory");               // one level's table words in flight at a time (not 42 x 4 registers)
    }
    if (!active) thrf = -1.0f;
    // ---- top-k threshold from the candidate histogram (main launch of the ladder only).  Stage B's merge leaves, per query, its
    // k' best keys spread over MF_HBINS bins of score bits between its best and its k'-th key, and every candidate this launch
    // appends adds one to its bin.  A task that starts later reads the bins: if bins 0 .. b already hold k' keys, every one of
    // the final k' best keys lies below bin b's upper edge -- a valid threshold (k' real keys bound the k'-th best), tighter than
    // stage B's with every task that has finished.  Counts are read while others add to them: a late count only loosens. ----
    bool hon = false;                                                     // (the range is re-read where a candidate is counted: no registers to spare)
    if (TOPK && m.hist && !a.keep_all && a.level_cap == 0 && active) {
        const uint2 hr = m.hrange[qi];
        if (hr.y != 0u) {
            hon = true;
            const unsigned hlob = hr.x, hshift = hr.y - 1u;
            const uint4 *hb = reinterpret_cast<const uint4 *>(m.hist + (size_t)qi * MF_HBINS);
            unsigned cum = 0u, edge_bin = MF_HBINS;
#pragma unroll
            for (int i = 0; i < MF_HBINS / 4; ++i) {
                const uint4 c4 = hb[i];
                const unsigned cs[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    cum += cs[jx];
                    if (cum >= (unsigned)a.k && edge_bin == MF_HBINS) edge_bin = (unsigned)(4 * i + jx);
                }
            }
            if (edge_bin < MF_HBINS - 1) {                                 // (the last bin also holds everything above the range: no edge)
                const unsigned long long tc = (unsigned long long)(hlob + ((edge_bin + 1u) << hshift)) << 32;
                if (tc < thr) {
                    thr = tc;
                    __hip_atomic_fetch_min(a.thr_global + qi, thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    // whole-video bounds of the rank keys (table pre-pass): see MfmaArgs::qbound
    // (re-read per video -- one 8-byte load from the L1 -- rather than held: the kernel has no two registers to spare)
    int defer_max = (NR > 0 && m.defer_max >= 0) ? (BF16 ? 0 : m.defer_max) : -1;   // (bf16 mode has no exact re-count: all-decided videos only)
    if (NR > 0 && defer_max >= 0 && m.diff) {
        // a wave whose every query left (nearly) all sample videos undecided will not see a decidable video either: it runs without
        // the test (mid-distribution keys: the whole batch)
        const int dq = active ? (m.diff[m.perm ? m.perm[qi] : qi] >> 24) : SORT_SAMPLE;
        if (__ballot(dq < SORT_SAMPLE) == 0ull) defer_max = -1;
    }


    // ---- B operand: the clip rows of a tile, k-block by k-block ----
    // f32: 6 blocks of 4 k-steps (one 16-byte load per lane: elements 24 g + 4 b .. + 3 of row j) + the 25th step (element
    // 96 + g).  Only a window of three blocks is held in registers: block b + 3 is requested while block b feeds its 16 MFMAs,
    // the last three requests of a tile are the first three blocks of the NEXT tile, which land under the moment triangles.
    const int64_t last_row = (int64_t)a.total_clips - 1;
    float4 Bq[BF16 ? 1 : 3];
    float Bl = 0.0f, Bl_next = 0.0f;                                     // 25th k-step of this / the next tile
    bf16x8 Bb[BF16 ? 4 : 1];
    float acv_next = 0.0f;
    auto brow = [&](int64_t c_tile) -> int64_t { const int64_t row = c_tile + j; return row < last_row ? row : last_row; };
    auto bblock = [&](int64_t row, int b) -> float4 { return *reinterpret_cast<const float4 *>(m.vc + row * FAST_D + 24 * g + 4 * b); };
    auto bload_bf16 = [&](int64_t c_tile) {
        const int64_t row = brow(c_tile);
        acv_next = m.va[row];
        const unsigned short *p = m.vb + row * 128;
#pragma unroll
        for (int s = 0; s < 4; ++s) Bb[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(p + 32 * s + 8 * g));
    };

    // per-lane ring addressing (bytes): write = row j of the tile, chunk (4t + g) ^ j; read = chunk (lane >> 2) ^ (row & 15)
    const unsigned rd_lane = (unsigned)(((lane >> 2) << 4) | ((lane & 3) << 2));
    // a clip's ring row is (clip - c_lo) mod RR, its swizzle key (clip - c_lo) & 15: the 16 rows of a tile always carry 16
    // different keys, wherever the tile wraps
    auto ring_read = [&](int row, int key) -> float {                    // row, key wave-uniform
        const unsigned off = (rd_lane ^ ((unsigned)(key & 15) << 4)) + (unsigned)row * 256u;
        return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ring) + off);
    };

    if (v0 >= v1) {
        if (TOPK) a.buf_cnt[(size_t)task * 64 + lane] = 0;
        return;
    }
    const int c_lo = clip_off[v0], c_hi = clip_off[v1];
    int v = v0, c_cur = c_lo, c_nxt = clip_off[v0 + 1];
    int64_t m_cur = mom_off[v0];
    if constexpr (BF16) bload_bf16(c_lo);
    else {
        const int64_t row = brow(c_lo);
        acv_next = m.va[row];
        Bl_next = m.vc[row * FAST_D + 96 + g];
#pragma unroll
        for (int b = 0; b < 3; ++b) Bq[b] = bblock(row, b);
    }
    int tile_row0 = 0;
    for (int c_tile = c_lo; c_tile < c_hi || v < v1; c_tile += 16) {
        // ---- 64 queries x 16 clips on the matrix cores ----
        f32x4 acc[4];
        const float acv_cur = acv_next;
        if constexpr (!BF16) {
            const int64_t row = brow(c_tile), row_n = brow((int64_t)c_tile + 16);
            Bl = Bl_next;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const float4 cur = Bq[b % 3];
                const float bs[4] = {cur.x, cur.y, cur.z, cur.w};
                f32x4 part[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) part[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#ifdef VFR_MF_SKIP_MFMA
                        part[t][s4] += Af[t][4 * b + s4] * bs[s4];      // timing experiment: one VALU op instead of the MFMA
#else
                        part[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Af[t][4 * b + s4], bs[s4], part[t], 0, 0, 0);
#endif
                }
                Bq[b % 3] = b < 3 ? bblock(row, b + 3) : bblock(row_n, b - 3);
                if (b == 3) { Bl_next = m.vc[row_n * FAST_D + 96 + g]; acv_next = m.va[row_n]; }
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = b == 0 ? part[t] : acc[t] + part[t];
            }
            {
                f32x4 part[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) part[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Af[t][24], Bl, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = acc[t] + part[t];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ab[t][s], Bb[s], acc[t], 0, 0, 0);
            }
            bload_bf16((int64_t)c_tile + 16);                            // next tile's operands land under the triangles
        }
        {
            int wr = tile_row0 + j;                                      // (tiles start at multiples of 16 clips: key of row j is j)
            wr = wr >= RR ? wr - RR : wr;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 o = acc[t] + acv_cur;
                const unsigned off = (unsigned)wr * 256u + ((unsigned)((4 * t + g) ^ j) << 4);
                *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(ring) + off) = o;
            }
        }
        tile_row0 = tile_row0 + 16 >= RR ? tile_row0 + 16 - RR : tile_row0 + 16;
        const int c_done = c_tile + 16;

        // ---- every video whose last clip is in the ring ----
        while (v < v1 && c_nxt <= c_done) {
            const int c0 = c_cur, n = c_nxt - c_cur;
            const int64_t mbase = m_cur;
            c_cur = c_nxt;
            ++v;
            if (v < v1) { c_nxt = clip_off[v + 1]; m_cur = mom_off[v]; }
            const int r0 = (c0 - c_lo) % RR, k0 = (c0 - c_lo) & 15;
#ifdef VFR_MF_SKIP_TRI
            if (ring_read(r0, k0) == 123.456f) nlt[0] += (int)mbase + n;    // timing experiment: no triangle
            continue;
#endif
            if (TOPK) {
                if (__ballot(cnt > CAP - n * (n + 1) / 2)) {
                    __threadfence_block();
                    const unsigned long long before = thr;
                    lane_tighten(col, lane, a.k, &cnt, &thr);
                    __threadfence_block();
                    if (thr < before && active) __hip_atomic_fetch_min(a.thr_global + qi, thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (((v - v0) & 7) == 1 && active) {
                    const unsigned long long gthr = __hip_atomic_load(a.thr_global + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    thr = gthr < thr ? gthr : thr;
                }
                // score <= thr  =>  fl(sum / L) <= thr  =>  sum <= thr L (1 + u): the level test is one multiply, no table
                if (active) thrf = thr == KEY_MAX ? __builtin_inff() : __uint_as_float((unsigned)(thr >> 32)) * 1.000001f;
            }
            // approximate distances of this lane's query to the n clips
            float d[NT], sums[NT];
            float dmin = __builtin_inff(), dmax = 0.0f;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                int row = r0 + c;
                row = row >= RR ? row - RR : row;
                const float x2 = ring_read(row, k0 + c) + bq;
                const float dd = __builtin_amdgcn_sqrtf(x2 > 0.0f ? x2 : 0.0f);
                d[c] = c < n ? dd : __builtin_inff();
                dmin = d[c] < dmin ? d[c] : dmin;
                if (NR > 0) { const float dr = c < n ? dd : 0.0f; dmax = dr > dmax ? dr : dmax; }
            }
            int amb[NRR] = {0}, cv[NRR] = {0};
            unsigned lvl = 0;
            const bool low = !BF16 && (wide || dmin < dfl);
            // ---- whole-video early-out.  Every moment score is a mean of the video's clip distances, so dmin / dmax decide all
            // n(n+1)/2 moments at once when they clear the keys' bounds (HAB / BBL: margins derived with the table, mfma_prep_tab_kernel).
            // Rank half: a lane is DECIDED when its video lies above both keys (count 0) or below both (count n(n+1)/2), or is
            // marked anyway (`low`).  When at most defer_max lanes of the wave are undecided the rank half of the triangle is not
            // run: the undecided lanes are marked ambiguous (their pairs are re-counted exactly by score_pairs_video_kernel, like
            // any pair the windows leave open) -- with keys in the near tail, what a trained model produces, that is almost every
            // video.  Top-k half: no moment can pass `sum <= thrf L` when dmin (1 - 21 u) > thrf (1 + u): skipped when no lane has one.
            bool do_rank = NR > 0, do_topk = TOPK;
            bool und = false, below = false;
            if (NR > 0 && defer_max >= 0) {
                // (address rebuilt from the lane id every time: a held pointer would cost two registers the kernel does not have)
                unsigned qx = (unsigned)(group * 64 + lane);
                asm volatile("" : "+v"(qx));
                qx = qx < (unsigned)a.Nq ? qx : (unsigned)a.Nq - 1u;
                const uint2 qb = reinterpret_cast<const uint2 *>(m.qbound)[qx];
                const float hab = __uint_as_float(qb.x), bbl = __uint_as_float(~qb.y);
                below = active && dmax < bbl;
                und = !low && !(below || dmin >= hab);
                if (__builtin_popcountll(__ballot(active && und)) <= defer_max) do_rank = false;
            }
            if (TOPK) do_topk = __ballot(dmin <= thrf * 1.00001f) != 0ull;
            auto triangle = [&](auto rank_c, auto topk_c) {
                constexpr bool RANK = decltype(rank_c)::value, TK = decltype(topk_c)::value;
                unsigned lxn[NRR];                                       // the next level's bounds, requested one level ahead
                if (RANK) {
#pragma unroll
                    for (int r = 0; r < NR; ++r) lxn[r] = lox_lds[(r * NT) * 64 + lane];
                }
#pragma unroll
                for (int L = 1; L <= NT; ++L) {
                    static_assert(NT <= 32, "one 32-bit sign collector per level");
                    unsigned lx[NRR], umin[NRR], below_[NRR];
                    const unsigned wd = ((wpk[(L - 1) / 3] >> (10 * ((L - 1) % 3))) & 1023u) << wsh;
                    if (RANK) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            below_[r] = 0u;
                            lx[r] = lxn[r];
                            if (L < NT) lxn[r] = lox_lds[(r * NT + L) * 64 + lane];
                            umin[r] = 0xFFFFFFFFu;
                        }
                    }
                    float tmin = __builtin_inff();
#pragma unroll
                    for (int s = 0; s + L <= NT; ++s) {
                        const float de = d[s + L - 1];
                        const float sum = L == 1 ? de : sums[s] + de;
                        sums[s] = sum;
                        const unsigned sb = __float_as_uint(sum);
                        if (TK) tmin = __builtin_fminf(sum, tmin);
                        if (RANK) {
#pragma unroll
                            for (int r = 0; r < NR; ++r) {
                                const unsigned u = sb - lx[r];
                                below_[r] = __builtin_amdgcn_alignbit(below_[r], u, 31);
                                umin[r] = u < umin[r] ? u : umin[r];
                            }
                        }
                    }
                    if (RANK) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            cv[r] += __builtin_popcount(below_[r]);
                            amb[r] += umin[r] < wd ? 1 : 0;
                        }
                    }
                    if (TK) lvl = lvl + lvl + (tmin <= thrf * (float)L ? 1u : 0u);
                    if (RANK && NR == 2) asm volatile("" : "+v"(cv[0]), "+v"(cv[NR > 1 ? 1 : 0]), "+v"(amb[0]), "+v"(amb[NR > 1 ? 1 : 0]), "+v"(lvl));
                    else asm volatile("" : "+v"(lvl));
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (do_rank && do_topk) triangle(std::true_type{}, std::integral_constant<bool, TOPK>{});
            else if (do_rank) triangle(std::true_type{}, std::false_type{});
            else if (do_topk) triangle(std::false_type{}, std::integral_constant<bool, TOPK>{});
            if (NR > 0 && !do_rank) {
                // decided by the bounds: all n(n+1)/2 moments below both keys, or none; an undecided lane is left to the exact kernel
                const int Mn = n * (n + 1) / 2;
#pragma unroll
                for (int r = 0; r < NR; ++r) { cv[r] = below ? Mn : 0; amb[r] = und ? 1 : 0; }
            }
            if (NR > 0) {
                unsigned rmask = 0u;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const bool am = low || amb[r] != 0;
                    nlt[r] += am ? 0 : cv[r];
                    rmask |= (am ? 1u : 0u) << r;
                }
                if (!BF16) {
                    // every (video, group) slot has exactly one writer (this wave) per call: no atomics, no capacity
                    const unsigned long long m0 = __ballot(active && (rmask & 1u)), m1 = __ballot(active && (rmask & 2u));
                    if (lane == 0) m.amb[(size_t)(v - 1) * a.num_groups + group] = ulonglong2{m0, m1};
                }
            }
            if (TOPK && lvl != 0) {
                // candidates by approximate key: the flagged levels again, then only the moments under the bound
                auto dist_at = [&](int c) -> float {                     // per-lane clip index
                    int row = r0 + c;
                    row = row >= RR ? row - RR : row;
                    const unsigned off = (rd_lane ^ ((unsigned)((k0 + c) & 15) << 4)) + (unsigned)row * 256u;
                    const float x2 = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ring) + off) + bq;
                    return __builtin_amdgcn_sqrtf(x2 > 0.0f ? x2 : 0.0f);
                };
                // (the distances as opaque values: otherwise the level sums below are "the same" expressions as the triangle's
                // and the compiler keeps all 231 of those alive up to here)
                float dd[NT];
#pragma unroll
                for (int c = 0; c < NT; ++c) { dd[c] = d[c]; asm volatile("" : "+v"(dd[c])); }
#pragma unroll
                for (int L = 1; L <= NT; ++L) {
                    if (((lvl >> (NT - L)) & 1u) && (a.level_cap == 0 || L <= a.level_cap)) {
                        const float hx = thrf * (float)L;
                        unsigned pm = 0;
#pragma unroll
                        for (int s = 0; s + L <= NT; ++s) {
                            float sum = dd[s];
#pragma unroll
                            for (int e = s + 1; e < s + L; ++e) sum += dd[e];
                            pm |= (sum <= hx ? 1u : 0u) << s;
                        }
                        while (pm) {
                            const int s = __builtin_ctz(pm);
                            pm &= pm - 1u;
                            if (s + L <= n) {
                                float sum = dist_at(s);
#pragma nounroll
                                for (int e = s + 1; e < s + L; ++e) sum += dist_at(e);
                                const float sc = sum / (float)L;
                                const unsigned id = (unsigned)(a.id_base + mbase + moment_index(n, s, s + L - 1));
                                const unsigned long long key = make_key(sc, id);
                                if (key < thr) {
                                    col[(size_t)cnt * 64 + lane] = key; ++cnt;
                                    if (hon) {
                                        const uint2 hr = m.hrange[qi];
                                        const unsigned hlob = hr.x, hshift = hr.y - 1u;
                                        const unsigned sbk = __float_as_uint(sc);
                                        const unsigned bin = (sbk > hlob ? sbk - hlob : 0u) >> hshift;
                                        atomicAdd(m.hist + (size_t)qi * MF_HBINS + (bin < MF_HBINS - 1 ? bin : MF_HBINS - 1), 1u);
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
    }

    if (NR > 0) {
        if (active)
            for (int r = 0; r < NR; ++r)
                if (nlt[r]) atomicAdd(m.cnt_ws + r * a.Nq + qi, (unsigned long long)nlt[r]);
    }
    if (TOPK) {
        if (!a.keep_all) {
            __threadfence_block();
            lane_tighten(col, lane, a.k, &cnt, &thr);
            __threadfence_block();
        }
        a.buf_cnt[(size_t)task * 64 + lane] = active ? cnt : 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// exact re-scoring of the marked (query, video) pairs, video by video.  One wave per video:
//   * the video's clip rows are staged once into LDS (coalesced 16-byte loads);
//   * the (group, key) bitmaps of the video are expanded, 64 query groups at a time, into a compact list of its marked
//     queries (lane = group: popcount, wave scan, every lane writes its own set bits);
//   * 64 pairs at a time, lane = query: the query row is gathered into registers, the canonical k-ascending chains of the
//     video's clips run with the clip row BROADCAST from LDS (what score_fast_kernel does for all pairs -- the same
//     instruction mix, full lanes), then the moment triangle against the EXACT bit bounds of the table (LOX / HIX: no
//     division); a sum that lands exactly on a key (LOX <= bits < HIX) sends the lane through the tie walk (ids compared).
// (Round 2 queued pairs per wave-task of the pre-filter kernel, in video order, and ran the chains with lane = (pair, clip)
// from per-lane LDS reads: ~4.5 pairs shared a video, 75 % of the lanes worked and the kernel waited on LDS -- 2.2 ms for the
// 7 % of pairs of the bench corpus.)
// ---------------------------------------------------------------------------------------------------------------------
// One or two query groups (Nq <= 128): a video has a handful of marked pairs, a wave per video would run its 7 us of chains for 5
// of 64 lanes (10 000 such waves cost 0.18 ms at 64 queries).  The VB = 8 instantiation gives a wave EIGHT consecutive videos: their
// rows are staged together (they are adjacent in V), lane = (video, group) slot expands the bitmaps (the slots of consecutive
// videos are adjacent in memory: one coalesced load), and a batch mixes pairs of all eight videos -- every lane reads ITS
// video's row from LDS (the rows of different videos sit 8400 bytes apart: distinct banks for up to 16 videos).
constexpr int PV_LIST = 1024;                // list window: entries expanded at a time (a 64-group chunk rarely marks more)
template <int NT, int NR, int VB>
__global__ __launch_bounds__(64) void score_pairs_video_kernel(const float *__restrict__ Qp, const float *__restrict__ Vp,
                                                               const int32_t *__restrict__ clip_off,
                                                               const int64_t *__restrict__ mom_off,
                                                               const float *__restrict__ rank_dist,
                                                               const int64_t *__restrict__ rank_idx, ScoreArgs a, MfmaArgs m)
{
    static_assert(NR == 2, "two rank keys per query (the bitmap holds one mask per key)");
    constexpr int ROW4 = FAST_D / 4;
    // VB = 1: 16 KB of LDS and ~135 VGPRs: ten waves per CU -- the kernel is a chain of dependent gathers (bitmap -> query rows ->
    // bound table) around 7 us of arithmetic per batch, and lives on the waves it has in flight
    __shared__ __attribute__((aligned(16))) float vst[(VB * NT + (VB > 1 ? 3 : 0)) * FAST_D];   // clip rows of this wave's video(s)
    __shared__ unsigned short list[PV_LIST];                             // slot << 8 | lane bit << 2 | key mask
    __shared__ float dx[64 * (NT + 1)];                                  // exact distances [pair][clip] (partial sums first)
    const int lane = threadIdx.x, v = blockIdx.x * VB;                   // first video of this wave
    const int groups = a.num_groups;
    const int nvid = VB > 1 ? (a.Nv - v < VB ? a.Nv - v : VB) : 1;
    const int c0 = clip_off[v], rows_total = clip_off[v + nvid] - c0;
    const int n = VB > 1 ? a.ds_rows : rows_total;                       // VB > 1: the bank's largest clip count bounds the chain loop
    const int64_t mbase = mom_off[v];
    bool staged = false;
    const float2v e2 = {a.eps, a.eps};
    unsigned long long npairs = 0;
    for (int gb = 0; gb < (VB > 1 ? 1 : groups); gb += 64) {
        // ---- the bitmaps: lane = group gb + lane of the video (VB = 1), or slot (video, group) of the wave's videos (VB > 1:
        // the host guarantees VB * groups <= 64) ----
        const int g = VB > 1 ? lane % groups : gb + lane;
        const bool slot_ok = VB > 1 ? lane < nvid * groups : g < groups;
        ulonglong2 mm = ulonglong2{0ull, 0ull};
        if (slot_ok && !m.fallback[g]) mm = m.amb[(size_t)v * groups + (VB > 1 ? lane : g)];
        const unsigned long long any = mm.x | mm.y;
        const int c = __builtin_popcountll(any);
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); incl += lane >= o ? t : 0; }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total == 0) continue;
        npairs += (unsigned long long)total;
        if (!staged) {                                                   // videos nobody marked cost one bitmap read
            const float4 *V4 = reinterpret_cast<const float4 *>(Vp);
            const int64_t v4_end = (int64_t)a.total_clips * ROW4;
            const int n4 = (VB > 1 ? rows_total : NT) * ROW4;            // (VB = 1 stages NT rows: the chain loop may touch the pad rows)
            for (int idx = lane; idx < n4; idx += 64) {
                const int64_t g4 = (int64_t)c0 * ROW4 + idx;
                reinterpret_cast<float4 *>(vst)[idx] = g4 < v4_end ? V4[g4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            staged = true;
        }
        for (int win = 0; win < total; win += PV_LIST) {
            // ---- expand: every lane writes the set bits of its slot that fall into the window ----
            __builtin_amdgcn_s_waitcnt(0xC07F);                          // the previous window's list reads are done
            __builtin_amdgcn_wave_barrier();
            {
                int pos = incl - c - win;
                unsigned long long rest = any;
                while (rest && pos < PV_LIST) {
                    const int b = __builtin_ctzll(rest);
                    rest &= rest - 1ull;
                    if (pos >= 0)
                        list[pos] = (unsigned short)((lane << 8) | (b << 2) | (unsigned)((mm.x >> b) & 1ull) | ((unsigned)((mm.y >> b) & 1ull) << 1));
                    ++pos;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            const int wtotal = total - win < PV_LIST ? total - win : PV_LIST;
            for (int base = 0; base < wtotal; base += 64) {
                const bool have = base + lane < wtotal;
                const unsigned ent = list[have ? base + lane : base];
                const unsigned rmask = have ? (ent & 3u) : 0u;
                const int slot = (int)(ent >> 8);
                const int vl = VB > 1 ? slot / groups : 0;                // this lane's video within the wave's range
                int64_t qi = (int64_t)(VB > 1 ? slot - vl * groups : gb + slot) * 64 + ((ent >> 2) & 63u);
                qi = qi < a.Nq ? qi : a.Nq - 1;
                // per-lane video geometry (VB = 1: the wave's one video)
                const int row0 = VB > 1 ? clip_off[v + vl] - c0 : 0;      // first LDS row of the lane's video
                const int n_l = VB > 1 ? clip_off[v + vl + 1] - clip_off[v + vl] : n;
                const int64_t mbase_l = VB > 1 ? mom_off[v + vl] : mbase;
                // ---- the canonical chains (clip rows broadcast from LDS), the query row in two parts of 13 and 12 float4: every
                // chain still runs k-ascending -- its accumulator waits in LDS between the parts -- and the row costs 52 registers
                // instead of 100.  Three clips at a time (a run-time loop: the code stays small), software-pipelined over k like
                // score_fast_kernel: the 3 broadcast reads of slice j4 + 1 are issued (and pinned by the sched_barrier) before the
                // 24 VALU ops of slice j4 -- left to the compiler the reads sat two instructions ahead of their use ----
                const float4 *q4 = reinterpret_cast<const float4 *>(Qp + qi * FAST_D);
                float *dl = dx + lane * (NT + 1);                        // this lane's partial sums, then distances
                const int ng = (n + 2) / 3;
                auto chain_part = [&](auto lo_c, auto hi_c, auto first_c) {
                    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value, NC = 3;
                    constexpr bool FIRST = decltype(first_c)::value;
                    float2v qp[2 * (HI - LO)];
#pragma unroll
                    for (int j4 = LO; j4 < HI; ++j4) {
#ifdef VFR_PV_SKIP_QLOAD
                        const float4 y = make_float4((float)(lane + j4), 1.0f, 2.0f, (float)qi);   // timing experiment: no gather
#else
                        const float4 y = q4[j4];
#endif
                        qp[2 * (j4 - LO)] = float2v{y.x, y.y}; qp[2 * (j4 - LO) + 1] = float2v{y.z, y.w};
                    }
#pragma nounroll
                    for (int g3 = 0; g3 < ng; ++g3) {
                        const float4 *vt = reinterpret_cast<const float4 *>(vst) + (row0 + g3 * NC) * ROW4;   // (VB = 1: wave-uniform, a broadcast)
                        float ac[NC];
#pragma unroll
                        for (int i = 0; i < NC; ++i) ac[i] = FIRST ? 0.0f : dl[g3 * NC + i];
                        float4 cur[NC], nxt[NC];
#pragma unroll
                        for (int i = 0; i < NC; ++i) cur[i] = vt[i * ROW4 + LO];
#pragma unroll
                        for (int j4 = LO; j4 < HI; ++j4) {
                            if (j4 + 1 < HI) {
#pragma unroll
                                for (int i = 0; i < NC; ++i) nxt[i] = vt[i * ROW4 + j4 + 1];
                            }
                            // (the first slice also waits for the query part: vmcnt 0; afterwards only LDS is outstanding)
                            if (j4 == LO) __builtin_amdgcn_s_waitcnt(0x0070 | (NC << 8));
                            else if (j4 + 1 < HI) __builtin_amdgcn_s_waitcnt(0xC07F | (NC << 8));
                            else __builtin_amdgcn_s_waitcnt(0xC07F);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < NC; ++i) {
#ifdef VFR_PV_SKIP_CHAIN
                                ac[i] += cur[i].x + qp[2 * (j4 - LO)].x;   // timing experiment: no chain
#else
                                const float2v d01 = (float2v{cur[i].x, cur[i].y} - qp[2 * (j4 - LO)]) + e2;
                                const float2v d23 = (float2v{cur[i].z, cur[i].w} - qp[2 * (j4 - LO) + 1]) + e2;
                                ac[i] = __builtin_fmaf(d01.x, d01.x, ac[i]);
                                ac[i] = __builtin_fmaf(d01.y, d01.y, ac[i]);
                                ac[i] = __builtin_fmaf(d23.x, d23.x, ac[i]);
                                ac[i] = __builtin_fmaf(d23.y, d23.y, ac[i]);
#endif
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < NC; ++i) cur[i] = nxt[i];
                        }
#pragma unroll
                        for (int i = 0; i < NC; ++i) dl[g3 * NC + i] = HI == ROW4 ? __builtin_sqrtf(ac[i]) : ac[i];
                    }
                };
                chain_part(std::integral_constant<int, 0>{}, std::integral_constant<int, 13>{}, std::true_type{});
                chain_part(std::integral_constant<int, 13>{}, std::integral_constant<int, ROW4>{}, std::false_type{});
                __builtin_amdgcn_s_waitcnt(0xC07F);
                float d[NT];
#pragma unroll
                for (int cc = 0; cc < NT; ++cc) d[cc] = cc < n_l ? dl[cc] : __builtin_inff();
                // ---- exact triangle: bits(sum) against the table's exclusive bounds (score < x <=> bits < LOX) ----
                // (the 42 bound pairs are requested here, after the query row's 100 registers are dead: hoisted above the chains
                // they would not fit)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                const unsigned *tq = m.tab + (size_t)qi * NR * NT * MF_TAB;
                int cntr[NR] = {0, 0};
                bool tie = false;
                float sums[NT];
#pragma unroll
                for (int L = 1; L <= NT; ++L) {
                    unsigned lox[NR], dhl[NR], below[NR], umin[NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
#ifdef VFR_PV_SKIP_TAB
                        const uint2 e = make_uint2(0x3F800000u + (unsigned)(L * 65536 + r), 0x3F800000u + (unsigned)(L * 65536 + r));   // timing experiment
#else
                        const uint2 e = *reinterpret_cast<const uint2 *>(tq + (r * NT + (L - 1)) * MF_TAB);
#endif
                        lox[r] = e.x; dhl[r] = e.y - e.x; below[r] = 0u; umin[r] = 0xFFFFFFFFu;
                    }
#ifdef VFR_PV_SKIP_TRI
                    if (L > 1) continue;                                 // timing experiment: one level of the triangle
#endif
#pragma unroll
                    for (int s2 = 0; s2 + L <= NT; ++s2) {
                        const float de = d[s2 + L - 1];
                        const float sum = L == 1 ? de : sums[s2] + de;
                        sums[s2] = sum;
                        const unsigned sb = __float_as_uint(sum);
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            const unsigned u = sb - lox[r];
                            below[r] = __builtin_amdgcn_alignbit(below[r], u, 31);
                            umin[r] = u < umin[r] ? u : umin[r];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        cntr[r] += __builtin_popcount(below[r]);
                        tie = tie || umin[r] < dhl[r];
                    }
                }
                if (__ballot(tie && rmask != 0u)) {
                    // bits(sum) inside [LOX, HIX) somewhere: score == a key distance (every query's own video gets here: the key
                    // IS one of its moments); this lane's video once more with the quotients, ties broken by moment id (the
                    // distances go through LDS: the walk indexes them dynamically)
#pragma unroll
                    for (int cc = 0; cc < NT; ++cc) dx[lane * (NT + 1) + cc] = d[cc];
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    if (tie) {
                        float xk[NR];
                        unsigned ik[NR];
#pragma unroll
                        for (int r = 0; r < NR; ++r) { xk[r] = rank_dist[r * a.Nq + qi]; ik[r] = (unsigned)rank_idx[r * a.Nq + qi]; }
#pragma nounroll
                        for (int s2 = 0; s2 < n_l; ++s2) {
                            float sum = 0.0f;
#pragma nounroll
                            for (int e = s2; e < n_l; ++e) {
                                const float de = dx[lane * (NT + 1) + e];
                                sum = e == s2 ? de : sum + de;
                                const float sc = sum / (float)(e - s2 + 1);
                                const unsigned id = (unsigned)(a.id_base + mbase_l + moment_index(n_l, s2, e));
#pragma unroll
                                for (int r = 0; r < NR; ++r)
                                    if (sc == xk[r] && id < ik[r]) cntr[r] += 1;
                            }
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                }
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    if (((rmask >> r) & 1u) && cntr[r]) atomicAdd(m.cnt_ws + r * a.Nq + qi, (unsigned long long)cntr[r]);
            }
        }
    }
    if (lane == 0 && npairs) atomicAdd(m.pairs_total, npairs);
}

// ---------------------------------------------------------------------------------------------------------------------
// finisher: one wave per query.  The k' approximate candidates (sorted keys, KEY_MAX padded) are re-scored with the
// canonical chain, sorted by exact (score, id) and the k best written out; the containment condition of the header is
// checked on the approximate keys and a failing query flags its group for the exact kernels.
// ---------------------------------------------------------------------------------------------------------------------
template <int KPL>
__global__ __launch_bounds__(256) void topk_finish_kernel(const unsigned long long *__restrict__ cand, int kp, int k,
                                                          const float *__restrict__ Q, int64_t Nq, const float *__restrict__ V,
                                                          const int32_t *__restrict__ clip_off, const int64_t *__restrict__ mom_off,
                                                          int Nv, int uniform_n, int D, float eps, int64_t id_base,
                                                          const float4 *__restrict__ qmeta, int *__restrict__ fallback,
                                                          const int64_t *__restrict__ thr_seed, float *__restrict__ out_dist,
                                                          int64_t *__restrict__ out_idx, int check)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + wv;
    if (q >= Nq) return;
    const float *qr = Q + q * D;
    __shared__ __attribute__((aligned(16))) float qs_all[4][FAST_D];
    float *qs = qs_all[wv];
    if (D == FAST_D) {
        for (int kk = lane; kk < FAST_D; kk += 64) qs[kk] = qr[kk];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
    unsigned long long key[KPL];
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const int e = i * 64 + lane;
        key[i] = KEY_MAX;
        const unsigned long long ak = e < kp ? cand[q * kp + e] : KEY_MAX;
        if (ak == KEY_MAX) continue;
        const unsigned local = (unsigned)(ak & 0xffffffffull) - (unsigned)id_base;
        int v, mloc, n;
        if (uniform_n > 0) {
            n = uniform_n;
            const unsigned M = (unsigned)(n * (n + 1) / 2);
            v = (int)(local / M);
            mloc = (int)(local - (unsigned)v * M);
        } else {
            int lo = 0, hi = Nv;                                        // largest v with mom_off[v] <= local
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (mom_off[mid] <= (int64_t)local) lo = mid; else hi = mid; }
            v = lo;
            mloc = (int)((int64_t)local - mom_off[v]);
            n = clip_off[v + 1] - clip_off[v];
        }
        int s = mloc, e2 = mloc;
        if (mloc >= n) {
            int rest = mloc - n;
            s = 0;
            while (rest >= n - 1 - s) { rest -= n - 1 - s; ++s; }
            e2 = s + 1 + rest;
        }
        const int c0 = clip_off[v];
        float sum = 0.0f;
        for (int c = s; c <= e2; ++c) {
            const float *vr = V + (int64_t)(c0 + c) * D;
            float acc = 0.0f;
            if (D == FAST_D) {                                            // 16-byte loads, five in flight; q from the wave's LDS copy
                const float4 *v4 = reinterpret_cast<const float4 *>(vr), *q4 = reinterpret_cast<const float4 *>(qs);
#pragma unroll 5
                for (int j4 = 0; j4 < FAST_D / 4; ++j4) {
                    const float4 x = v4[j4], y = q4[j4];
                    float dd = (x.x - y.x) + eps; acc = __builtin_fmaf(dd, dd, acc);
                    dd = (x.y - y.y) + eps; acc = __builtin_fmaf(dd, dd, acc);
                    dd = (x.z - y.z) + eps; acc = __builtin_fmaf(dd, dd, acc);
                    dd = (x.w - y.w) + eps; acc = __builtin_fmaf(dd, dd, acc);
                }
            } else {
                for (int kk = 0; kk < D; ++kk) {
                    const float dd = (vr[kk] - qr[kk]) + eps;
                    acc = __builtin_fmaf(dd, dd, acc);
                }
            }
            const float dc = __builtin_sqrtf(acc);
            sum = c == s ? dc : sum + dc;
        }
        key[i] = make_key(sum / (float)(e2 - s + 1), (unsigned)(ak & 0xffffffffull));
    }
    if (check) {
        // containment: s~_(k') > s~_(k) + 2 delta with delta = E2 / s~_(1) + (58 + 2 L) u s~  (L <= 64), or a short list
        const unsigned long long a1 = cand[q * kp], ak = k - 1 < kp ? cand[q * kp + (k - 1)] : KEY_MAX, aK = cand[q * kp + kp - 1];
        if (aK != KEY_MAX && ak != KEY_MAX) {
            const float s1 = __uint_as_float((unsigned)(a1 >> 32)), sk = __uint_as_float((unsigned)(ak >> 32)),
                        sK = __uint_as_float((unsigned)(aK >> 32));
            const float E2 = qmeta[q].z;
            const float delta = (s1 > 0.0f ? E2 / s1 : __builtin_inff()) + MF_U * qmeta[q].w * 1.01f + 190.0f * MF_U * sK * 1.02f;
            if (!(sK > sk + 2.0f * delta * 1.0001f) && lane == 0) fallback[q >> 6] = 1;
        }
        if (thr_seed && a1 != KEY_MAX) {
            // a seeded pass took its threshold margin for clips no closer than half the seed score (mfma_seed_kernel)
            const unsigned long long sd = (unsigned long long)thr_seed[q];
            const float x = __uint_as_float((unsigned)(sd >> 32)), s1 = __uint_as_float((unsigned)(a1 >> 32));
            if (sd < KEY_EMPTY && x < __builtin_inff() && !(s1 >= 0.5f * x) && lane == 0) fallback[q >> 6] = 1;
        }
    }
    wave_sort<KPL>(key, lane);
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const int e = i * 64 + lane;
        if (e < k) {
            const bool ok = key[i] != KEY_MAX;
            out_dist[q * k + e] = ok ? __uint_as_float((unsigned)(key[i] >> 32)) : __builtin_inff();
            out_idx[q * k + e] = ok ? (int64_t)(key[i] & 0xffffffffull) : -1;
        }
    }
}

// count_lt += the call's counts, for the query groups the approximate path answered
__global__ __launch_bounds__(256) void mfma_commit_counts_kernel(const unsigned long long *__restrict__ cnt_ws, int NR,
                                                                 int64_t Nq, const int *__restrict__ fallback,
                                                                 int64_t *__restrict__ count_lt)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)NR * Nq) return;
    const int64_t q = i % Nq;
    if (fallback && fallback[q >> 6]) return;
    count_lt[i] += (int64_t)cnt_ws[i];
}
