// a10 + a12: moment scoring and ranking (model/evaluate.py:49-80, evaluate_single.py:48-54).
//
//   dist[c]    = || (V[c] - q) + eps ||_2          F.pairwise_distance, eps on the difference (Q8)
//   score(s,e) = (dist[s] + ... + dist[e]) / (e-s+1)   index_select().mean()
//   ranking    = ascending (score, global moment id)    np.argsort, made deterministic
//
// Mapping onto gfx950.  Three families of kernels live here:
//   * score_kernel (generic shapes, any D, up to 64 clips): thread = query, V read per lane through L1, the clip distances
//     of the current video in a per-wave LDS column, candidates compacted by a cooperative register bitonic sort.
//   * score_fast_kernel (D = 100, <= 6 or exactly <= 21 clips): exact fp32 on the VALU -- the direct-difference form
//     ((v - q) + eps)^2 is not a GEMM.  thread = query with the query embedding in 50 float2 VGPR pairs; the clip rows being
//     scored are wave-uniform, staged per wave into LDS with coalesced 16-byte loads and read back as broadcast ds_read_b128
//     (no barrier, a wave never waits for another); branch-free moment triangle on register sums with thresholds as integer
//     bounds on the fp32 bits of the SUM (no division, no key on the common path).  Details at the kernel.
//   * score_mfma.h (included below): the MFMA pre-filter -- approximate distances from the GEMM form on the matrix cores,
//     three-way threshold tests with a rigorous error margin, exact re-scoring of whatever the margin cannot decide.
//   * keys are (fp32 bits of score << 32 | moment id): scores are >= 0 so unsigned order == the (score, id) lexicographic
//     order, one 64-bit compare per test.
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "vfr_common.h"
#include "vfr_math.h"

namespace vfr {

constexpr int NMAX_FUSED = 64;    // max clips per video on the fused path
constexpr int NMAX_DENSE = 128;   // max clips per video on the dense / own paths
constexpr unsigned long long KEY_MAX = ~0ull;
constexpr unsigned long long KEY_EMPTY = 0x7F800000FFFFFFFFull;   // (+inf, max id): an unused slot of an exchanged list
constexpr int MAX_RANK = 4;       // rank keys per query counted in one pass (IoU thresholds)

// ------------------------------------------------------------------------------------------------
// distance of NC (<= 4) clips to this lane's query: D is a template constant when DT > 0 (query in
// registers), runtime when DT == 0 (query re-read from global, L1-resident).
// ------------------------------------------------------------------------------------------------
template <int DT, int NC>
__device__ __forceinline__ void clip_dist4(const float *__restrict__ V, int64_t crow, int D, float eps,
                                           const float *qreg, const float *__restrict__ qrow, float *out)
{
    float acc[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = 0.0f;
    const float *v = V + crow * (DT > 0 ? DT : D);
    if constexpr (DT > 0) {
#pragma unroll
        for (int k = 0; k < DT; ++k) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                float d = (v[i * DT + k] - qreg[k]) + eps;
                acc[i] = __builtin_fmaf(d, d, acc[i]);
            }
        }
    } else {
        for (int k = 0; k < D; ++k) {
            float qk = qrow[k];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                float d = (v[(int64_t)i * D + k] - qk) + eps;
                acc[i] = __builtin_fmaf(d, d, acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) out[i] = __builtin_sqrtf(acc[i]);
}

// all n clip distances of one video into the wave's LDS column ds[c*64 + lane]
template <int DT>
__device__ __forceinline__ void video_distances(const float *__restrict__ V, int c0, int n, int D, float eps,
                                                const float *qreg, const float *__restrict__ qrow, float *ds,
                                                int lane)
{
    int c = 0;
    for (; c + 4 <= n; c += 4) {
        float d[4];
        clip_dist4<DT, 4>(V, c0 + c, D, eps, qreg, qrow, d);
#pragma unroll
        for (int i = 0; i < 4; ++i) ds[(c + i) * 64 + lane] = d[i];
    }
    for (; c < n; ++c) {
        float d[1];
        clip_dist4<DT, 1>(V, c0 + c, D, eps, qreg, qrow, d);
        ds[c * 64 + lane] = d[0];
    }
}

// value of lane (lane ^ STRIDE): within a row of 16 lanes through the DPP data path (one or two v_mov_b32_dpp per dword, a
// few cycles), across rows through ds_bpermute (an LDS round trip).  26 of the 33 cross-lane stages of the 256-key sort
// have STRIDE <= 8, and the stages depend on each other: the round trips were the sort's time.
template <int STRIDE>
__device__ __forceinline__ unsigned xor_lane_u32(unsigned x)
{
    if constexpr (STRIDE == 1) return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
    else if constexpr (STRIDE == 2) return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, false); // quad_perm [2,3,0,1]
    else if constexpr (STRIDE == 4) {
        const int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xF, 0x5, false);       // row_shl:4 into banks 0, 2 (lane <- lane + 4)
        return (unsigned)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xF, 0xA, false);          // row_shr:4 into banks 1, 3 (lane <- lane - 4)
    } else if constexpr (STRIDE == 8) {
        const int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x108, 0xF, 0x3, false);       // row_shl:8 into banks 0, 1
        return (unsigned)__builtin_amdgcn_update_dpp(t, (int)x, 0x118, 0xF, 0xC, false);          // row_shr:8 into banks 2, 3
    } else return (unsigned)__shfl_xor((int)x, STRIDE, 64);
}
template <int STRIDE>
__device__ __forceinline__ unsigned long long xor_lane_u64(unsigned long long x)
{
    return ((unsigned long long)xor_lane_u32<STRIDE>((unsigned)(x >> 32)) << 32) | xor_lane_u32<STRIDE>((unsigned)x);
}

// ------------------------------------------------------------------------------------------------
// wave-cooperative bitonic sort of KPL*64 keys, element e = i*64 + lane, ascending
// ------------------------------------------------------------------------------------------------
template <int KPL>
__device__ __forceinline__ void wave_sort(unsigned long long (&key)[KPL], int lane)
{
    constexpr int CAP = KPL * 64;
#pragma unroll
    for (int size = 2; size <= CAP; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            if (stride >= 64) {
                const int j = stride >> 6;
#pragma unroll
                for (int i = 0; i < KPL; ++i) {
                    if ((i & j) == 0) {
                        const bool asc = size >= CAP ? true : ((i & (size >> 6)) == 0);
                        unsigned long long a = key[i], b = key[i | j];
                        unsigned long long lo = a < b ? a : b, hi = a < b ? b : a;
                        key[i] = asc ? lo : hi;
                        key[i | j] = asc ? hi : lo;
                    }
                }
            } else {
                const bool lower = (lane & stride) == 0;
#pragma unroll
                for (int i = 0; i < KPL; ++i) {
                    bool asc;
                    if (size >= CAP) asc = true;
                    else if (size >= 64) asc = (i & (size >> 6)) == 0;
                    else asc = (lane & size) == 0;
                    // the lane keeps its key when (it wants the smaller one) == (its key is the smaller one): one 64-bit
                    // compare and two selects (min / max / select took ten vector instructions per key and stage)
                    const unsigned long long other = stride == 1 ? xor_lane_u64<1>(key[i]) : stride == 2 ? xor_lane_u64<2>(key[i])
                                                   : stride == 4 ? xor_lane_u64<4>(key[i]) : stride == 8 ? xor_lane_u64<8>(key[i])
                                                   : stride == 16 ? xor_lane_u64<16>(key[i]) : xor_lane_u64<32>(key[i]);
                    const bool keep = (lower == asc) == (key[i] < other);
                    key[i] = keep ? key[i] : other;
                }
            }
        }
    }
}

template <int KPL>
__device__ __forceinline__ unsigned long long key_at(const unsigned long long (&key)[KPL], int e)
{
    unsigned long long r = 0;
    const int i = e >> 6, l = e & 63;
#pragma unroll
    for (int j = 0; j < KPL; ++j)
        if (j == i) r = __shfl(key[j], l, 64);
    return r;
}

// compact the candidate buffer of lane L (wave-uniform): keep the k smallest keys, sorted.
// returns the new count; *thr_out = k-th smallest key when at least k keys exist.
template <int KPL>
__device__ __attribute__((noinline)) int compact_buffer(unsigned long long *base, int cnt, int k, int lane,
                                              unsigned long long *thr_out)
{
    unsigned long long key[KPL];
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        int e = i * 64 + lane;
        key[i] = e < cnt ? base[e] : KEY_MAX;
    }
    wave_sort<KPL>(key, lane);
    const int keep = cnt < k ? cnt : k;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        int e = i * 64 + lane;
        if (e < keep) base[e] = key[i];
    }
    if (cnt >= k) *thr_out = key_at<KPL>(key, k - 1);
    return keep;
}

struct MfmaArgs;

// read-only inputs are separate `const __restrict__` kernel parameters (not members of this struct):
// only then does the backend know they are never written by the kernel and may use the (non-coherent)
// scalar cache for the wave-uniform V / offset loads.
struct ScoreArgs {
    const float *Q; int64_t Nq;
    const float *V; const int32_t *clip_off; const int64_t *mom_off;
    int Nv, D; float eps;
    int64_t id_base;
    int k;
    int num_rank;                                     // rank keys per query (0..MAX_RANK)
    const float *rank_dist; const int64_t *rank_idx; int64_t *count_lt;
    float *scores; int64_t total_moments;            // dense mode
    unsigned long long *buf; int *buf_cnt;            // [tasks*64][CAP], [tasks*64]
    unsigned long long *thr_global;                   // [Nq]
    int num_groups, num_chunks;
    int ds_rows;                                      // LDS rows per wave = max clips per video
    int total_clips;                                  // rows of V (fast kernel: bound for the staged reads)
    int min_clips;                                    // smallest clip count in the bank (0 = unknown)
    int v_lo, v_hi;                                   // video range scored by this launch (chunks partition it)
    int force_generic;                                // 1: use score_kernel (cooperative compaction) even if fast applies
    int prof_site;                                    // profiler site of this launch (0: chosen from the mode)
    int keep_all;                                     // 1: hand every appended key to the merge (no final per-column cut)
    int level_cap;                                    // > 0: only moments of at most this many clips are appended (stage A)
    const int *group_mask;                            // != null: only query groups g with group_mask[g] != 0 are answered (MFMA fallback)
    const struct MfmaArgs *mf_host;                   // HOST pointer, != null: launch the MFMA pre-filter kernel (score_mfma.h)
    int mf_bf16;
#ifdef VFR_SCORE_STAMPS
    unsigned long long *stamps;                       // debug build: per-phase s_memtime totals of the fused kernel
#endif
};

// MODE 0: dense scores; MODE 1: fused top-k and/or rank counting
template <int DT, int MODE, int KPL>
__global__ __launch_bounds__(256) void score_kernel(const float *__restrict__ Qp, const float *__restrict__ Vp,
                                                    const int32_t *__restrict__ clip_off,
                                                    const int64_t *__restrict__ mom_off,
                                                    const float *__restrict__ rank_dist,
                                                    const int64_t *__restrict__ rank_idx, ScoreArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CAP = KPL * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = blockIdx.x * 4 + wave;
    if (task >= a.num_groups * a.num_chunks) return;
    const int chunk = task / a.num_groups, group = task - chunk * a.num_groups;
    if (a.group_mask && !a.group_mask[group]) return;
    float *ds = smem + (size_t)wave * a.ds_rows * 64;

    const int64_t qi = (int64_t)group * 64 + lane;
    const bool active = qi < a.Nq;
    const int64_t qrow_i = active ? qi : a.Nq - 1;
    const float *__restrict__ qrow = Qp + qrow_i * a.D;
    float qreg[DT > 0 ? DT : 1];
    if constexpr (DT > 0) {
#pragma unroll
        for (int kk = 0; kk < DT; ++kk) qreg[kk] = qrow[kk];
    }

    const int v0 = a.v_lo + (int)((int64_t)(a.v_hi - a.v_lo) * chunk / a.num_chunks);
    const int v1 = a.v_lo + (int)((int64_t)(a.v_hi - a.v_lo) * (chunk + 1) / a.num_chunks);

    // selection state (MODE 1)
    unsigned long long thr = KEY_MAX, kstar[MAX_RANK] = {0, 0, 0, 0};
    int cnt = 0, nlt[MAX_RANK] = {0, 0, 0, 0};
    unsigned long long *mybuf = nullptr;
    const bool want_topk = MODE == 1 && a.k > 0;
    const bool want_rank = MODE == 1 && rank_dist != nullptr;
    if (MODE == 1) {
        if (want_topk) mybuf = a.buf + ((size_t)task * 64 + lane) * CAP;
        if (want_rank && active) {
#pragma unroll
            for (int r = 0; r < MAX_RANK; ++r)
                if (r < a.num_rank) kstar[r] = make_key(rank_dist[r * a.Nq + qi], (unsigned)rank_idx[r * a.Nq + qi]);
        }
    }

    for (int v = v0; v < v1; ++v) {
        const int c0 = clip_off[v], n = clip_off[v + 1] - c0;
        const int64_t mbase = mom_off[v];
        video_distances<DT>(Vp, c0, n, a.D, a.eps, qreg, qrow, ds, lane);
        if (MODE == 1 && want_topk && active) {    // pick up thresholds published by other waves
            unsigned long long g = __hip_atomic_load(a.thr_global + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            thr = g < thr ? g : thr;
        }
        for (int s = 0; s < n; ++s) {
            if (MODE == 1 && want_topk) {
                // this start can append up to n - s keys per lane: make room first (wave-uniform)
                unsigned long long need = __ballot(cnt > CAP - (n - s));
                while (need) {
                    const int L = __builtin_ctzll(need);
                    need &= need - 1;
                    __threadfence_block();
                    unsigned long long nthr = KEY_MAX;
                    const int cL = __shfl(cnt, L, 64);
                    unsigned long long *bL = a.buf + ((size_t)task * 64 + L) * CAP;
                    const int keep = compact_buffer<KPL>(bL, cL, a.k, lane, &nthr);
                    __threadfence_block();
                    if (lane == L) {
                        cnt = keep;
                        if (nthr < thr) {
                            thr = nthr;
                            __hip_atomic_fetch_min(a.thr_global + qi, thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
            float sum = 0.0f;
            for (int e = s; e < n; ++e) {
                const float de = ds[e * 64 + lane];
                sum = e == s ? de : sum + de;
                const float sc = sum / (float)(e - s + 1);
                const int local = moment_index(n, s, e);
                if (MODE == 0) {
                    if (active) a.scores[qi * a.total_moments + mbase + local] = sc;
                } else {
                    const unsigned long long key = make_key(sc, (unsigned)(a.id_base + mbase + local));
                    if (want_rank) {
#pragma unroll
                        for (int r = 0; r < MAX_RANK; ++r) nlt[r] += key < kstar[r] ? 1 : 0;   // kstar = 0 when unused
                    }
                    if (want_topk && active && key < thr) mybuf[cnt++] = key;
                }
            }
        }
    }

    if (MODE == 1) {
        if (want_rank && active) {
#pragma unroll
            for (int r = 0; r < MAX_RANK; ++r)
                if (r < a.num_rank && nlt[r])
                    atomicAdd(reinterpret_cast<unsigned long long *>(a.count_lt + r * a.Nq + qi), (unsigned long long)nlt[r]);
        }
        if (want_topk) {
            // final: every lane's buffer sorted and cut to k (the merge kernel reads the first cnt keys)
            // cut every over-full buffer to its k best (the merge kernel sorts, so order is free)
            unsigned long long need = __ballot(cnt > a.k);
            while (need) {
                const int L = __builtin_ctzll(need);
                need &= need - 1;
                __threadfence_block();
                unsigned long long nthr = KEY_MAX;
                const int cL = __shfl(cnt, L, 64);
                unsigned long long *bL = a.buf + ((size_t)task * 64 + L) * CAP;
                const int keep = compact_buffer<KPL>(bL, cL, a.k, lane, &nthr);
                if (lane == L) cnt = keep;
            }
            __threadfence_block();
            a.buf_cnt[(size_t)task * 64 + lane] = active ? cnt : 0;
        }
    }
}


// ================================================================================================
// Fast fused kernel (D = 100, every video has at most NT clips; NT in {6, 21}; NR = rank keys per query).
//
// thread = query: a wave owns 64 queries, each lane keeps its query embedding in 50 float2 VGPR pairs.
//   * V delivery: the clip rows being scored are the same for all 64 lanes.  Each wave stages the next NC
//     rows (NC * 400 contiguous bytes of V) into a private LDS buffer with coalesced 16-byte vector loads
//     (issued one group ahead, held in registers while the current group is computed) and reads them back as
//     broadcast ds_read_b128 -- all lanes the same address, so no bank conflicts and no barrier: a wave never
//     synchronises with another wave.  (Streaming V through the scalar cache was measured first: s_load returns
//     out of order, so every 64-byte batch pays a full lgkmcnt(0) round trip -- 10x off the VALU rate.)
//   * (v - q), (+ eps) are v_pk_add_f32 on float2 pairs; the square-accumulate is a sequential v_fma chain per
//     clip (canonical order), NC independent chains in flight.
//   * moment triangle L-outer on the wave's LDS distance column: sums[s] += d[s+L-1] is exactly the canonical
//     left-to-right sum of d[s..s+L-1]; both loops are fully unrolled so sums[] and the threshold tables are
//     statically indexed registers.
//   * thresholds live in SUM space, so the common path has NO division and builds NO key.  For a bound x and
//     span length L:  hi(x,L) = max{S : fl(S/L) <= x},  lo(x,L) = max{S : fl(S/L) < x}  (fl(S/L) is monotone
//     in S; found by walking a few ulps around fl(x*L)).  Then
//         score <  x  <=>  sum <= lo        (rank count, exact)
//         score == x  <=>  lo < sum <= hi   (a tie: resolved by moment id on the exact path)
//         score <= thr => sum <= hi(thr)    (top-k candidate, conservative superset)
//   * only when some lane of the wave has a candidate or a tie is the video re-walked exactly (IEEE division,
//     64-bit keys, buffer appends) -- rare once the shared threshold has tightened.
// ================================================================================================
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int FAST_D = 100;

__device__ __forceinline__ float next_up(float x) { return __uint_as_float(__float_as_uint(x) + 1u); }
__device__ __forceinline__ float next_down(float x) { return __uint_as_float(__float_as_uint(x) - 1u); }

// largest S >= 0 with fl(S/L) < x (STRICT) or <= x;  -1 when there is none;  +inf when x is +inf / NaN
template <bool STRICT>
__device__ __attribute__((noinline)) float sum_bound(float x, int L)
{
    const float inf = __builtin_inff();
    if (!(x < inf)) return inf;
    const float Lf = (float)L;
    float S = x * Lf;
    if (!(S < inf)) S = 3.4028234663852886e38f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float u = next_up(S), y = u / Lf;
        const bool ok = (STRICT ? (y < x) : (y <= x)) && (u < inf);
        S = ok ? u : S;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float y = S / Lf;
        const bool ok = STRICT ? (y < x) : (y <= x);
        S = ok ? S : (S > 0.0f ? next_down(S) : -1.0f);
    }
    const float y = S / Lf;
    return (STRICT ? (y < x) : (y <= x)) ? S : -1.0f;
}


// Lane-parallel tightening of all 64 candidate columns of a task (layout col[slot * 64 + lane]).
// Every lane bisects on the fp32 bit pattern of the distance for a bound t with count(dist <= t) >= k over ITS
// column (any such t is a valid filter: at least k moments are that good, so nothing above t can be in the
// final top-k), drops the entries above t in place and publishes (t, id = max) as its threshold.  No sorting and
// no cross-lane traffic; all loads are coalesced across the wave.  The merge kernel does the exact ordering.
__device__ __attribute__((noinline)) void lane_tighten(unsigned long long *col, int lane, int k, int *cnt_io,
                                                       unsigned long long *thr_io)
{
    const int cnt = *cnt_io;
    int cmax = cnt;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(cmax, off, 64); cmax = o > cmax ? o : cmax; }
    if (cmax <= k) return;
    unsigned lo = 0xFFFFFFFFu, hi = 0u;                       // min / max distance bits of this lane's column
    for (int i = 0; i < cmax; ++i) {
        if (i < cnt) {
            const unsigned d = (unsigned)(col[(size_t)i * 64 + lane] >> 32);
            lo = d < lo ? d : lo;
            hi = d > hi ? d : hi;
        }
    }
    const bool mine = cnt > k;                                // lanes at or below k keep everything
    // invariant: count(<= hi) >= k.  count(<= lo-1) = 0 < k.
    for (int it = 0; it < 16; ++it) {
        if (__ballot(mine && lo < hi) == 0) break;
        const unsigned mid = lo + ((hi - lo) >> 1);
        int c = 0;
        for (int i = 0; i < cmax; ++i)
            if (i < cnt) c += ((unsigned)(col[(size_t)i * 64 + lane] >> 32) <= mid) ? 1 : 0;
        if (mine && lo < hi) { if (c >= k) hi = mid; else lo = mid + 1; }
    }
    if (mine) {
        int w = 0;
        for (int i = 0; i < cnt; ++i) {
            const unsigned long long key = col[(size_t)i * 64 + lane];
            if ((unsigned)(key >> 32) <= hi) { col[(size_t)w * 64 + lane] = key; ++w; }
        }
        *cnt_io = w;
        const unsigned long long t = ((unsigned long long)hi << 32) | 0xFFFFFFFFull;
        if (t < *thr_io) *thr_io = t;
    }
}

template <int NT> struct FastCfg { static constexpr int NC = 3; };   // 6 = 2 x 3 and 21 = 7 x 3: no wasted clip slots

// Thresholds are kept as EXCLUSIVE upper bounds on the fp32 bit pattern of the (non-negative) sum:
//   score <  x  <=>  bits(sum) < LOX,   score <= x  <=>  bits(sum) < HIX = LOX + delta   (delta in 0..3)
// "no such sum" is simply 0, +inf needs no special case, and hi costs 2 bits instead of a register.
__device__ __forceinline__ unsigned excl_bound(float s) { return s < 0.0f ? 0u : __float_as_uint(s) + 1u; }

template <int NT, int KPL, int NR, bool TOPK, bool EXACT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8)))
void score_fast_kernel(const float *__restrict__ Qp, const float *__restrict__ Vp, const int32_t *__restrict__ clip_off,
                       const int64_t *__restrict__ mom_off, const float *__restrict__ rank_dist,
                       const int64_t *__restrict__ rank_idx, ScoreArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CAP = KPL * 64, NC = FastCfg<NT>::NC, ROW4 = FAST_D / 4, G4 = NC * ROW4;   // float4 per group
    constexpr int NLD = (G4 + 63) / 64;                                                       // loads per lane
    constexpr int NRR = NR > 0 ? NR : 1;
    // n = 21 with rank keys keeps the per-length rank bounds in LDS ([r][L][lane] column, 2 ds_read per span length):
    // 42 registers fewer; 18.5 KB of LDS per wave, so 8 waves/CU still fit in 160 KB
    constexpr bool LOX_LDS = NT > 6 && NR > 0;
    const int lane = threadIdx.x;
    const int task = blockIdx.x;
    const int chunk = task / a.num_groups, group = task - chunk * a.num_groups;
    if (a.group_mask && !a.group_mask[group]) return;
    float *stage = smem;                              // [2][NC*100]  staged clip rows (double buffer)
    float *ds = smem + 2 * NC * FAST_D;               // [ceil(NT/NC)*NC][64] clip distances of the current video
    unsigned *lox_lds = reinterpret_cast<unsigned *>(ds + ((NT + NC - 1) / NC) * NC * 64);   // [NR][NT][64] when LOX_LDS

    const int64_t qi = (int64_t)group * 64 + lane;
    const bool active = qi < a.Nq;
    const float *__restrict__ qrow = Qp + (active ? qi : a.Nq - 1) * FAST_D;
    float2v qp[FAST_D / 2];
#pragma unroll
    for (int j = 0; j < FAST_D / 2; ++j) qp[j] = reinterpret_cast<const float2v *>(qrow)[j];

    const int v0 = a.v_lo + (int)((int64_t)(a.v_hi - a.v_lo) * chunk / a.num_chunks);
    const int v1 = a.v_lo + (int)((int64_t)(a.v_hi - a.v_lo) * (chunk + 1) / a.num_chunks);
    const int64_t v4_end = (int64_t)a.total_clips * ROW4;     // float4 count of V: never read past it

    unsigned long long thr = KEY_MAX;
    int cnt = 0, nlt[NRR] = {0};
    unsigned long long *col = TOPK ? a.buf + (size_t)task * 64 * CAP : nullptr;   // [slot][lane] candidate columns

    // threshold tables, statically indexed (registers): index L-1
    unsigned hix_t[TOPK ? NT : 1];                    // top-k filter: bits(sum) < hix_t  <=  score <= thr distance
    unsigned lox[LOX_LDS ? 1 : NRR][LOX_LDS ? 1 : NT];   // rank r: score <  key distance (registers unless LOX_LDS)
    unsigned long long dl[NRR];                       // rank r: 2-bit (HIX - LOX) per L, packed
    bool wide = false;                                // some delta did not fit 2 bits: treat every video as a tie
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float x = active ? rank_dist[r * a.Nq + qi] : 0.0f;
        dl[r] = 0;
#pragma unroll
        for (int L = 1; L <= NT; ++L) {
            // never above bits(+inf): a sum that overflowed (or the +inf padding of a short video) is not < any key
            const unsigned lo = active ? min(excl_bound(sum_bound<true>(x, L)), 0x7F800000u) : 0u;
            const unsigned hi = active ? min(excl_bound(sum_bound<false>(x, L)), 0x7F800000u) : 0u;
            const unsigned d = hi - lo;
            wide = wide || d > 3u;
            if (LOX_LDS) lox_lds[(r * NT + (L - 1)) * 64 + lane] = lo; else lox[r][L - 1] = lo;
            dl[r] |= (unsigned long long)(d & 3u) << (2 * (L - 1));
        }
    }
    if (TOPK) {
#pragma unroll
        for (int L = 1; L <= NT; ++L) hix_t[L - 1] = active ? 0xFFFFFFFFu : 0u;       // thr = +inf: everything passes
    }
    bool dirty = false;
    // Whole-video skip.  Every moment's score is a mean of clip distances, so score >= dmin * (1 - 21 * 2^-24) with
    // dmin = the video's smallest clip distance (rounding is monotone: the chain sum of L values >= dmin is >= the chain sum
    // of L copies of dmin, which is within (L-1) roundings of L * dmin; the division adds one more).  If dmin exceeds every
    // bound this lane compares scores with -- its rank keys' distances and its top-k threshold distance -- by that margin, the
    // video holds no candidate, nothing below a rank key and no tie for this query; when that is true for all 64 lanes the
    // moment triangle is skipped.  Only the rank-only instantiation (TOPK = false: what evaluate() runs) carries the test: with
    // a trained model its rank keys sit in the far tail of the distribution and most videos are skipped (9.7 -> 8.7 ms with keys
    // at the 50th / 100th best moment).  In the fused top-k launches the ladder's threshold is ~10x looser than the final
    // k-th key, some lane of 64 almost always has a candidate, and the test (+1.5 %) would be pure cost there.  (A second
    // triangle instantiation without the rank half, for videos whose rank keys clear but whose top-k threshold does not, was
    // measured too: -7 % for selective keys, +2 % on the bench from the extra register pressure: not kept.)
    float rank_bound = 0.0f;
#pragma unroll
    for (int r = 0; r < NR; ++r) { const float x = active ? rank_dist[r * a.Nq + qi] : 0.0f; rank_bound = x > rank_bound ? x : rank_bound; }
    constexpr float SKIP_MARGIN = 1.000004f;
    const float skipb = active ? rank_bound * SKIP_MARGIN : 0.0f;

    float4 pre[NLD];                                  // next group's rows, in flight while this group is computed
    auto gload = [&](int64_t row) {                   // rows [row, row+NC) of V -> registers (coalesced 16 B/lane)
#pragma unroll
        for (int t = 0; t < NLD; ++t) {
            const int idx = lane + 64 * t;
            const int64_t g4 = row * ROW4 + idx;
            pre[t] = (idx < G4 && g4 < v4_end) ? reinterpret_cast<const float4 *>(Vp)[g4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int t = 0; t < NLD; ++t) {
            const int idx = lane + 64 * t;
            if (idx < G4) reinterpret_cast<float4 *>(stage + buf * NC * FAST_D)[idx] = pre[t];
        }
    };

    int buf = 0;
    if (v0 < v1) { gload(clip_off[v0]); swrite(0); }

#ifdef VFR_SCORE_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
#define STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; }
#else
#define STAMP(i)
#endif
    // per-video offsets off the critical path: analytic when every video has NT clips, else loaded one video ahead
    int c_cur = v0 < v1 ? clip_off[v0] : 0, c_nxt = (!EXACT && v0 < v1) ? clip_off[v0 + 1] : 0;
    int64_t m_cur = v0 < v1 ? mom_off[v0] : 0;
    for (int v = v0; v < v1; ++v) {
        const int c0 = c_cur, n = EXACT ? NT : c_nxt - c_cur;
        const int64_t mbase = m_cur;
        if (EXACT) { c_cur += NT; m_cur += NT * (NT + 1) / 2; }
        else {
            c_cur = c_nxt;
            if (v + 1 < v1) { c_nxt = clip_off[v + 2]; m_cur = mom_off[v + 1]; }     // consumed next iteration
        }
        const int ng = (n + NC - 1) / NC;
        STAMP(0)
        if (TOPK) {
            // a video can append up to M = n(n+1)/2 keys per lane: make room first (CAP >= k + M by construction)
            if (__ballot(cnt > CAP - n * (n + 1) / 2)) {
                __threadfence_block();
                const unsigned long long before = thr;
                lane_tighten(col, lane, a.k, &cnt, &thr);
                __threadfence_block();
                if (thr < before) {
                    dirty = true;
                    if (active) __hip_atomic_fetch_min(a.thr_global + qi, thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (((v - v0) & 7) == 0 && active) {          // thresholds published by other waves
                const unsigned long long g = __hip_atomic_load(a.thr_global + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (g < thr) { thr = g; dirty = true; }
            }
            if (__ballot(dirty)) {
                const float x = thr == KEY_MAX ? __builtin_inff() : __uint_as_float((unsigned)(thr >> 32));
#pragma unroll
                for (int L = 1; L <= NT; ++L) hix_t[L - 1] = active ? excl_bound(sum_bound<false>(x, L)) : 0u;
                dirty = false;
            }
        }
        STAMP(1)
        // ---- clip distances, NC chains at a time, V rows broadcast from the staging buffer ----
#pragma nounroll
        for (int g = 0; g < ng; ++g) {
            const int64_t next_row = g + 1 < ng ? (int64_t)c0 + (g + 1) * NC : (int64_t)c0 + n;   // next video's c0
            gload(next_row);
            const float4 *vt = reinterpret_cast<const float4 *>(stage + buf * NC * FAST_D);
            const float2v e2 = {a.eps, a.eps};
            float acc[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) acc[i] = 0.0f;
            // software pipeline over k: the NC broadcast reads of slice j4+1 are issued (and pinned there by the
            // sched_barrier) before the 8*NC VALU ops of slice j4, so LDS latency hides under the arithmetic.
            float4 cur[NC], nxt[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) cur[i] = vt[i * ROW4];
#pragma unroll
            for (int j4 = 0; j4 < ROW4; ++j4) {
                if (j4 + 1 < ROW4) {
#pragma unroll
                    for (int i = 0; i < NC; ++i) nxt[i] = vt[i * ROW4 + j4 + 1];
                }
                // ONE wait per slice (LDS returns in order: all but the NC reads just issued have landed) instead of one
                // per clip: every instruction costs the wave an issue slot here
                if (j4 + 1 < ROW4) __builtin_amdgcn_s_waitcnt(0xC07F | (NC << 8)); else __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const float2v d01 = (float2v{cur[i].x, cur[i].y} - qp[2 * j4]) + e2;
                    const float2v d23 = (float2v{cur[i].z, cur[i].w} - qp[2 * j4 + 1]) + e2;
                    acc[i] = __builtin_fmaf(d01.x, d01.x, acc[i]);
                    acc[i] = __builtin_fmaf(d01.y, d01.y, acc[i]);
                    acc[i] = __builtin_fmaf(d23.x, d23.x, acc[i]);
                    acc[i] = __builtin_fmaf(d23.y, d23.y, acc[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NC; ++i) cur[i] = nxt[i];
            }
#pragma unroll
            for (int i = 0; i < NC; ++i) ds[(g * NC + i) * 64 + lane] = __builtin_sqrtf(acc[i]);   // rows >= n: unused
            buf ^= 1;
            swrite(buf);
        }
        // ---- moment triangle, L-outer, fully unrolled: sums[] and the tables are static registers ----
        // all n clip distances into registers in one burst (one wait), then the triangle touches no memory
        STAMP(2)
        float d[NT], sums[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) d[c] = (EXACT || c < n) ? ds[c * 64 + lane] : __builtin_inff();
        if (!TOPK) {
            float dmin = d[0];
#pragma unroll
            for (int c = 1; c < NT; ++c) dmin = d[c] < dmin ? d[c] : dmin;
            if (__ballot(!(dmin > skipb)) == 0) continue;          // no lane can see a score at or below any of its bounds
        }
        // The triangle is branch-free: per moment one add, and per rank key u = bits(sum) - LOX, whose sign bit IS
        // "score < key" (both operands <= bits(+inf)), shifted into a per-level bit collector (v_alignbit; one popcount per level).  The top-k filter and the tie test
        // need only each level's MINIMUM of bits(sum) resp. u (v_min3: half an instruction per moment), folded once per
        // level into a per-lane level mask resp. a tie counter.  Moments that reach past the video's last clip carry
        // +inf sums (d[c >= n] = +inf), which no bound admits: no guards.
        int ties = wide ? 1 : 0;
        unsigned lvl = 0;                               // bit (NT - L): some moment of length L may enter the top-k
#pragma unroll
        for (int L = 1; L <= NT; ++L) {
            static_assert(NT <= 32, "one 32-bit sign collector per level");
            unsigned lx[NRR], dlt[NRR], umin[NRR], below[NRR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                below[r] = 0u;
                lx[r] = LOX_LDS ? lox_lds[(r * NT + (L - 1)) * 64 + lane] : lox[LOX_LDS ? 0 : r][LOX_LDS ? 0 : L - 1];
                dlt[r] = (unsigned)(dl[r] >> (2 * (L - 1))) & 3u;
                umin[r] = 0xFFFFFFFFu;
            }
            unsigned tmin = 0xFFFFFFFFu;
#pragma unroll
            for (int s = 0; s + L <= NT; ++s) {
                const float de = d[s + L - 1];
                const float sum = L == 1 ? de : sums[s] + de;
                sums[s] = sum;
                const unsigned sb = __float_as_uint(sum);
                if (TOPK) tmin = sb < tmin ? sb : tmin;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const unsigned u = sb - lx[r];          // both <= 0x7F800000: bit 31 of u IS "bits(sum) < LOX"
                    below[r] = __builtin_amdgcn_alignbit(below[r], u, 31);   // shift that bit into the level's collector
                    umin[r] = u < umin[r] ? u : umin[r];
                }
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                nlt[r] += __builtin_popcount(below[r]);
                ties += umin[r] < dlt[r] ? 1 : 0;           // some LOX <= bits(sum) < HIX: score == key
            }
            if (TOPK) { unsigned t; lvl = lvl + lvl + (__builtin_sub_overflow(tmin, hix_t[L - 1], &t) ? 1u : 0u); }
            // pin this level's results here: without it the rank half is sunk below the later levels and every sum spills
            if (NR == 2) asm volatile("" : "+v"(nlt[0]), "+v"(nlt[NR > 1 ? 1 : 0]), "+v"(ties), "+v"(lvl));
            else asm volatile("" : "+v"(ties), "+v"(lvl));
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(3)
        if (TOPK && lvl != 0) {
            // Some moment may enter the top-k.  Two phases per flagged level, so that the common work stays in registers:
            // (1) the level's sums again from the 21 distances (one LDS burst for all levels; the same left-to-right sums)
            // -> a per-lane bit mask of the moments under the bound; (2) only those (usually one) are popped: exact score,
            // key, append.
            float dd[NT];
#pragma unroll
            for (int c = 0; c < NT; ++c) dd[c] = (EXACT || c < n) ? ds[c * 64 + lane] : __builtin_inff();
#pragma unroll
            for (int L = 1; L <= NT; ++L) {
                if (((lvl >> (NT - L)) & 1u) && (a.level_cap == 0 || L <= a.level_cap)) {
                    const unsigned hx = hix_t[L - 1];
                    unsigned pm = 0;
#pragma unroll
                    for (int s = 0; s + L <= NT; ++s) {
                        float sum = dd[s];
#pragma unroll
                        for (int e = s + 1; e < s + L; ++e) sum += dd[e];
                        pm |= (__float_as_uint(sum) < hx ? 1u : 0u) << s;
                    }
                    while (pm) {
                        const int s = __builtin_ctz(pm);
                        pm &= pm - 1u;
                        if (s + L <= n) {                // (+inf padding can pass only while the bound is still +inf)
                            float sum = ds[s * 64 + lane];
#pragma nounroll
                            for (int e = s + 1; e < s + L; ++e) sum += ds[e * 64 + lane];
                            const float sc = sum / (float)L;
                            const unsigned id = (unsigned)(a.id_base + mbase + moment_index(n, s, s + L - 1));
                            const unsigned long long key = make_key(sc, id);
                            if (key < thr) { col[(size_t)cnt * 64 + lane] = key; ++cnt; }
                        }
                    }
                }
            }
        }
        const bool tie = ties != 0;
        STAMP(4)
        if (NR == 0 || __ballot(tie) == 0) continue;

        // ---- some lane has score == a rank key: re-walk this video exactly and break the ties by moment id ----
#pragma nounroll
        for (int s = 0; s < n; ++s) {
            float sum = 0.0f;
#pragma nounroll
            for (int e = s; e < n; ++e) {
                const float de = ds[e * 64 + lane];
                sum = e == s ? de : sum + de;
                const float sc = sum / (float)(e - s + 1);
                const unsigned id = (unsigned)(a.id_base + mbase + moment_index(n, s, e));
                for (int r = 0; r < NR; ++r)
                    if (active && sc == rank_dist[r * a.Nq + qi] && id < (unsigned)rank_idx[r * a.Nq + qi]) nlt[r] += 1;
            }
        }
    }

#ifdef VFR_SCORE_STAMPS
    if (lane == 0 && a.stamps)
        for (int i = 0; i < 6; ++i) atomicAdd(a.stamps + i, st_acc[i]);
#endif
    if (active)
        for (int r = 0; r < NR; ++r)
            if (nlt[r]) atomicAdd(reinterpret_cast<unsigned long long *>(a.count_lt + r * a.Nq + qi), (unsigned long long)nlt[r]);
    if (TOPK) {
        if (!a.keep_all) {
            __threadfence_block();
            lane_tighten(col, lane, a.k, &cnt, &thr);        // cut over-full columns close to k (merge sorts exactly)
            __threadfence_block();
        }
        a.buf_cnt[(size_t)task * 64 + lane] = active ? cnt : 0;
    }
}

#include "score_mfma.h"

template <int NT, bool EXACT>
static void launch_fast_nt(const ScoreArgs &a, int kpl, dim3 grid, size_t lds, hipStream_t st)
{
#define VFR_FAST(KPL, NRV, TOPKV)                                                                                    \
    hipLaunchKernelGGL((score_fast_kernel<NT, KPL, NRV, TOPKV, EXACT>), grid, dim3(64), lds, st, a.Q, a.V, a.clip_off, \
                       a.mom_off, a.rank_dist, a.rank_idx, a)
    // NT = 21 with top-k AND two rank keys needs more than the 256 registers of 2 waves/SIMD: that instantiation is
    // built for 1 wave/SIMD (512-register budget, no spills).  vfr_set_option("score_split", 1) runs the two
    // spill-free 2-waves/SIMD specialisations back to back instead (kept for A/B measurements).
    const bool split = NT > 6 && a.k > 0 && a.num_rank > 0 && opt_score_split();
    if (a.k > 0) {
        if (a.num_rank == 0 || split) { if (kpl == 4) VFR_FAST(4, 0, true); else VFR_FAST(8, 0, true); }
        else                          { if (kpl == 4) VFR_FAST(4, 2, true); else VFR_FAST(8, 2, true); }
    }
    if (a.num_rank > 0 && (a.k == 0 || split)) VFR_FAST(4, 2, false);
#undef VFR_FAST
}

// one wave per query: merge the per-chunk candidate lists (+ an optional pre-sorted k-list) into the final top-k.
//   * the chunk lists are consumed as ONE flat sequence: all counts are loaded and prefix-summed first, then CAP keys per
//     round are gathered with all loads in flight at once (each element finds its (chunk, offset) by binary search in the
//     LDS prefix table) -- not one dependent count + key round trip per chunk;
//   * the pool of survivors lives in LDS; an incoming key is appended (ballot + mbcnt) only if it beats the running k-th
//     key, so after the first two rounds almost nothing is appended and the 512-key bitonic sorts (ds_bpermute-bound,
//     ~10 us each) drop from one per 412 keys to a handful per query.
constexpr int MERGE_MAX_CHUNKS = 1024;                   // plan_tasks never makes more
template <int KPL>
__global__ __launch_bounds__(256) void topk_merge_tasks_kernel(const unsigned long long *__restrict__ buf,
                                                               const int *__restrict__ buf_cnt, int num_groups,
                                                               int num_chunks, int64_t Nq, int k, int cap_t,
                                                               const unsigned long long *__restrict__ extra,
                                                               unsigned long long *__restrict__ out_keys,
                                                               unsigned long long *__restrict__ thr_seed,
                                                               float *__restrict__ out_dist,
                                                               int64_t *__restrict__ out_idx, int seed_inclusive,
                                                               const int *__restrict__ group_mask,
                                                               unsigned *__restrict__ hist = nullptr, uint2 *__restrict__ hrange = nullptr)
{
    constexpr int CAP = KPL * 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + wv;
    if (q >= Nq) return;
    const int group = (int)(q >> 6), ql = (int)(q & 63);
    if (group_mask && !group_mask[group]) return;
    __shared__ int pre_s[4][MERGE_MAX_CHUNKS + 1];
    __shared__ unsigned long long pool_s[4][CAP];
    int *pre = pre_s[wv];
    unsigned long long *pool = pool_s[wv];
    unsigned long long key[KPL];
    unsigned long long thr = KEY_MAX;                    // running k-th best key (KEY_MAX until k keys have been sorted)
    int fill = 0;                                        // pool[0, fill) holds the survivors

    auto lds_sync = [&]() { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); };
    auto sort_pool = [&](bool final_pass) {              // pool -> registers -> sorted; keep the k best
        lds_sync();
#pragma unroll
        for (int i = 0; i < KPL; ++i) { const int e = i * 64 + lane; key[i] = e < fill ? pool[e] : KEY_MAX; }
        wave_sort<KPL>(key, lane);
        if (final_pass) return;
        lds_sync();
#pragma unroll
        for (int i = 0; i < KPL; ++i) { const int e = i * 64 + lane; if (e < k) pool[e] = key[i]; }
        if (fill >= k) thr = key_at<KPL>(key, k - 1);
        fill = fill < k ? fill : k;
    };
    auto append = [&](unsigned long long x, bool have) {  // one candidate per lane
        if (fill + 64 > CAP) sort_pool(false);
        const bool pass = have && x < thr;
        const unsigned long long m = __ballot(pass);
        if (pass) pool[fill + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = x;
        fill += __builtin_popcountll(m);
    };

    int total = 0;
    // all chunk counts are requested before the first is used: with few queries in flight (a serving request runs 1024 chunks
    // for its one query group) sixteen dependent load latencies were most of this kernel's time
    constexpr int NSC = MERGE_MAX_CHUNKS / 64;
    int cc[NSC];
#pragma unroll
    for (int j = 0; j < NSC; ++j) {
        const int ch = j * 64 + lane;
        cc[j] = ch < num_chunks ? buf_cnt[((size_t)ch * num_groups + group) * 64 + ql] : 0;
    }
#pragma unroll
    for (int j = 0; j < NSC; ++j) {
        const int base = j * 64;
        if (base >= num_chunks) break;
        const int ch = base + lane;
        const int c = cc[j];
        int incl = c;                                   // inclusive wave scan
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        if (ch < num_chunks) pre[ch + 1] = total + incl;
        total += __shfl(incl, 63, 64);
    }
    if (lane == 0) pre[0] = 0;
    lds_sync();
    for (int done = 0; done < total; done += CAP) {
        unsigned long long x[KPL];
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const int gi = done + i * 64 + lane;
            x[i] = KEY_MAX;
            if (gi < total) {
                int lo = 0, hi = num_chunks;            // largest ch with pre[ch] <= gi  (pre[num_chunks] = total > gi)
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre[mid] <= gi) lo = mid; else hi = mid; }
                const int off = gi - pre[lo];
                // cap_t > 0: the fast kernel's transposed columns [task][slot][lane]; else lane-major [task][lane][CAP]
                x[i] = cap_t > 0 ? buf[(((size_t)lo * num_groups + group) * cap_t + off) * 64 + ql]
                                 : buf[(((size_t)lo * num_groups + group) * 64 + ql) * CAP + off];
            }
        }
#pragma unroll
        for (int i = 0; i < KPL; ++i) append(x[i], done + i * 64 + lane < total);
    }
    if (extra)                            // a pre-sorted k-list per query (the threshold ladder's previous stage), KEY_MAX padded
        for (int off = 0; off < k; off += 64) append(off + lane < k ? extra[q * k + off + lane] : KEY_MAX, off + lane < k);
    sort_pool(true);
    if (hist) {
        // the top-k threshold histogram of the main launch (score_mfma.h): MF_HBINS bins of score BITS, a power-of-two wide, from
        // this list's best key up to its k-th; the list's keys are its first entries
        const unsigned long long kb = key_at<KPL>(key, 0), kk = key_at<KPL>(key, k - 1);
        if (kk != KEY_MAX) {
            const unsigned lob = (unsigned)(kb >> 32), span = (unsigned)(kk >> 32) - lob + 1u;
            unsigned shift = 0u;
            while ((span >> shift) > (unsigned)(MF_HBINS - 2)) ++shift;      // the k-th key falls in a bin below the last
            if (lane == 0) hrange[q] = make_uint2(lob, shift + 1u);
#pragma unroll
            for (int i = 0; i < KPL; ++i) {
                const int e = i * 64 + lane;
                if (e < k && key[i] != KEY_MAX) {
                    const unsigned bin = ((unsigned)(key[i] >> 32) - lob) >> shift;
                    atomicAdd(hist + (size_t)q * MF_HBINS + (bin < (unsigned)(MF_HBINS - 1) ? bin : (unsigned)(MF_HBINS - 1)), 1u);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        int e = i * 64 + lane;
        if (e < k) {
            const bool ok = key[i] != KEY_MAX;
            if (out_keys) out_keys[q * k + e] = key[i];
            // k-th best so far (KEY_MAX if fewer): only ever tightens the stored threshold (it may hold a caller's seed).
            // seed_inclusive: the keys merged here will be scored AGAIN under the seed (the kernels keep key < seed), so
            // the k-th key itself must still pass
            if (thr_seed && e == k - 1) {
                const unsigned long long t = (seed_inclusive && key[i] != KEY_MAX) ? key[i] + 1ull : key[i];
                if (t < thr_seed[q]) thr_seed[q] = t;
            }
            if (out_dist) {
                out_dist[q * k + e] = ok ? __uint_as_float((unsigned)(key[i] >> 32)) : __builtin_inff();
                out_idx[q * k + e] = ok ? (int64_t)(key[i] & 0xffffffffull) : -1;
            }
        }
    }
}

// Stage A of the threshold ladder only wants ONE number per query from its candidates: a key that at least k of them do not
// exceed.  The merge kernel above sorts for it (2 600 candidates per query through 256-key bitonic sorts: 0.20 ms at 5 000
// queries, three times the stage's scoring kernel).  Here a wave (= a query) holds the score bits of all its candidates in
// registers -- lane = task, up to KTH_SLOTS per lane -- and finds the k-th smallest SCORE by bisection on the 32 bits: one
// compare-and-count per candidate and bit, a wave sum, no sort, no LDS.  The seed written is (that score + 1 ulp, id 0): every
// key of that score passes whatever its id -- as valid a bound as the merge's exact k-th key + 1, looser only among exact ties.
constexpr int KTH_SLOTS = 44;
__global__ __launch_bounds__(256) void topk_kth_seed_kernel(const unsigned long long *__restrict__ buf, const int *__restrict__ buf_cnt,
                                                            int num_groups, int num_chunks, int64_t Nq, int k, int cap_t,
                                                            unsigned long long *__restrict__ thr_seed, const int *__restrict__ group_mask)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + wv;
    if (q >= Nq) return;
    const int group = (int)(q >> 6), ql = (int)(q & 63);
    if (group_mask && !group_mask[group]) return;
    const int c = lane < num_chunks ? buf_cnt[((size_t)lane * num_groups + group) * 64 + ql] : 0;
    const int cnt = c < KTH_SLOTS ? c : KTH_SLOTS;                        // (the host only comes here when no column can hold more)
    unsigned sb[KTH_SLOTS];
    const unsigned long long *col = buf + (((size_t)lane * num_groups + group) * cap_t) * 64 + ql;
#pragma unroll
    for (int i = 0; i < KTH_SLOTS; ++i) sb[i] = i < cnt ? (unsigned)(col[(size_t)i * 64] >> 32) : 0xFFFFFFFFu;
    int total = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
    if (total < k) return;                                                // fewer than k candidates: no bound from them
    unsigned lo = 0u, hi = 0xFFFFFFFEu;                                   // smallest t with #(score bits <= t) >= k
    while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int n = 0;
#pragma unroll
        for (int i = 0; i < KTH_SLOTS; ++i) n += sb[i] <= mid ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
        if (n >= k) hi = mid; else lo = mid + 1u;
    }
    if (lane == 0) {
        const unsigned long long t = ((unsigned long long)lo + 1ull) << 32;
        if (t < thr_seed[q]) thr_seed[q] = t;
    }
}

// merge G shard lists [G, Nq, k] -> [Nq, k] (after the all-gather, 8e).  Lists arrive either as (dist, idx) arrays or as
// the packed int64 keys the ranks exchange (KEY_EMPTY = (+inf, 0xffffffff) marks an unused slot).  Same LDS pool as the
// task merge: all G*k keys are read as one flat sequence, CAP per round with the loads in flight together, and a key is
// appended only if it beats the running k-th key -- after the first round almost nothing is, so a query costs ~2 sorts.
template <int KPL, bool PACKED>
__global__ __launch_bounds__(256) void topk_merge_parts_kernel(const float *__restrict__ pd,
                                                               const int64_t *__restrict__ pi, int64_t slot_stride,
                                                               int64_t q_stride, int G, int64_t Nq, int k,
                                                               float *__restrict__ out_dist,
                                                               int64_t *__restrict__ out_idx,
                                                               int64_t *__restrict__ out_keys)
{
    constexpr int CAP = KPL * 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + wv;
    if (q >= Nq) return;
    __shared__ unsigned long long pool_s[4][CAP];
    unsigned long long *pool = pool_s[wv];
    unsigned long long key[KPL];
    unsigned long long thr = KEY_MAX;
    int fill = 0;
    auto lds_sync = [&]() { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); };
    auto sort_pool = [&](bool final_pass) {
        lds_sync();
#pragma unroll
        for (int i = 0; i < KPL; ++i) { const int e = i * 64 + lane; key[i] = e < fill ? pool[e] : KEY_MAX; }
        wave_sort<KPL>(key, lane);
        if (final_pass) return;
        lds_sync();
#pragma unroll
        for (int i = 0; i < KPL; ++i) { const int e = i * 64 + lane; if (e < k) pool[e] = key[i]; }
        if (fill >= k) thr = key_at<KPL>(key, k - 1);
        fill = fill < k ? fill : k;
    };
    const int total = G * k;
    for (int done = 0; done < total; done += CAP) {
        unsigned long long x[KPL];
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const int gi = done + i * 64 + lane;
            x[i] = KEY_MAX;
            if (gi < total) {
                const int g = gi / k, off = gi - g * k;
                const int64_t at = (int64_t)g * slot_stride + q * q_stride + off;  // [G, Nq, k]: slot_stride = Nq * k, q_stride = k
                if constexpr (PACKED) {
                    const unsigned long long v = (unsigned long long)pi[at];
                    x[i] = v >= KEY_EMPTY ? KEY_MAX : v;
                } else {
                    const int64_t id = pi[at];
                    x[i] = id < 0 ? KEY_MAX : make_key(pd[at], (unsigned)id);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            if (fill + 64 > CAP) sort_pool(false);
            const bool pass = x[i] < thr;                        // KEY_MAX (empty) never passes
            const unsigned long long m = __ballot(pass);
            if (pass) pool[fill + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = x[i];
            fill += __builtin_popcountll(m);
        }
    }
    sort_pool(true);
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        int e = i * 64 + lane;
        if (e < k) {
            const bool ok = key[i] != KEY_MAX;
            if (out_dist) {
                out_dist[q * k + e] = ok ? __uint_as_float((unsigned)(key[i] >> 32)) : __builtin_inff();
                out_idx[q * k + e] = ok ? (int64_t)(key[i] & 0xffffffffull) : -1;
            }
            if (out_keys) out_keys[q * k + e] = (int64_t)(ok ? key[i] : KEY_EMPTY);
        }
    }
}

// (dist, idx) lists -> the packed int64 keys the ranks exchange (one collective instead of two); idx < 0 -> KEY_EMPTY
__global__ __launch_bounds__(256) void topk_pack_keys_kernel(const float *__restrict__ d, const int64_t *__restrict__ ix,
                                                             int64_t n, int64_t *__restrict__ keys)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t id = ix[i];
    keys[i] = (int64_t)(id < 0 ? KEY_EMPTY : make_key(d[i], (unsigned)id));
}

// best ground-truth-positive key per (threshold, query):  min over the labelled moments of the query's own video of
// (score, global id)  (evaluate.py:67-77: the first positive in the sorted order).  One wave per selected query.
__global__ __launch_bounds__(256) void gt_fill_keys_kernel(int64_t n, int64_t *__restrict__ keys)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = (int64_t)KEY_EMPTY;
}
// the same initialisation with the unpacked forms of vfr_gt_rank_keys_f32: distance +inf, id 0xffffffff, counts 0, flag 0
__global__ __launch_bounds__(256) void gt_fill_rank_kernel(int64_t n, int64_t *__restrict__ keys, float *__restrict__ rank_dist,
                                                           int64_t *__restrict__ rank_idx, int64_t *__restrict__ count0,
                                                           int *__restrict__ missing)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && missing) *missing = 0;
    if (i >= n) return;
    keys[i] = (int64_t)KEY_EMPTY;
    if (rank_dist) { rank_dist[i] = __builtin_inff(); rank_idx[i] = 0xFFFFFFFFll; }
    if (count0) count0[i] = 0;
}
__global__ __launch_bounds__(256) void gt_best_keys_kernel(const float *__restrict__ sc, int64_t n_sel, int M,
                                                           int score_stride, const uint8_t *__restrict__ labels, int R,
                                                           int label_stride, const int64_t *__restrict__ id_base,
                                                           const int64_t *__restrict__ sel, int64_t Nq,
                                                           int64_t *__restrict__ keys, float *__restrict__ rank_dist = nullptr,
                                                           int64_t *__restrict__ rank_idx = nullptr, int *__restrict__ missing = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_sel) return;
    const int64_t base = id_base[s], q = sel[s];
    for (int r = 0; r < R; ++r) {
        unsigned long long best = KEY_MAX;
        const uint8_t *lab = labels + ((int64_t)r * n_sel + s) * label_stride;
        for (int m = lane; m < M; m += 64)
            if (lab[m]) {
                const unsigned long long kk = make_key(sc[s * score_stride + m], (unsigned)(base + m));
                best = kk < best ? kk : best;
            }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(best, d, 64);
            best = o < best ? o : best;
        }
        if (lane == 0) {
            const unsigned long long kq = best >= KEY_EMPTY ? KEY_EMPTY : best;
            keys[(int64_t)r * Nq + q] = (int64_t)kq;
            if (rank_dist) {
                rank_dist[(int64_t)r * Nq + q] = __uint_as_float((unsigned)(kq >> 32));
                rank_idx[(int64_t)r * Nq + q] = (int64_t)(kq & 0xffffffffull);
            }
            if (missing && kq == KEY_EMPTY) atomicOr(missing, 1);
        }
    }
}

// a11 (model/evaluate.py:59-62, utils.get_iou model/utils.py:78-82): labels[r][q][m] = 1 iff at least two annotators have
// IoU > (or >=) thr[r] with local moment m of the query's own video.  One thread per (query, moment); the IoU is the
// reference's float64 intersection / union (IEEE division: the same bits as numpy's).
constexpr int MAX_LABEL_THR = 16;
struct LabelThr { double t[MAX_LABEL_THR]; };
__global__ __launch_bounds__(256) void gt_labels_kernel(const int32_t *__restrict__ times, const int32_t *__restrict__ nannot,
                                                        const int32_t *__restrict__ n_own, int64_t Nq, int A, int R,
                                                        LabelThr thr, int strict, int Mmax, uint8_t *__restrict__ labels)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Nq * Mmax) return;
    const int64_t q = i / Mmax;
    const int m = (int)(i - q * Mmax), n = n_own[q];
    int s = m, e = m;                                     // generate_moments order: (j, j) first, then combinations
    bool real = m < n * (n + 1) / 2;
    if (real && m >= n) {
        int rest = m - n;
        s = 0;
        while (rest >= n - 1 - s) { rest -= n - 1 - s; ++s; }
        e = s + 1 + rest;
    }
    int cnt[MAX_LABEL_THR];
#pragma unroll
    for (int r = 0; r < MAX_LABEL_THR; ++r) cnt[r] = 0;
    const int na = nannot[q];
    for (int a = 0; a < na && real; ++a) {
        const int ts = times[(q * A + a) * 2], te = times[(q * A + a) * 2 + 1];
        const int inter = max(min(te, e) + 1 - max(ts, s), 0), uni = max(te, e) + 1 - min(ts, s);
        const double iou = (double)inter / (double)uni;
#pragma unroll
        for (int r = 0; r < MAX_LABEL_THR; ++r)
            if (r < R) cnt[r] += (strict ? iou > thr.t[r] : iou >= thr.t[r]) ? 1 : 0;
    }
#pragma unroll
    for (int r = 0; r < MAX_LABEL_THR; ++r)
        if (r < R) labels[((int64_t)r * Nq + q) * Mmax + m] = cnt[r] >= 2 ? 1 : 0;
}

// each query against its own video (evaluate_single.py:48-53).  One WAVE per query: lane c runs clip c's k-ascending
// distance chain (the clips of a video are independent chains), the distances go through LDS, then lane s walks the spans
// (s, s), (s, s+1), ... keeping the left-to-right running sum of the oracle.  (Thread-per-query left the chip to 79 waves
// of 21 serial chains each: 0.27 ms for 5000 queries; this is ~15 us.)
__global__ __launch_bounds__(256) void score_own_kernel(const float *__restrict__ Q, int64_t Nq,
                                                        const float *__restrict__ V,
                                                        const int32_t *__restrict__ clip_off,
                                                        const int32_t *__restrict__ own, int D, float eps, int Mmax,
                                                        float *__restrict__ scores)
{
    __shared__ float ds_all[4][NMAX_DENSE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + wv;
    if (q >= Nq) return;
    float *ds = ds_all[wv];
    const int v = own[q], c0 = clip_off[v], n = clip_off[v + 1] - c0;
    float *out = scores + q * Mmax;
    for (int m = n * (n + 1) / 2 + lane; m < Mmax; m += 64) out[m] = __builtin_inff();     // slots past this video's moments
    const bool vec = (D & 3) == 0 && ((((uintptr_t)V) | ((uintptr_t)Q)) & 15) == 0;
    const float *qr = Q + q * D;
    for (int c = lane; c < n; c += 64) {
        float acc = 0.0f;
        const float *vr = V + (int64_t)(c0 + c) * D;
        if (vec) {                                       // same k-ascending chain, 16-byte loads
#pragma unroll 5
            for (int k = 0; k < D; k += 4) {
                const float4 a = *reinterpret_cast<const float4 *>(vr + k), b = *reinterpret_cast<const float4 *>(qr + k);
                float d = (a.x - b.x) + eps; acc = __builtin_fmaf(d, d, acc);
                d = (a.y - b.y) + eps; acc = __builtin_fmaf(d, d, acc);
                d = (a.z - b.z) + eps; acc = __builtin_fmaf(d, d, acc);
                d = (a.w - b.w) + eps; acc = __builtin_fmaf(d, d, acc);
            }
        } else {
            for (int k = 0; k < D; ++k) {
                float d = (vr[k] - qr[k]) + eps;
                acc = __builtin_fmaf(d, d, acc);
            }
        }
        ds[c] = __builtin_sqrtf(acc);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < n; s += 64) {
        float sum = 0.0f;
        for (int e = s; e < n; ++e) {
            const float de = ds[e];
            sum = e == s ? de : sum + de;
            out[moment_index(n, s, e)] = sum / (float)(e - s + 1);
        }
    }
}

static void plan_tasks(int64_t Nq, int Nv, int *groups, int *chunks)
{
    int g = (int)cdiv(Nq, 64);
    // Wave-tasks = query groups x video chunks.  More tasks even out the tail round of the 2048 wave slots, fewer tasks
    // amortise a task's fixed cost (100 query registers, candidate flush) over more videos: the measured optimum
    // (tools/seeded_scale.py, 5000 queries) follows ~174 * sqrt(Nv): 3k tasks at 312 videos, 6k at 1218, 16k at 9744.
    int64_t total = opt_score_tasks() > 0 ? opt_score_tasks() : (int64_t)(174.0 * __builtin_sqrt((double)(Nv > 1 ? Nv : 1)));
    if (opt_score_tasks() <= 0) total = total < 2048 ? 2048 : (total > 16384 ? 16384 : total);
    int64_t want = cdiv(total, g);
    int c = (int)(want < 1 ? 1 : want);
    if (c > Nv) c = Nv < 1 ? 1 : Nv;
    if (c > 1024) c = 1024;
    *groups = g;
    *chunks = c;
}
static int kpl_for(int k) { return k <= 128 ? 4 : 8; }

// Threshold ladder of the top-k pass (k > 0, fast path).  The fused kernel appends every moment that beats the query's
// current threshold, so the threshold's quality decides how often its exact path runs:
//   stage A  the short moments (<= PRE_LEVELS clips) of the first PRE_VIDEOS videos, one video per wave-task, threshold
//            +inf -> merge -> their k-th key: a valid (any k keys bound the k-th best) and already tight bound, for 0.4 ms;
//   stage B  the first pre_b_videos() videos (the sample included) under A's threshold -> merge: the k-th key of ~Nv/16 (<= 640; tools/pre_b_sweep.py: the optimum at 2500, 5000 and 10000 videos)
//            videos, which admits only ~8k candidates per query in ...
//   stage C  ... the rest of the corpus (the main launch); the final merge takes B's merged list as its `extra` input.
// Every video's full moment set is scored exactly once (stage A's partial pass over 64 videos is the only repeated work).
#ifndef VFR_PRE_VIDEOS
#define VFR_PRE_VIDEOS 64
#endif
#ifndef VFR_PRE_LEVELS
#define VFR_PRE_LEVELS 2
#endif
constexpr int PRE_VIDEOS = VFR_PRE_VIDEOS;
constexpr int PRE_CHUNKS = 64;
constexpr int PRE_LEVELS = VFR_PRE_LEVELS;       // stage A keeps moments of at most this many clips
static int pre_b_videos(int Nv) { if (opt_score_pre_b() > 0) return opt_score_pre_b() < Nv ? opt_score_pre_b() : Nv; const int b = Nv / 16; return b > 640 ? 640 : b; }

struct TopkWs { unsigned long long *buf, *buf_pre, *pre_keys, *pre_keys2; int *cnt, *cnt_pre; unsigned long long *thr; size_t total; };
static TopkWs carve_topk(void *base, int64_t Nq, int Nv, int k)
{
    TopkWs w{};
    int g, c;
    plan_tasks(Nq, Nv, &g, &c);
    size_t tasks = (size_t)g * c, tasks_pre = (size_t)g * PRE_CHUNKS, cap = 512, off = 0;   // cap: largest either kernel uses
    auto take = [&](size_t bytes) { char *p = static_cast<char *>(base) + off; off += align_up(bytes, 256); return p; };
    w.thr = reinterpret_cast<unsigned long long *>(take((size_t)Nq * 8));
    w.cnt = reinterpret_cast<int *>(take(tasks * 64 * 4));
    w.cnt_pre = reinterpret_cast<int *>(take(tasks_pre * 64 * 4));
    w.pre_keys = reinterpret_cast<unsigned long long *>(take(k > 0 ? (size_t)Nq * k * 8 : 0));
    w.pre_keys2 = reinterpret_cast<unsigned long long *>(take(k > 0 ? (size_t)Nq * k * 8 : 0));
    w.buf_pre = reinterpret_cast<unsigned long long *>(take(k > 0 ? tasks_pre * 64 * cap * 8 : 0));
    w.buf = reinterpret_cast<unsigned long long *>(take(k > 0 ? tasks * 64 * cap * 8 : 0));
    w.total = off;
    return w;
}

static bool fast_applicable(const ScoreArgs &a)
{
    const int NT = a.ds_rows <= 6 ? 6 : 21, M = NT * (NT + 1) / 2;
    return a.D == FAST_D && opt_score_fast() && (a.num_rank == 0 || a.num_rank == 2) && (a.k > 0 || a.num_rank > 0) &&
           (a.ds_rows <= 6 || a.ds_rows == 21) && a.k + M <= 512;
}


// MFMA pre-filter launch (score_mfma.h): NT = 6 for banks of <= 6 clips per video, else 21 (any mix of lengths <= 21)
static int launch_mfma(const ScoreArgs &a, hipStream_t st, int *cap_transposed)
{
    const MfmaArgs &m = *a.mf_host;
    const int tasks = a.num_groups * a.num_chunks;
    const int NT = a.ds_rows <= 6 ? 6 : 21, M = NT * (NT + 1) / 2;
    const int kpl = (a.k > 0 && a.k + M > 256) ? 8 : 4;
    if (cap_transposed) *cap_transposed = a.k > 0 ? kpl * 64 : 0;
    const int RR = MF_RING_ROWS(NT);
    const size_t lds = ((size_t)RR * 64 + (size_t)a.num_rank * NT * 64) * sizeof(float);
    ProfScope prof(a.prof_site ? a.prof_site : a.k == 0 ? SITE_SCORE_RANK : SITE_SCORE_FUSED, st);
    dim3 grid((unsigned)tasks);
#define VFR_MF(NTV, KPL, NRV, TOPKV, BF)                                                                                  \
    hipLaunchKernelGGL((score_mfma_kernel<NTV, KPL, NRV, TOPKV, BF>), grid, dim3(64), lds, st, a.Q, a.V, a.clip_off, a.mom_off, a, m)
#define VFR_MF_NT(NTV, BF)                                                                                                \
    do {                                                                                                                  \
        if (a.k > 0) {                                                                                                    \
            if (a.num_rank == 0) { if (kpl == 4 && NTV == 6) VFR_MF(NTV, (NTV == 6 ? 4 : 8), 0, true, BF); else VFR_MF(NTV, 8, 0, true, BF); } \
            else                 { if (kpl == 4 && NTV == 6) VFR_MF(NTV, (NTV == 6 ? 4 : 8), 2, true, BF); else VFR_MF(NTV, 8, 2, true, BF); } \
        } else VFR_MF(NTV, 4, 2, false, BF);                                                                              \
    } while (0)
    if (NT == 6) { if (a.mf_bf16) VFR_MF_NT(6, true); else VFR_MF_NT(6, false); }
    else         { if (a.mf_bf16) VFR_MF_NT(21, true); else VFR_MF_NT(21, false); }
#undef VFR_MF_NT
#undef VFR_MF
    VFR_CHECK_LAUNCH("score_mfma_kernel");
    return VFR_OK;
}

static bool mfma_applicable(int D, int max_clips, int num_rank, int k, int Nv, int total_clips)
{
    const int NT = max_clips <= 6 ? 6 : 21, M = NT * (NT + 1) / 2;
    return D == FAST_D && max_clips <= 21 && (num_rank == 0 || num_rank == 2) && (k > 0 || num_rank > 0) && Nv > 0 &&
           total_clips > 0 && (k == 0 || k + MF_EXTRA + M <= 512);
}
// banks this small (a multi-GPU threshold sample, a handful of videos) cost less on the exact kernels than the pre-filter's
// fixed launches (pre-passes, finisher, masked fallback ladder)
static bool mfma_worthwhile(int Nv, int dtype)
{
    return dtype == VFR_MFMA_BF16 || Nv >= opt_mfma_min();
}

template <int MODE>
static int launch_score(const ScoreArgs &a, int kpl, hipStream_t st, int *cap_transposed = nullptr)
{
    if (cap_transposed) *cap_transposed = 0;
    if (MODE == 1 && a.mf_host) return launch_mfma(a, st, cap_transposed);
    const int tasks = a.num_groups * a.num_chunks;
    ProfScope prof(a.prof_site ? a.prof_site : MODE == 0 ? SITE_SCORE_DENSE : (a.force_generic && a.k > 0) ? SITE_SCORE_PREPASS
                   : a.k == 0 ? SITE_SCORE_RANK : SITE_SCORE_FUSED, st);
    const int NTsel = a.ds_rows <= 6 ? 6 : 21, Msel = NTsel * (NTsel + 1) / 2;
    if (MODE == 1 && fast_applicable(a) && !a.force_generic) {
        const int NT = NTsel;
        kpl = a.k + Msel <= 256 ? 4 : 8;                   // candidate columns hold k kept + one video's worth
        if (cap_transposed && a.k > 0) *cap_transposed = kpl * 64;
        const bool lox_lds = NT > 6 && a.num_rank > 0;
        const bool exact = a.min_clips == NT;              // every video has exactly NT clips: n is a constant
        const int NCg = 3, rows = (NT + NCg - 1) / NCg * NCg;
        const size_t lds = ((size_t)2 * NCg * FAST_D + (size_t)rows * 64 + (lox_lds ? (size_t)a.num_rank * NT * 64 : 0)) * sizeof(float);
        dim3 grid((unsigned)tasks);
#ifdef VFR_SCORE_STAMPS
        static unsigned long long *stamps_dev = nullptr;
        if (!stamps_dev) (void)hipMalloc(&stamps_dev, 6 * 8);
        (void)hipMemsetAsync(stamps_dev, 0, 6 * 8, st);
        const_cast<ScoreArgs &>(a).stamps = stamps_dev;
#endif
        if (NT == 6) { if (exact) launch_fast_nt<6, true>(a, kpl, grid, lds, st); else launch_fast_nt<6, false>(a, kpl, grid, lds, st); }
        else         { if (exact) launch_fast_nt<21, true>(a, kpl, grid, lds, st); else launch_fast_nt<21, false>(a, kpl, grid, lds, st); }
        VFR_CHECK_LAUNCH("score_fast_kernel");
#ifdef VFR_SCORE_STAMPS
        {
            unsigned long long h[6];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(h, stamps_dev, sizeof h, hipMemcpyDeviceToHost);
            double tot = 0; for (int i = 0; i < 6; ++i) tot += (double)h[i];
            fprintf(stderr, "[stamps] k=%d nr=%d videos=%d tasks=%d  head %.1f%%  tighten/thr %.1f%%  distances %.1f%%  sqrt->regs+triangle %.1f%%  slow-levels %.1f%%  ties/loop %.1f%%  (%.3g ticks/wave)\n",
                    a.k, a.num_rank, a.v_hi - a.v_lo, tasks, 100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot,
                    100 * h[4] / tot, 100 * h[5] / tot, tot / tasks);
        }
#endif
        return VFR_OK;
    }
    const size_t lds = (size_t)4 * a.ds_rows * 64 * sizeof(float);
    dim3 grid((unsigned)cdiv(tasks, 4)), block(256);
#define VFR_LAUNCH(DT, KPL)                                                                              \
    hipLaunchKernelGGL((score_kernel<DT, MODE, KPL>), grid, block, lds, st, a.Q, a.V, a.clip_off, a.mom_off, \
                       a.rank_dist, a.rank_idx, a)
    if (a.D == 100) { if (kpl == 4) VFR_LAUNCH(100, 4); else VFR_LAUNCH(100, 8); }
    else            { if (kpl == 4) VFR_LAUNCH(0, 4);   else VFR_LAUNCH(0, 8); }
#undef VFR_LAUNCH
    VFR_CHECK_LAUNCH("score_kernel");
    return VFR_OK;
}


// The whole fused pass over one bank with whichever scoring kernel `a` selects (exact: score_fast_kernel / score_kernel;
// a.mf_host: the MFMA pre-filter): threshold ladder, merges, final list.  w.thr holds the initial per-query thresholds
// (+inf keys, or the caller's seed: `seeded`).  out_keys != null: the final list is also written as sorted keys [Nq, k]
// (KEY_MAX padded).  a.group_mask limits every launch to the flagged query groups.
static int run_pass(ScoreArgs a, const TopkWs &w, bool seeded, float *out_dist, int64_t *out_idx,
                    unsigned long long *out_keys, hipStream_t st)
{
    const int64_t Nq = a.Nq;
    const int Nv = a.Nv, k = a.k;
    a.buf = w.buf; a.buf_cnt = w.cnt; a.thr_global = w.thr;
    a.v_lo = 0; a.v_hi = Nv;
    plan_tasks(Nq, Nv, &a.num_groups, &a.num_chunks);
    const int kpl = kpl_for(k);
    int cap_t = 0, cap_pre = 0;
    const unsigned long long *extra = nullptr;
    auto merge = [&](const unsigned long long *buf, const int *cnt, int chunks, int capt, const unsigned long long *ex,
                     unsigned long long *okeys, unsigned long long *seed, float *od, int64_t *oi, int seed_incl = 0,
                     unsigned *hist = nullptr, uint2 *hrange = nullptr) {
        dim3 grid((unsigned)cdiv(Nq, 4)), block(256);
        if (kpl == 4)
            hipLaunchKernelGGL((topk_merge_tasks_kernel<4>), grid, block, 0, st, buf, cnt, a.num_groups, chunks, Nq, k,
                               capt, ex, okeys, seed, od, oi, seed_incl, a.group_mask, hist, hrange);
        else
            hipLaunchKernelGGL((topk_merge_tasks_kernel<8>), grid, block, 0, st, buf, cnt, a.num_groups, chunks, Nq, k,
                               capt, ex, okeys, seed, od, oi, seed_incl, a.group_mask, hist, hrange);
    };
    if (Nv > 0) {
        // threshold ladder (see PRE_VIDEOS).  Stage A needs no seed and runs for every bank size (a 32-video sample shard is
        // just a stage A plus its main pass); a caller-provided seed (the multi-GPU sample) replaces it.  Stage B only pays off on a long rest.
        const bool fast = k > 0 && (a.mf_host != nullptr || fast_applicable(a));
        const int na = (fast && !seeded) ? (Nv < PRE_VIDEOS ? Nv : PRE_VIDEOS) : 0;
        // (after stage A's 64 videos a sixteenth of any corpus >= 256 tightens a lot; after a seed -- normally the k-th key of
        // a 256-video global sample -- only a stage B of >= 512 videos can tighten further)
        int nb = (fast && Nv >= (seeded ? 4096 : 256)) ? pre_b_videos(Nv) : 0;
        // (with the candidate histogram tightening the main launch's threshold as it goes, stage B need not be as long: half --
        // Nv / 32, at most 320 videos -- measured 1-2 % better at 10 000 videos, 21 and 6 clips: profiles/r4m_scorer_ab_pre_b.txt)
        // (not behind a caller's seed: there stage B is what turns the seed -- the k-th key of a 256-video sample -- into a list, and
        // half of it starts the main launch looser than no histogram at all: tools/rank_sim.py 2, fused 2.49 -> 2.58 ms)
        if (nb > 0 && !seeded && a.mf_host && a.mf_host->hist && opt_score_pre_b() <= 0) nb = nb / 2 > 0 ? nb / 2 : nb;
        if (k > 0 && !fast && !seeded && Nv <= 512) {
            // generic shapes, small banks: ~10 videos per task for the cooperative-compaction kernel
            const int c = Nv / 10 < 1 ? 1 : Nv / 10;
            a.num_chunks = c < a.num_chunks ? c : a.num_chunks;
        }
        if (na > 0) {
            // Stage A only has to produce a VALID threshold cheaply: one video per wave-task, threshold +inf, and only the
            // moments of <= PRE_LEVELS clips are kept (the best moments are short ones, and any k keys bound the k-th
            // best from above); the merge's k-th key (+1: inclusive) becomes every query's threshold.  The sample videos
            // are then scored like all others by the next stage, so nothing here needs rank keys or an output list.
            ScoreArgs pre = a;
            pre.v_lo = 0; pre.v_hi = na; pre.num_chunks = na; pre.num_rank = 0; pre.rank_dist = nullptr; pre.rank_idx = nullptr;
            pre.buf = w.buf_pre; pre.buf_cnt = w.cnt_pre; pre.keep_all = 1; pre.level_cap = PRE_LEVELS;
            pre.prof_site = a.prof_site ? a.prof_site : SITE_SCORE_PREPASS;
            if (int rc = launch_score<1>(pre, kpl, st, &cap_pre)) return rc;
            {
                ProfScope prof(SITE_TOPK_MERGE, st);
                // (a column of stage A holds at most PRE_LEVELS x clips-per-video keys: the bisection kernel when they fit its registers)
                if (opt_score_kth_seed() && cap_pre > 0 && na <= 64 && PRE_LEVELS * a.ds_rows <= KTH_SLOTS)
                    hipLaunchKernelGGL(topk_kth_seed_kernel, dim3((unsigned)cdiv(Nq, 4)), dim3(256), 0, st, w.buf_pre, w.cnt_pre, a.num_groups, na, Nq, k,
                                       cap_pre, w.thr, a.group_mask);
                else
                    merge(w.buf_pre, w.cnt_pre, na, cap_pre, nullptr, nullptr, w.thr, nullptr, nullptr, 1);
            }
            VFR_CHECK_LAUNCH("topk_merge_tasks_kernel(A)");
        }
        if (nb > 0) {
            ScoreArgs b = a;                       // same buffers as the main launch: the stream orders B, its merge, C
            b.v_hi = a.v_lo + nb;
            b.prof_site = a.prof_site ? a.prof_site : SITE_SCORE_PREPASS;
            if (int rc = launch_score<1>(b, kpl, st, &cap_t)) return rc;
            {
                ProfScope prof(SITE_TOPK_MERGE, st);
                // (MFMA pre-filter: the merged list also seeds the main launch's threshold histogram)
                merge(w.buf, w.cnt, b.num_chunks, cap_t, extra, w.pre_keys2, w.thr, nullptr, nullptr, 0,
                      a.mf_host ? a.mf_host->hist : nullptr, a.mf_host ? a.mf_host->hrange : nullptr);
            }
            VFR_CHECK_LAUNCH("topk_merge_tasks_kernel(B)");
            extra = w.pre_keys2;
            a.v_lo = b.v_hi;
        }
        if (int rc = launch_score<1>(a, kpl, st, &cap_t)) return rc;
    } else if (k > 0) {
        if (hipMemsetAsync(w.cnt, 0, (size_t)a.num_groups * a.num_chunks * 64 * 4, st) != hipSuccess)
            return fail(VFR_EHIP, "score pass: hipMemsetAsync failed");
    }
    if (k > 0) {
        ProfScope prof(SITE_TOPK_MERGE, st);
        merge(w.buf, w.cnt, a.num_chunks, cap_t, extra, out_keys, nullptr, out_dist, out_idx);
        VFR_CHECK_LAUNCH("topk_merge_tasks_kernel");
    }
    return VFR_OK;
}

}  // namespace vfr

extern "C" {

int vfr_score_moments_f32(const float *Q, int64_t Nq, const float *V, const int32_t *clip_offsets,
                          const int64_t *moment_offsets, int Nv, int max_clips, int64_t total_moments, int D,
                          float eps, float *scores, vfr_stream_t stream)
{
    VFR_REQUIRE(Q && V && clip_offsets && moment_offsets && scores && Nq >= 0 && Nv >= 0 && D > 0 && total_moments >= 0,
                VFR_EINVAL, "vfr_score_moments_f32: bad argument");
    VFR_REQUIRE(max_clips <= vfr::NMAX_DENSE, VFR_EUNSUPPORTED, "vfr_score_moments_f32: max_clips=%d > %d", max_clips,
                vfr::NMAX_DENSE);
    if (Nq == 0 || Nv == 0) return VFR_OK;
    vfr::ScoreArgs a{};
    a.Q = Q; a.Nq = Nq; a.V = V; a.clip_off = clip_offsets; a.mom_off = moment_offsets; a.Nv = Nv; a.D = D; a.eps = eps;
    a.scores = scores; a.total_moments = total_moments; a.ds_rows = max_clips < 1 ? 1 : max_clips;
    a.v_lo = 0; a.v_hi = Nv;
    vfr::plan_tasks(Nq, Nv, &a.num_groups, &a.num_chunks);
    return vfr::launch_score<0>(a, 4, vfr::as_stream(stream));
}

int vfr_score_own_f32(const float *Q, int64_t Nq, const float *V, const int32_t *clip_offsets, const int32_t *own,
                      int max_clips, int D, float eps, int Mmax, float *scores, vfr_stream_t stream)
{
    VFR_REQUIRE(Q && V && clip_offsets && own && scores && Nq >= 0 && D > 0 && Mmax >= 0, VFR_EINVAL,
                "vfr_score_own_f32: bad argument");
    VFR_REQUIRE(max_clips <= vfr::NMAX_DENSE && Mmax >= max_clips * (max_clips + 1) / 2, VFR_EUNSUPPORTED,
                "vfr_score_own_f32: max_clips=%d (limit %d) needs Mmax >= %d, got %d", max_clips, vfr::NMAX_DENSE,
                max_clips * (max_clips + 1) / 2, Mmax);
    if (Nq == 0) return VFR_OK;
    hipLaunchKernelGGL(vfr::score_own_kernel, dim3((unsigned)vfr::cdiv(Nq, 4)), dim3(256), 0, vfr::as_stream(stream), Q, Nq,
                       V, clip_offsets, own, D, eps, Mmax, scores);
    VFR_CHECK_LAUNCH("score_own_kernel");
    return VFR_OK;
}

size_t vfr_score_topk_workspace_bytes(int64_t Nq, int Nv, int k)
{
    if (Nq < 0 || Nv < 0 || k < 0) return 0;
    return vfr::carve_topk(nullptr, Nq, Nv, k).total;
}

int vfr_score_topk_f32(const float *Q, int64_t Nq, const float *V, const int32_t *clip_offsets,
                       const int64_t *moment_offsets, int Nv, int total_clips, int min_clips, int max_clips, int D,
                       float eps, int64_t id_base, int k, float *out_dist, int64_t *out_idx, int num_rank,
                       const float *rank_dist, const int64_t *rank_idx, int64_t *count_lt, const int64_t *thr_seed,
                       void *workspace, size_t workspace_bytes, vfr_stream_t stream)
{
    VFR_REQUIRE(num_rank >= 0 && num_rank <= vfr::MAX_RANK, VFR_EUNSUPPORTED, "vfr_score_topk_f32: num_rank=%d > %d",
                num_rank, vfr::MAX_RANK);
    if (num_rank == 0) { rank_dist = nullptr; rank_idx = nullptr; }
    VFR_REQUIRE(Q && V && clip_offsets && moment_offsets && Nq >= 0 && Nv >= 0 && D > 0 && k >= 0 && id_base >= 0 &&
                    total_clips >= 0,
                VFR_EINVAL, "vfr_score_topk_f32: bad argument");
    VFR_REQUIRE(k == 0 || (out_dist && out_idx), VFR_EINVAL, "vfr_score_topk_f32: k > 0 needs out_dist/out_idx");
    VFR_REQUIRE(num_rank == 0 || (rank_dist && rank_idx && count_lt), VFR_EINVAL,
                "vfr_score_topk_f32: num_rank > 0 needs rank_dist, rank_idx and count_lt");
    VFR_REQUIRE(max_clips <= vfr::NMAX_FUSED, VFR_EUNSUPPORTED, "vfr_score_topk_f32: max_clips=%d > %d", max_clips,
                vfr::NMAX_FUSED);
    VFR_REQUIRE(k <= 448, VFR_EUNSUPPORTED, "vfr_score_topk_f32: k=%d > 448", k);
    if (Nq == 0) return VFR_OK;
    hipStream_t st = vfr::as_stream(stream);
    VFR_REQUIRE(workspace && workspace_bytes >= vfr_score_topk_workspace_bytes(Nq, Nv, k), VFR_EWORKSPACE,
                "vfr_score_topk_f32: workspace %zu < %zu bytes", workspace_bytes,
                vfr_score_topk_workspace_bytes(Nq, Nv, k));
    vfr::TopkWs w = vfr::carve_topk(workspace, Nq, Nv, k);
    vfr::ScoreArgs a{};
    a.Q = Q; a.Nq = Nq; a.V = V; a.clip_off = clip_offsets; a.mom_off = moment_offsets; a.Nv = Nv; a.D = D; a.eps = eps;
    a.id_base = id_base; a.k = k; a.num_rank = num_rank; a.rank_dist = rank_dist; a.rank_idx = rank_idx; a.count_lt = count_lt;
    a.ds_rows = max_clips < 1 ? 1 : max_clips;
    a.total_clips = total_clips;
    a.min_clips = min_clips;
    if (Nv > 0 && k > 0) {
        if (int rc = thr_seed ? vfr::copy_region(w.thr, thr_seed, (size_t)Nq * 8, st) : vfr::fill_region(w.thr, 0xFFFFFFFFu, (size_t)Nq * 8, st)) return rc;
    }
    return vfr::run_pass(a, w, thr_seed != nullptr, out_dist, out_idx, nullptr, st);
}


}  // extern "C"

// ---- MFMA pre-filter path (score_mfma.h) ------------------------------------------------------------------------------
namespace vfr {
struct MfmaWs {
    float *rv; BankSig *sig; unsigned long long *hash_now; int *stale; int *fallback; unsigned long long *pairs_total; unsigned long long *cnt_ws; size_t zero_bytes; char *zero_base;
    float *va; float4 *qmeta; unsigned *tab; unsigned *wmax, *qbound; unsigned short *vb; ulonglong2 *amb; int tasks, groups;
    float *mu, *vc, *qc; double *mean_partial; int mean_blocks, mean_rows;
    void *topk; size_t topk_bytes; size_t total;
    int *perm, *diff; unsigned *hist; uint2 *hrange; float *Qs, *rds; int64_t *ris, *seeds, *cnts; float *ods; int64_t *ois;
};
static MfmaWs carve_mfma(void *base, int64_t Nq, int Nv, int total_clips, int k)
{
    MfmaWs w{};
    int g, c;
    plan_tasks(Nq, Nv, &g, &c);
    w.groups = g; w.tasks = g * c;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = static_cast<char *>(base) + off; off += align_up(bytes, 256); return p; };
    // bank side first, at offsets that depend on the bank alone (total_clips): a caller that scores many query batches against
    // one resident bank keeps these across calls (VFR_MFMA_BANK_READY)
    w.rv = reinterpret_cast<float *>(take(4));
    w.sig = reinterpret_cast<BankSig *>(take(sizeof(BankSig)));          // what the products below were computed from (guard of BANK_READY)
    w.mean_rows = 512;
    w.mean_blocks = (int)cdiv(total_clips > 0 ? total_clips : 1, w.mean_rows);
    w.mu = reinterpret_cast<float *>(take(128 * 4));
    w.mean_partial = reinterpret_cast<double *>(take((size_t)w.mean_blocks * 128 * 8));
    w.vc = reinterpret_cast<float *>(take((size_t)total_clips * FAST_D * 4));
    w.va = reinterpret_cast<float *>(take((size_t)total_clips * 4));
    w.vb = reinterpret_cast<unsigned short *>(take((size_t)total_clips * 128 * 2));
    // zeroed at every call
    w.zero_base = static_cast<char *>(base) + off;
    const size_t z0 = off;
    w.hash_now = reinterpret_cast<unsigned long long *>(take(8));
    w.stale = reinterpret_cast<int *>(take(4));
    w.fallback = reinterpret_cast<int *>(take((size_t)g * 4));
    w.pairs_total = reinterpret_cast<unsigned long long *>(take(8));
    w.cnt_ws = reinterpret_cast<unsigned long long *>(take((size_t)MAX_RANK * Nq * 8));
    w.wmax = reinterpret_cast<unsigned *>(take((size_t)Nq * 4));
    w.qbound = reinterpret_cast<unsigned *>(take((size_t)Nq * 8));
    w.diff = reinterpret_cast<int *>(take((size_t)Nq * 4));
    w.hist = reinterpret_cast<unsigned *>(take(k > 0 ? (size_t)Nq * MF_HBINS * 4 : 0));
    w.hrange = reinterpret_cast<uint2 *>(take(k > 0 ? (size_t)Nq * 8 : 0));
    w.zero_bytes = off - z0;
    w.qc = reinterpret_cast<float *>(take((size_t)Nq * FAST_D * 4));
    w.qmeta = reinterpret_cast<float4 *>(take((size_t)Nq * 16));
    w.tab = reinterpret_cast<unsigned *>(take((size_t)Nq * 2 * 21 * MF_TAB * 4));
    // ambiguous-pair bitmap: two 64-bit masks per (video, query group), every slot written by exactly one wave per call
    w.amb = reinterpret_cast<ulonglong2 *>(take((size_t)(Nv > 0 ? Nv : 1) * g * sizeof(ulonglong2)));
    w.topk_bytes = carve_topk(nullptr, Nq, Nv, k > 0 ? k + MF_EXTRA : 0).total;
    w.topk = take(w.topk_bytes);
    // the pass on a permutation of the batch (queries sorted by difficulty, score_mfma.h): sorted inputs, sorted outputs
    w.perm = reinterpret_cast<int *>(take((size_t)Nq * 4));
    w.Qs = reinterpret_cast<float *>(take((size_t)Nq * FAST_D * 4));
    w.rds = reinterpret_cast<float *>(take((size_t)MAX_RANK * Nq * 4));
    w.ris = reinterpret_cast<int64_t *>(take((size_t)MAX_RANK * Nq * 8));
    w.seeds = reinterpret_cast<int64_t *>(take((size_t)Nq * 8));
    w.cnts = reinterpret_cast<int64_t *>(take((size_t)MAX_RANK * Nq * 8));
    w.ods = reinterpret_cast<float *>(take((size_t)Nq * (k > 0 ? k : 0) * 4));
    w.ois = reinterpret_cast<int64_t *>(take((size_t)Nq * (k > 0 ? k : 0) * 8));
    w.total = off;
    return w;
}
// threshold of a seeded pass in approximate-key space: seed score + 3 delta (delta for a clip floor of half the seed,
// which the finisher verifies), id = max
__global__ __launch_bounds__(256) void mfma_seed_kernel(const int64_t *__restrict__ seed, const float4 *__restrict__ qmeta,
                                                        int64_t Nq, unsigned long long *__restrict__ thr)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= Nq) return;
    const unsigned long long s = (unsigned long long)seed[q];
    const float x = __uint_as_float((unsigned)(s >> 32));
    unsigned long long t = KEY_MAX;
    if (s < KEY_EMPTY && x < __builtin_inff() && x > 0.0f) {
        const float delta = qmeta[q].z / (0.5f * x) + 190.0f * MF_U * x * 1.02f;
        const float y = next_up((x + 3.0f * delta * 1.0001f) * 1.000001f);
        t = y < __builtin_inff() ? (((unsigned long long)__float_as_uint(y) << 32) | 0xFFFFFFFFull) : KEY_MAX;
    }
    thr[q] = t;
}
// ---- a handful of queries (one serving request) ---------------------------------------------------------------------
constexpr int SMALLQ_CLIPS_MAX = 21;
// The fused kernels put QUERIES on the lanes: with 1-8 queries a wave is 2-12 % full and the pass costs 0.55 ms however
// little there is to score.  Here the lanes are clips and videos: every clip distance once (lane = clip, the canonical chain
// against all NQ queries while the row streams past), then lane = video writes the keys of its moments -- the canonical sums
// and IEEE quotients of the dense kernel -- and counts them against the rank keys; the top-k is two rounds of the pool
// selection of topk_merge_parts_kernel over the key array (P ranges per query, then the P lists).  Exact like every f32 path.
template <int NQ>
__global__ __launch_bounds__(256) void smallq_dist_kernel(const float *__restrict__ Q, int Nq, const float *__restrict__ V,
                                                          int total_clips, int D, float eps, float *__restrict__ dist)
{
    // D = 100 (the model's embedding width; other widths take the general loop below): the lane's whole row is requested at
    // once (25 loads in flight), the queries sit in LDS and are read as broadcasts
    __shared__ __attribute__((aligned(16))) float qs[NQ][FAST_D];
    // blockIdx.y = group of NQ queries: all groups of a request in ONE launch (eight launches of 27 us each were 12 us of
    // arithmetic and 15 of launch gap, ramp and a 3.2-blocks-per-CU round each)
    Q += (int64_t)blockIdx.y * NQ * D;
    dist += (int64_t)blockIdx.y * NQ * total_clips;
    Nq = Nq - (int)blockIdx.y * NQ < NQ ? Nq - (int)blockIdx.y * NQ : NQ;
    const int c = blockIdx.x * 256 + threadIdx.x;
    float acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.0f;
    if (D == FAST_D) {
        for (int i = threadIdx.x; i < NQ * FAST_D; i += 256) qs[i / FAST_D][i % FAST_D] = Q[(int64_t)(i / FAST_D < Nq ? i / FAST_D : 0) * FAST_D + i % FAST_D];
        float4 row[FAST_D / 4];
        const float4 *v4 = reinterpret_cast<const float4 *>(V + (int64_t)(c < total_clips ? c : total_clips - 1) * FAST_D);
#pragma unroll
        for (int k4 = 0; k4 < FAST_D / 4; ++k4) row[k4] = v4[k4];
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < FAST_D / 4; ++k4) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float4 qq = *reinterpret_cast<const float4 *>(&qs[q][4 * k4]);
                float t = (row[k4].x - qq.x) + eps; acc[q] = __builtin_fmaf(t, t, acc[q]);
                t = (row[k4].y - qq.y) + eps; acc[q] = __builtin_fmaf(t, t, acc[q]);
                t = (row[k4].z - qq.z) + eps; acc[q] = __builtin_fmaf(t, t, acc[q]);
                t = (row[k4].w - qq.w) + eps; acc[q] = __builtin_fmaf(t, t, acc[q]);
            }
        }
    } else if (c < total_clips) {
        const float *v = V + (int64_t)c * D;
        for (int k = 0; k < D; ++k) {
            const float vk = v[k];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float t = (vk - Q[(int64_t)(q < Nq ? q : 0) * D + k]) + eps;
                acc[q] = __builtin_fmaf(t, t, acc[q]);
            }
        }
    }
    if (c >= total_clips) return;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (q < Nq) dist[(int64_t)q * total_clips + c] = __builtin_sqrtf(acc[q]);
}

// 16 videos per block, 16 threads per video (thread = the moments starting at clip s, s + 16, ...): canonical sums and
// quotients, keys staged in LDS at their moment index and written out as one contiguous run (the videos' moments are
// adjacent), rank counts reduced over the block
constexpr int SQ_VIDEOS = 16, SQ_MOM = SMALLQ_CLIPS_MAX * (SMALLQ_CLIPS_MAX + 1) / 2;
__global__ __launch_bounds__(256) void smallq_moments_kernel(const float *__restrict__ dist, int Nq, const int32_t *__restrict__ clip_off,
                                                             const int64_t *__restrict__ mom_off, int Nv, int total_clips, int64_t id_base,
                                                             unsigned long long *__restrict__ keys, int64_t Mpad, int num_rank,
                                                             const float *__restrict__ rank_dist, const int64_t *__restrict__ rank_idx,
                                                             int64_t *__restrict__ count_lt, float *__restrict__ dmin)
{
    __shared__ unsigned long long stage[SQ_VIDEOS * SQ_MOM];
    __shared__ float ds[SQ_VIDEOS][SMALLQ_CLIPS_MAX + 3];
    __shared__ int red[MAX_RANK][4];
    const int q = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int vl = threadIdx.x >> 4, sl = threadIdx.x & 15;                   // video within the block, first start clip
    const int v0 = blockIdx.x * SQ_VIDEOS, v = v0 + vl;
    unsigned long long kstar[MAX_RANK] = {0, 0, 0, 0};
    for (int r = 0; r < num_rank; ++r) kstar[r] = make_key(rank_dist[(int64_t)r * Nq + q], (unsigned)rank_idx[(int64_t)r * Nq + q]);
    int nlt[MAX_RANK] = {0, 0, 0, 0};
    const int vhi = v0 + SQ_VIDEOS < Nv ? v0 + SQ_VIDEOS : Nv;
    const int64_t mb0 = mom_off[v0], mb1 = mom_off[vhi];                      // the block's run of moments
    int n = 0, c0 = 0;
    int64_t m0 = 0;
    if (v < Nv) { c0 = clip_off[v]; n = clip_off[v + 1] - c0; m0 = mom_off[v]; }
    float dm = __builtin_inff();
    for (int c = sl; c < n; c += 16) { const float dv = dist[(int64_t)q * total_clips + c0 + c]; ds[vl][c] = dv; dm = dv < dm ? dv : dm; }
    if (dmin) {           // the video's smallest clip distance = the score of its best one-clip moment (top-k selection below)
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { const float t = __shfl_xor(dm, o, 64); dm = t < dm ? t : dm; }
        if (sl == 0 && v < Nv) dmin[(int64_t)q * Nv + v] = dm;
    }
    __syncthreads();
    for (int s = sl; s < n; s += 16) {
        float sum = 0.0f;
        for (int e = s; e < n; ++e) {
            const float de = ds[vl][e];
            sum = e == s ? de : sum + de;
            const float sc = sum / (float)(e - s + 1);
            const int64_t m = m0 + moment_index(n, s, e);
            const unsigned long long key = make_key(sc, (unsigned)(id_base + m));
            stage[m - mb0] = key;
#pragma unroll
            for (int r = 0; r < MAX_RANK; ++r) nlt[r] += key < kstar[r] ? 1 : 0;     // kstar = 0 when unused
        }
    }
    for (int r = 0; r < num_rank; ++r) {
        int x = nlt[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        if (lane == 0) red[r][wv] = x;
    }
    __syncthreads();
    if (keys)
        for (int64_t i = threadIdx.x; i < mb1 - mb0; i += 256) keys[(int64_t)q * Mpad + mb0 + i] = stage[i];
    if ((int)threadIdx.x < num_rank) {
        const int r = threadIdx.x;
        const long long t = (long long)red[r][0] + red[r][1] + red[r][2] + red[r][3];
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(count_lt + (int64_t)r * Nq + q), (unsigned long long)t);
    }
}

// Rank counts + per-video smallest clip distance for the video-selection form (no key array wanted): lane = VIDEO.  The moment
// kernel above gives a video to 16 threads by start clip -- thread 0 of a 21-clip video walks 26 moments, thread 15 six, and every
// moment pays an LDS key store, four 64-bit compares, a run-time moment index and an 11-instruction IEEE division: 0.32 ms at 64
// queries x 10 000 videos, a quarter of that request.  Here a lane keeps its video's distances in registers (+inf past its clip
// count: such moments compare false), the triangle is straight-line code -- the canonical sum (clips left to right from the start
// clip), the quotient by a compile-time length as c_div_small (3 instructions, the IEEE quotient for every sum in [2^-60, 2^100]:
// vfr_math.h), two float compares per rank key -- and the moment id only exists on the slow path: a lane whose score EQUALS a rank
// distance somewhere (the id decides), or whose distances leave that range, redoes its video with keys and IEEE divisions.
// One wave per 64 videos and query; from `score_smallq_rank` (8) queries on -- below, the 16-threads-per-video form has more waves
// in flight for the same work and finishes sooner.
template <int NT, int NR>
__global__ __launch_bounds__(64) void smallq_rank_kernel(const float *__restrict__ dist, int Nq, const int32_t *__restrict__ clip_off,
                                                         const int64_t *__restrict__ mom_off, int Nv, int total_clips, int64_t id_base,
                                                         int num_rank, const float *__restrict__ rank_dist, const int64_t *__restrict__ rank_idx,
                                                         int64_t *__restrict__ count_lt, float *__restrict__ dmin)
{
    const int q = blockIdx.y, lane = threadIdx.x;
    const int v = blockIdx.x * 64 + lane;
    int n = 0, c0 = 0;
    if (v < Nv) { c0 = clip_off[v]; n = clip_off[v + 1] - c0; }
    float d[NT];
    float dm = __builtin_inff(), dx = 0.0f;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const float x = c < n ? dist[(int64_t)q * total_clips + c0 + c] : 0.0f;
        dm = c < n && x < dm ? x : dm;
        dx = x > dx ? x : dx;
        d[c] = c < n ? x : __builtin_inff();
    }
    if (dmin && v < Nv) dmin[(int64_t)q * Nv + v] = dm;
    if (num_rank == 0) return;
    float kd[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) kd[r] = r < num_rank ? rank_dist[(int64_t)r * Nq + q] : -1.0f;     // (scores are >= 0: an unused key counts nothing)
    int nlt[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) nlt[r] = 0;
    // every sum of this video lies in [smallest distance, NT x largest]: inside c_div_small's range?
    bool slow = n > 0 && (dm < C_DIV_SMALL_LO || dx > C_DIV_SMALL_HI / 32.0f);
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        float sum = d[s];
#pragma unroll
        for (int e = s; e < NT; ++e) {
            if (e > s) sum = sum + d[e];
            const float sc = c_div_small(sum, e - s + 1);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                nlt[r] += sc < kd[r] ? 1 : 0;
                slow = slow || sc == kd[r];
            }
        }
    }
    if (slow) {
        unsigned long long kstar[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) { nlt[r] = 0; kstar[r] = r < num_rank ? make_key(kd[r], (unsigned)rank_idx[(int64_t)r * Nq + q]) : 0ull; }
        const int64_t m0 = mom_off[v < Nv ? v : 0];
        for (int s = 0; s < n; ++s) {
            float sum = 0.0f;
            for (int e = s; e < n; ++e) {
                const float de = dist[(int64_t)q * total_clips + c0 + e];
                sum = e == s ? de : sum + de;
                const unsigned long long key = make_key(sum / (float)(e - s + 1), (unsigned)(id_base + m0 + moment_index(n, s, e)));
#pragma unroll
                for (int r = 0; r < NR; ++r) nlt[r] += key < kstar[r] ? 1 : 0;
            }
        }
    }
    for (int r = 0; r < NR; ++r) {
        if (r >= num_rank) break;
        int x = nlt[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        if (lane == 0 && x) atomicAdd(reinterpret_cast<unsigned long long *>(count_lt + (int64_t)r * Nq + q), (unsigned long long)x);
    }
}

// One level of the selection tree over key lists [q][Pin][k]: wave (q, r) reads lists r*F .. r*F + F - 1 of query q as one
// flat range -- the next round's keys requested before the current round is filtered -- keeps the k smallest in the LDS
// pool of the merge kernels and writes them as list r of [q][Pout][k] (or, at the root, as the query's (dist, idx) rows).
template <int KPL>
__global__ __launch_bounds__(256) void topk_tree_kernel(const unsigned long long *__restrict__ in, int64_t Pin, int F, int64_t Pout,
                                                        int64_t n_vq, int k, unsigned long long *__restrict__ out_keys,
                                                        float *__restrict__ out_dist, int64_t *__restrict__ out_idx)
{
    constexpr int CAP = KPL * 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t vq = (int64_t)blockIdx.x * 4 + wv;
    if (vq >= n_vq) return;
    __shared__ unsigned long long pool_s[4][CAP];
    unsigned long long *pool = pool_s[wv];
    unsigned long long key[KPL];
    unsigned long long thr = KEY_MAX;
    int fill = 0;
    auto lds_sync = [&]() { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); };
    auto sort_pool = [&](bool final_pass) {
        lds_sync();
#pragma unroll
        for (int i = 0; i < KPL; ++i) { const int e = i * 64 + lane; key[i] = e < fill ? pool[e] : KEY_MAX; }
        wave_sort<KPL>(key, lane);
        if (final_pass) return;
        lds_sync();
#pragma unroll
        for (int i = 0; i < KPL; ++i) { const int e = i * 64 + lane; if (e < k) pool[e] = key[i]; }
        if (fill >= k) thr = key_at<KPL>(key, k - 1);
        fill = fill < k ? fill : k;
    };
    const int64_t q = vq / Pout, r = vq - q * Pout, first = r * F;
    const int64_t nl = Pin - first < F ? Pin - first : F;
    const unsigned long long *src = in + (q * Pin + first) * k;
    const int64_t total = nl * k;
    // NB rounds of CAP keys are requested together and the next NB before these are filtered: a wave is alone with its
    // range, so the memory latency it pays is per batch of requests, not per round
    constexpr int NB = 4;
    unsigned long long x[NB][KPL], xn[NB][KPL];
    auto load = [&](unsigned long long (&dst)[NB][KPL], int64_t done) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < KPL; ++i) {
                const int64_t gi = done + (b * KPL + i) * 64 + lane;
                const unsigned long long vv = src[gi < total ? gi : total - 1];
                dst[b][i] = (gi < total && vv < KEY_EMPTY) ? vv : KEY_MAX;
            }
    };
    load(x, 0);
    for (int64_t done = 0; done < total; done += NB * CAP) {
        load(xn, done + NB * CAP);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < KPL; ++i) {
                if (fill + 64 > CAP) sort_pool(false);
                const bool pass = x[b][i] < thr;                 // KEY_MAX (empty) never passes
                const unsigned long long m = __ballot(pass);
                if (pass) pool[fill + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = x[b][i];
                fill += __builtin_popcountll(m);
            }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < KPL; ++i) x[b][i] = xn[b][i];
    }
    sort_pool(true);
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const int e = i * 64 + lane;
        if (e < k) {
            const bool ok = key[i] != KEY_MAX;
            if (out_dist) {
                out_dist[vq * k + e] = ok ? __uint_as_float((unsigned)(key[i] >> 32)) : __builtin_inff();
                out_idx[vq * k + e] = ok ? (int64_t)(key[i] & 0xffffffffull) : -1;
            } else {
                out_keys[vq * k + e] = ok ? key[i] : KEY_EMPTY;
            }
        }
    }
}

// Top-k of a handful of queries WITHOUT the key array and the selection tree (three latency-bound levels: 0.09 ms at one
// query).  Every moment's score is a mean of its video's clip distances, so score >= dmin_v * (1 - 22 u) (dmin_v = the
// video's smallest clip distance; each rounding is monotone), and dmin_v itself is the score of a one-clip moment.  Hence any
// thr that is >= the k-th smallest dmin over the videos bounds the k-th best moment, and only videos with
// dmin_v <= thr * (1 + 32 u) can hold a moment of the top-k.  One 1024-thread workgroup per query:
//   A  thr: min / max of the dmin row, a 2048-bin histogram over [min, max], the first bin b* at which the running count
//      reaches k; thr = the largest dmin in bins <= b* (>= k videos lie at or below it).  Nv <= k: thr = +inf (everything);
//   B  the videos passing the margin test are listed in LDS (4096 at a time);
//   C  thread = (listed video, start clip): the canonical sums and IEEE quotients of smallq_moments_kernel, keys <= (thr, max id)
//      appended to the query's candidate run in global memory (the slot from an LDS counter: one workgroup owns the query);
//   D  wave 0 runs the pool selection of the merge kernels over the candidates (typically a few hundred) and writes the rows.
// Exact: the candidates are a superset of the top-k with the dense kernel's keys.
constexpr int SQS_BINS = 2048, SQS_LIST = 4096, SQS_THREADS = 1024;
static_assert(SQS_BINS == 2 * SQS_THREADS, "the bin scan gives every thread two bins");
static_assert(SQS_LIST >= 2 * SQS_THREADS, "the video list must take at least one full chunk of candidates after a partial one");
template <int KPL>
__global__ __launch_bounds__(SQS_THREADS) void smallq_select_kernel(const float *__restrict__ dist, const float *__restrict__ dmin,
                                                                    const int32_t *__restrict__ clip_off, const int64_t *__restrict__ mom_off,
                                                                    int Nv, int total_clips, int64_t id_base, int k,
                                                                    unsigned long long *__restrict__ cand, int64_t cand_stride,
                                                                    float *__restrict__ out_dist, int64_t *__restrict__ out_idx)
{
    constexpr int CAP = KPL * 64;
    __shared__ int hist[SQS_BINS];
    __shared__ int vlist[SQS_LIST];
    __shared__ float redf[2][SQS_THREADS / 64];
    __shared__ int redi[SQS_THREADS / 64];
    __shared__ int s_bstar, s_nlist, s_ncand;
    __shared__ unsigned long long pool[CAP];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *dq = dmin + (int64_t)q * Nv;
    unsigned long long *cq = cand + (int64_t)q * cand_stride;
    // ---- A: threshold ----
    float mn = __builtin_inff(), mx = 0.0f;
    for (int v = tid; v < Nv; v += SQS_THREADS) { const float x = dq[v]; mn = x < mn ? x : mn; mx = (x > mx && x < __builtin_inff()) ? x : mx; }   // (a video without clips: +inf)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64); mn = a < mn ? a : mn; mx = b > mx ? b : mx; }
    if (lane == 0) { redf[0][wv] = mn; redf[1][wv] = mx; }
    for (int i = tid; i < SQS_BINS; i += SQS_THREADS) hist[i] = 0;
    if (tid == 0) { s_nlist = 0; s_ncand = 0; s_bstar = SQS_BINS - 1; }
    __syncthreads();
    mn = redf[0][0]; mx = redf[1][0];
#pragma unroll
    for (int i = 1; i < SQS_THREADS / 64; ++i) { mn = redf[0][i] < mn ? redf[0][i] : mn; mx = redf[1][i] > mx ? redf[1][i] : mx; }
    const float scale = mx > mn ? (float)(SQS_BINS - 1) / (mx - mn) : 0.0f;
    auto bin_of = [&](float x) {
        if (!(x < __builtin_inff())) return SQS_BINS - 1;
        const int b = (int)((x - mn) * scale);
        return b < 0 ? 0 : (b > SQS_BINS - 1 ? SQS_BINS - 1 : b);
    };
    float thr = __builtin_inff();
    if (Nv > k) {
        for (int v = tid; v < Nv; v += SQS_THREADS) atomicAdd(&hist[bin_of(dq[v])], 1);
        __syncthreads();
        // running count over the bins: thread t owns bins 2t, 2t + 1
        const int h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
        int part = h0 + h1, incl = part;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) redi[wv] = incl;
        __syncthreads();
        int before = 0;
        for (int i = 0; i < wv; ++i) before += redi[i];
        const int excl = before + incl - part;                              // videos in bins < 2t
        if (excl < k && excl + h0 >= k) s_bstar = 2 * tid;
        else if (excl + h0 < k && excl + part >= k) s_bstar = 2 * tid + 1;
        __syncthreads();
        const int bstar = s_bstar;
        float tm = 0.0f;
        for (int v = tid; v < Nv; v += SQS_THREADS) { const float x = dq[v]; if (bin_of(x) <= bstar) tm = x > tm ? x : tm; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float a = __shfl_xor(tm, o, 64); tm = a > tm ? a : tm; }
        if (lane == 0) redf[0][wv] = tm;
        __syncthreads();
        tm = redf[0][0];
#pragma unroll
        for (int i = 1; i < SQS_THREADS / 64; ++i) tm = redf[0][i] > tm ? redf[0][i] : tm;
        thr = tm;
    }
    const unsigned long long thrkey = thr < __builtin_inff() ? (((unsigned long long)__float_as_uint(thr) << 32) | 0xFFFFFFFFull) : KEY_MAX;
    const float vthr = thr < __builtin_inff() ? thr * (1.0f + 32.0f * 5.9604645e-8f) : thr;
    // ---- B + C: listed videos, SQS_LIST at a time ----
    for (int vbase = 0; vbase < Nv; ) {
        __syncthreads();
        if (tid == 0) s_nlist = 0;
        __syncthreads();
        // fill the list from vbase on; stop at the chunk of SQS_THREADS videos that would overflow it
        int vend = vbase;
        while (vend < Nv) {
            const int v = vend + tid;
            const bool sel = v < Nv && dq[v] <= vthr;
            if (sel) { const int slot = atomicAdd(&s_nlist, 1); if (slot < SQS_LIST) vlist[slot] = v; }
            vend += SQS_THREADS;
            __syncthreads();
            if (s_nlist + SQS_THREADS > SQS_LIST) break;                     // uniform: the next chunk might not fit
            __syncthreads();
        }
        __syncthreads();
        const int nlist = s_nlist < SQS_LIST ? s_nlist : SQS_LIST;          // (never exceeds: checked before every chunk)
        for (int it = tid; it < nlist * SMALLQ_CLIPS_MAX; it += SQS_THREADS) {
            const int v = vlist[it / SMALLQ_CLIPS_MAX], s0 = it % SMALLQ_CLIPS_MAX;
            const int c0 = clip_off[v], n = clip_off[v + 1] - c0;
            if (s0 >= n) continue;
            const int64_t m0 = mom_off[v];
            const float *dv = dist + (int64_t)q * total_clips + c0;
            float sum = 0.0f;
            for (int e = s0; e < n; ++e) {
                const float de = dv[e];
                sum = e == s0 ? de : sum + de;
                const float sc = sum / (float)(e - s0 + 1);
                const unsigned long long key = make_key(sc, (unsigned)(id_base + m0 + moment_index(n, s0, e)));
                if (key <= thrkey) cq[atomicAdd(&s_ncand, 1)] = key;
            }
        }
        vbase = vend < Nv ? vend : Nv;
    }
    __threadfence_block();
    __syncthreads();
    if (wv != 0) return;
    // ---- D: wave 0 selects the k best of the candidates ----
    const int64_t total = s_ncand;
    unsigned long long key[KPL];
    unsigned long long pthr = KEY_MAX;
    int fill = 0;
    auto lds_sync = [&]() { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); };
    auto sort_pool = [&](bool final_pass) {
        lds_sync();
#pragma unroll
        for (int i = 0; i < KPL; ++i) { const int e = i * 64 + lane; key[i] = e < fill ? pool[e] : KEY_MAX; }
        wave_sort<KPL>(key, lane);
        if (final_pass) return;
        lds_sync();
#pragma unroll
        for (int i = 0; i < KPL; ++i) { const int e = i * 64 + lane; if (e < k) pool[e] = key[i]; }
        if (fill >= k) pthr = key_at<KPL>(key, k - 1);
        fill = fill < k ? fill : k;
    };
    for (int64_t done = 0; done < total; done += CAP) {
        unsigned long long x[KPL];
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const int64_t gi = done + i * 64 + lane;
            const unsigned long long vv = cq[gi < total ? gi : total - 1];
            x[i] = gi < total ? vv : KEY_MAX;
        }
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            if (fill + 64 > CAP) sort_pool(false);
            const bool pass = x[i] < pthr;
            const unsigned long long m = __ballot(pass);
            if (pass) pool[fill + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = x[i];
            fill += __builtin_popcountll(m);
        }
    }
    sort_pool(true);
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const int e = i * 64 + lane;
        if (e < k) {
            const bool ok = key[i] != KEY_MAX;
            out_dist[(int64_t)q * k + e] = ok ? __uint_as_float((unsigned)(key[i] >> 32)) : __builtin_inff();
            out_idx[(int64_t)q * k + e] = ok ? (int64_t)(key[i] & 0xffffffffull) : -1;
        }
    }
}

constexpr int SMALLQ_MAX = 64, SMALLQ_CLIPS = SMALLQ_CLIPS_MAX;
struct SmallqWs { float *dist; unsigned long long *keys, *bufa, *bufb; float *dmin; int64_t Mpad, P0, P1; size_t total; };
constexpr int SQ_F1 = 20, SQ_F = 32;        // lists of k per first-level range; fan-in of the later levels
static SmallqWs carve_smallq(void *base, int64_t Nq, int total_clips, int k)
{
    SmallqWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = static_cast<char *>(base) + off; off += align_up(bytes, 256); return p; };
    const int64_t Mb = (int64_t)total_clips * (SMALLQ_CLIPS + 1) / 2 + 1;            // >= total moments when every video has <= 21 clips
    w.dist = reinterpret_cast<float *>(take((size_t)Nq * total_clips * 4));
    if (k > 0) {
        w.P0 = cdiv(Mb, k);                                                           // the key array as lists of k
        w.Mpad = w.P0 * k;
        w.P1 = cdiv(w.P0, SQ_F1);
        w.keys = reinterpret_cast<unsigned long long *>(take((size_t)Nq * w.Mpad * 8));
        w.bufa = reinterpret_cast<unsigned long long *>(take((size_t)Nq * w.P1 * k * 8));
        w.bufb = reinterpret_cast<unsigned long long *>(take((size_t)Nq * cdiv(w.P1, SQ_F) * k * 8));
        w.dmin = reinterpret_cast<float *>(take((size_t)Nq * total_clips * 4));           // (Nv <= total_clips)
    }
    w.total = off;
    return w;
}
static bool smallq_applicable(int64_t Nq, int Nv, int total_clips, int max_clips, int num_rank, int k)
{
    const int lim = opt_score_smallq() < SMALLQ_MAX ? opt_score_smallq() : SMALLQ_MAX;
    return Nq > 0 && Nq <= lim && Nv > 0 && total_clips > 0 && max_clips <= SMALLQ_CLIPS && num_rank <= MAX_RANK && k <= 448;
}
static int run_smallq(const float *Q, int64_t Nq, const float *V, const int32_t *clip_offsets, const int64_t *moment_offsets, int Nv,
                      int total_clips, int max_clips, int D, float eps, int64_t id_base, int k, float *out_dist, int64_t *out_idx, int num_rank,
                      const float *rank_dist, const int64_t *rank_idx, int64_t *count_lt, void *workspace, vfr_stream_t stream);

}  // namespace vfr

extern "C" {


size_t vfr_score_topk_mfma_workspace_bytes(int64_t Nq, int Nv, int total_clips, int k)
{
    if (Nq < 0 || Nv < 0 || k < 0 || total_clips < 0) return 0;
    const size_t a = vfr::carve_mfma(nullptr, Nq, Nv, total_clips, k).total, b = vfr_score_topk_workspace_bytes(Nq, Nv, k);
    const size_t c = Nq <= vfr::SMALLQ_MAX ? vfr::carve_smallq(nullptr, Nq, total_clips, k).total : 0;
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

int vfr_score_topk_mfma_prefilter(int64_t Nq, int Nv, int total_clips, int max_clips, int D, int num_rank, int k, int dtype)
{
    dtype &= ~VFR_MFMA_BANK_READY;
    if (dtype == VFR_MFMA_F32 && vfr::smallq_applicable(Nq, Nv, total_clips, max_clips, num_rank, k)) return 0;
    return Nq > 0 && k >= 0 && num_rank >= 0 && vfr::mfma_applicable(D, max_clips, num_rank, k, Nv, total_clips) &&
           vfr::mfma_worthwhile(Nv, dtype) && vfr::opt_score_fast();
}

int vfr_score_topk_mfma(const float *Q, int64_t Nq, const float *V, const int32_t *clip_offsets,
                        const int64_t *moment_offsets, int Nv, int total_clips, int min_clips, int max_clips, int D,
                        float eps, int64_t id_base, int k, float *out_dist, int64_t *out_idx, int num_rank,
                        const float *rank_dist, const int64_t *rank_idx, int64_t *count_lt, const int64_t *thr_seed,
                        int dtype, void *workspace, size_t workspace_bytes, vfr_stream_t stream)
{
    const bool bank_ready = (dtype & VFR_MFMA_BANK_READY) != 0;
    dtype &= ~VFR_MFMA_BANK_READY;
    VFR_REQUIRE(dtype == VFR_MFMA_F32 || dtype == VFR_MFMA_BF16, VFR_EINVAL, "vfr_score_topk_mfma: dtype %d (0 = f32, 1 = bf16)", dtype);
    VFR_REQUIRE(workspace && workspace_bytes >= vfr_score_topk_mfma_workspace_bytes(Nq, Nv, total_clips, k), VFR_EWORKSPACE,
                "vfr_score_topk_mfma: workspace %zu < %zu bytes", workspace_bytes,
                vfr_score_topk_mfma_workspace_bytes(Nq, Nv, total_clips, k));
    if (dtype == VFR_MFMA_F32 && Q && V && clip_offsets && moment_offsets && k >= 0 && num_rank >= 0 &&
        vfr::smallq_applicable(Nq, Nv, total_clips, max_clips, num_rank, k)) {
        // a handful of queries: lanes = clips / videos instead of queries (exact; thr_seed is only an accelerator)
        VFR_REQUIRE(k == 0 || (out_dist && out_idx), VFR_EINVAL, "vfr_score_topk_mfma: k > 0 needs out_dist/out_idx");
        VFR_REQUIRE(num_rank == 0 || (rank_dist && rank_idx && count_lt), VFR_EINVAL,
                    "vfr_score_topk_mfma: num_rank > 0 needs rank_dist, rank_idx and count_lt");
        VFR_REQUIRE(id_base >= 0 && D > 0, VFR_EINVAL, "vfr_score_topk_mfma: bad id_base / D");
        return vfr::run_smallq(Q, Nq, V, clip_offsets, moment_offsets, Nv, total_clips, max_clips, D, eps, id_base, k, out_dist, out_idx, num_rank,
                               rank_dist, rank_idx, count_lt, workspace, stream);
    }
    if (!(Q && V && clip_offsets && moment_offsets && Nq > 0 && k >= 0 && num_rank >= 0) ||
        !vfr::mfma_applicable(D, max_clips, num_rank, k, Nv, total_clips) || !vfr::mfma_worthwhile(Nv, dtype) || !vfr::opt_score_fast()) {
        // shapes the pre-filter is not built for: the exact kernels give the same results in f32 mode
        VFR_REQUIRE(dtype == VFR_MFMA_F32, VFR_EUNSUPPORTED,
                    "vfr_score_topk_mfma(bf16): needs D = 100, <= 21 clips per video, 0 or 2 rank keys, k <= %d", 512 - 231 - vfr::MF_EXTRA);
        return vfr_score_topk_f32(Q, Nq, V, clip_offsets, moment_offsets, Nv, total_clips, min_clips, max_clips, D, eps, id_base, k,
                                  out_dist, out_idx, num_rank, rank_dist, rank_idx, count_lt, thr_seed, workspace, workspace_bytes, stream);
    }
    VFR_REQUIRE(k == 0 || (out_dist && out_idx), VFR_EINVAL, "vfr_score_topk_mfma: k > 0 needs out_dist/out_idx");
    VFR_REQUIRE(num_rank == 0 || (rank_dist && rank_idx && count_lt), VFR_EINVAL,
                "vfr_score_topk_mfma: num_rank > 0 needs rank_dist, rank_idx and count_lt");
    VFR_REQUIRE(id_base >= 0, VFR_EINVAL, "vfr_score_topk_mfma: bad id_base");
    hipStream_t st = vfr::as_stream(stream);
    const bool bf16 = dtype == VFR_MFMA_BF16;
    const int NT = max_clips <= 6 ? 6 : 21;
    const int kp = k > 0 ? k + vfr::MF_EXTRA : 0;
    vfr::MfmaWs mw = vfr::carve_mfma(workspace, Nq, Nv, total_clips, k);
    vfr::TopkWs w = vfr::carve_topk(mw.topk, Nq, Nv, kp);
    // the difficulty sample (8 videos per query: ~10 us) runs for every pass with rank keys -- a wave whose queries are all hard
    // switches the early-out test off (the test costs ~6 % of the fused launch where it never fires) --; the SORT on top of it
    // (rank, gather, scatter: ~0.06 ms) from 1024 queries x 4096 videos on (measured with tools/rank_sim.py: worth it at a rank's
    // 5 000 videos, not at 2 500)
    const bool diff_plan = num_rank > 0 && vfr::opt_score_defer() >= 0 && vfr::opt_score_sort() && Nv >= 64 && D == vfr::FAST_D &&
                           ((((uintptr_t)V) | ((uintptr_t)Q)) & 15) == 0;
    const bool sorted_plan = diff_plan && Nq >= (vfr::opt_score_sort() > 1 ? 128 : 1024) && Nq <= vfr::SORT_MAX_QUERIES &&
                             (Nv >= 4096 || vfr::opt_score_sort() > 1);
    {
        // one launch: the zeroed region, the +inf thresholds of an unseeded top-k pass, the sorted pass's count buffer
        const vfr::FillJob jobs[3] = {{mw.zero_base, nullptr, mw.zero_bytes, 0u},
                                      {w.thr, nullptr, (k > 0 && !thr_seed) ? (size_t)Nq * 8 : 0, 0xFFFFFFFFu},
                                      {mw.cnts, nullptr, sorted_plan ? (size_t)num_rank * Nq * 8 : 0, 0u}};
        if (int rc = vfr::fill_regions(jobs, 3, st)) return rc;
    }
    // ---- queries sorted by difficulty (score_mfma.h "Query order"): the pass below runs on sorted copies of the per-query inputs
    // and writes sorted outputs, which the last kernel of the call scatters back (counts added) ----
    // (worth its fixed ~0.08 ms -- sample pass, rank, gather, scatter -- only on a long pass: from ~2000 videos x 1000 queries on)
    const bool sorted = sorted_plan;
    float *const out_dist_user = out_dist;
    int64_t *const out_idx_user = out_idx, *const count_lt_user = count_lt;
    if (diff_plan) {
        vfr::ProfScope prof(vfr::SITE_SCORE_PREP, st);
        hipLaunchKernelGGL(vfr::mfma_difficulty_kernel, dim3((unsigned)(vfr::cdiv(Nq, 64) * vfr::SORT_SAMPLE)), dim3(64), 0, st, Q, Nq, V,
                           clip_offsets, Nv, num_rank, rank_dist, mw.diff);
        VFR_CHECK_LAUNCH("mfma_difficulty_kernel");
    }
    if (sorted) {
        vfr::ProfScope prof(vfr::SITE_SCORE_PREP, st);
        hipLaunchKernelGGL(vfr::mfma_sort_perm_kernel, dim3((unsigned)vfr::cdiv(Nq, 16)), dim3(256), 0, st, mw.diff, (int)Nq, mw.perm);
        hipLaunchKernelGGL(vfr::mfma_gather_queries_kernel, dim3((unsigned)vfr::cdiv(Nq * vfr::FAST_D, 256)), dim3(256), 0, st, mw.perm, Nq, Q,
                           num_rank, rank_dist, rank_idx, thr_seed, mw.Qs, mw.rds, mw.ris, mw.seeds);
        VFR_CHECK_LAUNCH("query sort");
        Q = mw.Qs; rank_dist = mw.rds; rank_idx = mw.ris; count_lt = mw.cnts;
        if (thr_seed) thr_seed = mw.seeds;
        if (k > 0) { out_dist = mw.ods; out_idx = mw.ois; }
    }
    auto unsort = [&]() -> int {
        if (!sorted) return VFR_OK;
        vfr::ProfScope prof(vfr::SITE_SCORE_FINISH, st);
        const int per = k > num_rank ? k : num_rank;
        hipLaunchKernelGGL(vfr::mfma_scatter_results_kernel, dim3((unsigned)vfr::cdiv(Nq * per, 256)), dim3(256), 0, st, mw.perm, Nq, k, num_rank,
                           mw.ods, mw.ois, mw.cnts, out_dist_user, out_idx_user, count_lt_user);
        VFR_CHECK_LAUNCH("mfma_scatter_results_kernel");
        return VFR_OK;
    };
    vfr::MfmaArgs m{};
    m.va = mw.va; m.qmeta = mw.qmeta; m.tab = mw.tab; m.cnt_ws = mw.cnt_ws; m.amb = mw.amb; m.pairs_total = mw.pairs_total; m.wmax = mw.wmax;
    m.fallback = mw.fallback; m.vb = bf16 ? mw.vb : nullptr; m.rv = mw.rv; m.vc = mw.vc; m.qc = mw.qc;
    m.qbound = mw.qbound; m.defer_max = vfr::opt_score_defer();
    m.diff = diff_plan ? mw.diff : nullptr; m.perm = sorted ? mw.perm : nullptr;
    m.hist = (k > 0 && vfr::opt_score_hist()) ? mw.hist : nullptr; m.hrange = mw.hrange;
    {
        vfr::ProfScope prof(vfr::SITE_SCORE_PREP, st);
        {
            // bank side: mean, centred rows, squared norms, the largest norm (bf16: the rounded copy) -- recomputed unless the
            // caller claims BANK_READY AND the bank still hashes to the signature stored with the products (score_mfma.h)
            vfr::BankSig now{};
            now.total_clips = total_clips; now.Nv = Nv; now.D = D; now.eps_bits = __builtin_bit_cast(unsigned, eps); now.has_bf16 = bf16 ? 1 : 0;
            const int64_t nwords = (int64_t)total_clips * D;
            int hb = (int)vfr::cdiv(nwords, 256 * 4 * 8);
            hb = hb < 1 ? 1 : (hb > 2048 ? 2048 : hb);
            hipLaunchKernelGGL(vfr::mfma_bank_hash_kernel, dim3((unsigned)hb), dim3(256), 0, st, reinterpret_cast<const unsigned *>(V), nwords,
                               reinterpret_cast<const unsigned *>(clip_offsets), (int64_t)Nv + 1, mw.hash_now);
            hipLaunchKernelGGL(vfr::mfma_bank_check_kernel, dim3(1), dim3(1), 0, st, mw.hash_now, now, bank_ready ? 1 : 0, mw.sig, mw.stale, mw.rv);
            hipLaunchKernelGGL(vfr::mfma_mean_partial_kernel, dim3((unsigned)mw.mean_blocks), dim3(1024), 0, st, V, total_clips, D, mw.mean_rows,
                               mw.mean_partial, mw.stale);
            hipLaunchKernelGGL(vfr::mfma_mean_final_kernel, dim3(1), dim3(1024), 0, st, mw.mean_partial, mw.mean_blocks, D, total_clips, mw.mu, mw.stale);
            hipLaunchKernelGGL(vfr::mfma_prep_v_kernel, dim3((unsigned)vfr::cdiv(total_clips, 4)), dim3(256), 0, st, V, total_clips, D, eps,
                               mw.mu, mw.vc, mw.va, mw.rv, bf16 ? mw.vb : nullptr, mw.stale);
        }
        hipLaunchKernelGGL(vfr::mfma_prep_q_kernel, dim3((unsigned)vfr::cdiv(Nq, 16)), dim3(256), 0, st, Q, Nq, D, eps, mw.mu, mw.qc, mw.rv,
                           num_rank, rank_dist, mw.qmeta);
        if (num_rank > 0) {
            const dim3 tg((unsigned)vfr::cdiv(Nq * num_rank * NT, 256));
            if (NT == 6)
                hipLaunchKernelGGL((vfr::mfma_prep_tab_kernel<6>), tg, dim3(256), 0, st, Nq, num_rank, rank_dist, mw.qmeta, mw.tab, bf16 ? 1 : 0, mw.fallback, mw.wmax, mw.qbound);
            else
                hipLaunchKernelGGL((vfr::mfma_prep_tab_kernel<21>), tg, dim3(256), 0, st, Nq, num_rank, rank_dist, mw.qmeta, mw.tab, bf16 ? 1 : 0, mw.fallback, mw.wmax, mw.qbound);
        }
        if (k > 0) {
            if (thr_seed)
                hipLaunchKernelGGL(vfr::mfma_seed_kernel, dim3((unsigned)vfr::cdiv(Nq, 256)), dim3(256), 0, st, thr_seed, mw.qmeta, Nq, w.thr);

        }
        VFR_CHECK_LAUNCH("mfma pre-pass");
    }
    vfr::ScoreArgs a{};
    a.Q = Q; a.Nq = Nq; a.V = V; a.clip_off = clip_offsets; a.mom_off = moment_offsets; a.Nv = Nv; a.D = D; a.eps = eps;
    a.id_base = id_base; a.k = kp; a.num_rank = num_rank; a.rank_dist = rank_dist; a.rank_idx = rank_idx; a.count_lt = count_lt;
    a.ds_rows = max_clips < 1 ? 1 : max_clips;
    a.total_clips = total_clips;
    a.min_clips = min_clips;
    a.mf_host = &m; a.mf_bf16 = bf16 ? 1 : 0;
    if (int rc = vfr::run_pass(a, w, thr_seed != nullptr, nullptr, nullptr, kp > 0 ? w.pre_keys : nullptr, st)) return rc;
    int g, c;
    vfr::plan_tasks(Nq, Nv, &g, &c);
    a.num_groups = g; a.num_chunks = c;
    if (num_rank > 0 && !bf16) {
        vfr::ProfScope prof(vfr::SITE_SCORE_PAIRS, st);
        // a wave per video; with one or two query groups (Nq <= 128) a wave per 8 consecutive videos, whose marked pairs share its
        // batches (measured: 0.152 against 0.177 ms at 64 queries; from 256 queries on the one-video form is faster again -- its clip
        // rows are LDS broadcasts, the eight-video form's are per-lane reads and its 76 KB of LDS leave two waves per CU)
        const bool multi = g <= 2;
        const dim3 pgrid((unsigned)(multi ? vfr::cdiv(Nv, 8) : Nv));
#define VFR_PV(NTV, VBV) hipLaunchKernelGGL((vfr::score_pairs_video_kernel<NTV, 2, VBV>), pgrid, dim3(64), 0, st, Q, V, clip_offsets, moment_offsets, \
                                            rank_dist, rank_idx, a, m)
        if (NT == 6) { if (multi) VFR_PV(6, 8); else VFR_PV(6, 1); }
        else         { if (multi) VFR_PV(21, 8); else VFR_PV(21, 1); }
#undef VFR_PV
        VFR_CHECK_LAUNCH("score_pairs_video_kernel");
    }
    {
        vfr::ProfScope prof(vfr::SITE_SCORE_FINISH, st);
        if (k > 0) {
            const int uniform = (min_clips == max_clips && min_clips > 0) ? max_clips : 0;
            dim3 grid((unsigned)vfr::cdiv(Nq, 4)), block(256);
            if (kp <= 256)
                hipLaunchKernelGGL((vfr::topk_finish_kernel<4>), grid, block, 0, st, w.pre_keys, kp, k, Q, Nq, V, clip_offsets, moment_offsets, Nv,
                                   uniform, D, eps, id_base, mw.qmeta, mw.fallback, thr_seed, out_dist, out_idx, bf16 ? 0 : 1);
            else
                hipLaunchKernelGGL((vfr::topk_finish_kernel<8>), grid, block, 0, st, w.pre_keys, kp, k, Q, Nq, V, clip_offsets, moment_offsets, Nv,
                                   uniform, D, eps, id_base, mw.qmeta, mw.fallback, thr_seed, out_dist, out_idx, bf16 ? 0 : 1);
        }
        if (num_rank > 0)
            hipLaunchKernelGGL(vfr::mfma_commit_counts_kernel, dim3((unsigned)vfr::cdiv((int64_t)num_rank * Nq, 256)), dim3(256), 0, st, mw.cnt_ws,
                               num_rank, Nq, bf16 ? nullptr : mw.fallback, count_lt);
        VFR_CHECK_LAUNCH("mfma finisher");
    }
    if (bf16) return unsort();
    // ---- flagged query groups: the exact kernels, every launch limited to them (returns at once when nothing is flagged) ----
    vfr::TopkWs wx = vfr::carve_topk(mw.topk, Nq, Nv, k);
    vfr::ScoreArgs x{};
    x.Q = Q; x.Nq = Nq; x.V = V; x.clip_off = clip_offsets; x.mom_off = moment_offsets; x.Nv = Nv; x.D = D; x.eps = eps;
    x.id_base = id_base; x.k = k; x.num_rank = num_rank; x.rank_dist = rank_dist; x.rank_idx = rank_idx; x.count_lt = count_lt;
    x.ds_rows = max_clips < 1 ? 1 : max_clips; x.total_clips = total_clips; x.min_clips = min_clips;
    x.group_mask = mw.fallback;
    x.prof_site = vfr::SITE_SCORE_FALLBACK;
    if (k > 0) {
        if (int rc = thr_seed ? vfr::copy_region(wx.thr, thr_seed, (size_t)Nq * 8, st) : vfr::fill_region(wx.thr, 0xFFFFFFFFu, (size_t)Nq * 8, st)) return rc;
    }
    if (int rc = vfr::run_pass(x, wx, thr_seed != nullptr, out_dist, out_idx, nullptr, st)) return rc;
    return unsort();
}


int vfr_score_topk_mfma_stats(const void *workspace, int64_t Nq, int Nv, int total_clips, int k, int64_t *stats_host,
                              vfr_stream_t stream)
{
    VFR_REQUIRE(workspace && stats_host && Nq > 0 && Nv > 0, VFR_EINVAL, "vfr_score_topk_mfma_stats: bad argument");
    hipStream_t st = vfr::as_stream(stream);
    vfr::MfmaWs mw = vfr::carve_mfma(const_cast<void *>(workspace), Nq, Nv, total_clips, k);
    std::vector<int> fb(mw.groups);
    unsigned long long np = 0;
    if (hipMemcpyAsync(fb.data(), mw.fallback, fb.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&np, mw.pairs_total, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return vfr::fail(VFR_EHIP, "vfr_score_topk_mfma_stats: copy failed");
    int64_t nf = 0;
    for (int v : fb) nf += v != 0;
    // [3]: pairs the bitmap can mark (every (query, video) pair: the marking has no capacity limit)
    stats_host[0] = mw.groups; stats_host[1] = nf; stats_host[2] = (int64_t)np; stats_host[3] = (int64_t)Nq * Nv;
    if (getenv("VFR_MFMA_DEBUG")) {
        float rv = 0, qm[4]; unsigned tab[2 * 21 * 4];
        (void)hipMemcpy(&rv, mw.rv, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(qm, mw.qmeta, 16, hipMemcpyDeviceToHost);
        (void)hipMemcpy(tab, mw.tab, sizeof tab, hipMemcpyDeviceToHost);
        fprintf(stderr, "[mfma] R %.6g  q0: bq %.6g dfl %.6g E2 %.6g |q| %.6g\n", rv, qm[0], qm[1], qm[2], qm[3]);
        for (int i = 0; i < 2 * 21; ++i)
            fprintf(stderr, "[mfma] tab[%d] LOX %08x HIX %08x LOW %08x width %u  (lo sum %.7g)\n", i, tab[4 * i], tab[4 * i + 1], tab[4 * i + 2],
                    tab[4 * i + 3], __builtin_bit_cast(float, tab[4 * i] - 1u));
    }
    return VFR_OK;
}

static int launch_merge_parts(const float *pd, const int64_t *pi, bool packed, int G, int64_t Nq, int k, float *od,
                              int64_t *oi, int64_t *okeys, vfr_stream_t stream, int64_t slot_stride = 0, int64_t q_stride = 0)
{
    if (slot_stride == 0) slot_stride = Nq * k;
    if (q_stride == 0) q_stride = k;
    dim3 grid((unsigned)vfr::cdiv(Nq, 4)), block(256);
    const int kpl = vfr::kpl_for(k);
    vfr::ProfScope prof(vfr::SITE_EXCHANGE, vfr::as_stream(stream));
#define VFR_MERGE(KPL, PACKED) hipLaunchKernelGGL((vfr::topk_merge_parts_kernel<KPL, PACKED>), grid, block, 0, \
                                                  vfr::as_stream(stream), pd, pi, slot_stride, q_stride, G, Nq, k, od, oi, okeys)
    if (packed) { if (kpl == 4) VFR_MERGE(4, true); else VFR_MERGE(8, true); }
    else        { if (kpl == 4) VFR_MERGE(4, false); else VFR_MERGE(8, false); }
#undef VFR_MERGE
    VFR_CHECK_LAUNCH("topk_merge_parts_kernel");
    return VFR_OK;
}

int vfr_topk_merge_f32(const float *part_dist, const int64_t *part_idx, int G, int64_t Nq, int k, float *out_dist,
                       int64_t *out_idx, vfr_stream_t stream)
{
    VFR_REQUIRE(part_dist && part_idx && out_dist && out_idx && G > 0 && Nq >= 0 && k > 0, VFR_EINVAL,
                "vfr_topk_merge_f32: bad argument");
    VFR_REQUIRE(k <= 448, VFR_EUNSUPPORTED, "vfr_topk_merge_f32: k=%d > 448", k);
    VFR_REQUIRE((int64_t)G * k < (1ll << 31), VFR_EUNSUPPORTED, "vfr_topk_merge_f32: G*k too large");
    if (Nq == 0) return VFR_OK;
    return launch_merge_parts(part_dist, part_idx, false, G, Nq, k, out_dist, out_idx, nullptr, stream);
}

int vfr_topk_pack_keys(const float *dist, const int64_t *idx, int64_t n, int64_t *keys, vfr_stream_t stream)
{
    VFR_REQUIRE(n >= 0 && (n == 0 || (dist && idx && keys)), VFR_EINVAL, "vfr_topk_pack_keys: bad argument");
    if (n == 0) return VFR_OK;
    vfr::ProfScope prof(vfr::SITE_EXCHANGE, vfr::as_stream(stream));
    hipLaunchKernelGGL(vfr::topk_pack_keys_kernel, dim3((unsigned)vfr::cdiv(n, 256)), dim3(256), 0,
                       vfr::as_stream(stream), dist, idx, n, keys);
    VFR_CHECK_LAUNCH("topk_pack_keys_kernel");
    return VFR_OK;
}

int vfr_topk_merge_keys_strided(const int64_t *part_keys, int64_t slot_stride, int G, int64_t Nq, int k, float *out_dist,
                                int64_t *out_idx, int64_t *out_keys, vfr_stream_t stream)
{
    VFR_REQUIRE(part_keys && G > 0 && Nq >= 0 && k > 0 && ((out_dist && out_idx) || out_keys) &&
                (out_dist == nullptr) == (out_idx == nullptr) && slot_stride >= Nq * k, VFR_EINVAL,
                "vfr_topk_merge_keys: bad argument");
    VFR_REQUIRE(k <= 448, VFR_EUNSUPPORTED, "vfr_topk_merge_keys: k=%d > 448", k);
    VFR_REQUIRE((int64_t)G * k < (1ll << 31), VFR_EUNSUPPORTED, "vfr_topk_merge_keys: G*k too large");
    if (Nq == 0) return VFR_OK;
    return launch_merge_parts(nullptr, part_keys, true, G, Nq, k, out_dist, out_idx, out_keys, stream, slot_stride);
}

int vfr_topk_merge_keys(const int64_t *part_keys, int G, int64_t Nq, int k, float *out_dist, int64_t *out_idx,
                        int64_t *out_keys, vfr_stream_t stream)
{
    return vfr_topk_merge_keys_strided(part_keys, Nq * k, G, Nq, k, out_dist, out_idx, out_keys, stream);
}

int vfr_gt_best_keys_f32(const float *own_scores, int64_t n_sel, int M, int score_stride, const uint8_t *labels, int R,
                         int label_stride, const int64_t *id_base, const int64_t *sel, int64_t Nq, int64_t *keys,
                         vfr_stream_t stream)
{
    VFR_REQUIRE(keys && R > 0 && Nq >= 0 && n_sel >= 0 && M >= 0 && score_stride >= M && label_stride >= M, VFR_EINVAL,
                "vfr_gt_best_keys_f32: bad argument");
    VFR_REQUIRE(n_sel == 0 || (own_scores && labels && id_base && sel), VFR_EINVAL, "vfr_gt_best_keys_f32: null input");
    if (Nq == 0) return VFR_OK;
    vfr::ProfScope prof(vfr::SITE_EXCHANGE, vfr::as_stream(stream));
    hipLaunchKernelGGL(vfr::gt_fill_keys_kernel, dim3((unsigned)vfr::cdiv((int64_t)R * Nq, 256)), dim3(256), 0,
                       vfr::as_stream(stream), (int64_t)R * Nq, keys);
    VFR_CHECK_LAUNCH("gt_fill_keys_kernel");
    if (n_sel == 0) return VFR_OK;
    hipLaunchKernelGGL(vfr::gt_best_keys_kernel, dim3((unsigned)vfr::cdiv(n_sel, 4)), dim3(256), 0, vfr::as_stream(stream),
                       own_scores, n_sel, M, score_stride, labels, R, label_stride, id_base, sel, Nq, keys);
    VFR_CHECK_LAUNCH("gt_best_keys_kernel");
    return VFR_OK;
}

int vfr_gt_rank_keys_f32(const float *own_scores, int64_t n_sel, int M, int score_stride, const uint8_t *labels, int R,
                         int label_stride, const int64_t *id_base, const int64_t *sel, int64_t Nq, int64_t *keys,
                         float *rank_dist, int64_t *rank_idx, int64_t *count_zero, int *missing, vfr_stream_t stream)
{
    VFR_REQUIRE(keys && R > 0 && Nq >= 0 && n_sel >= 0 && M >= 0 && score_stride >= M && label_stride >= M, VFR_EINVAL,
                "vfr_gt_rank_keys_f32: bad argument");
    VFR_REQUIRE(n_sel == 0 || (own_scores && labels && id_base && sel), VFR_EINVAL, "vfr_gt_rank_keys_f32: null input");
    VFR_REQUIRE((rank_dist == nullptr) == (rank_idx == nullptr), VFR_EINVAL, "vfr_gt_rank_keys_f32: rank_dist and rank_idx go together");
    if (Nq == 0) return VFR_OK;
    vfr::ProfScope prof(vfr::SITE_EXCHANGE, vfr::as_stream(stream));
    hipLaunchKernelGGL(vfr::gt_fill_rank_kernel, dim3((unsigned)vfr::cdiv((int64_t)R * Nq, 256)), dim3(256), 0, vfr::as_stream(stream),
                       (int64_t)R * Nq, keys, rank_dist, rank_idx, count_zero, missing);
    VFR_CHECK_LAUNCH("gt_fill_rank_kernel");
    if (n_sel == 0) return VFR_OK;
    hipLaunchKernelGGL(vfr::gt_best_keys_kernel, dim3((unsigned)vfr::cdiv(n_sel, 4)), dim3(256), 0, vfr::as_stream(stream),
                       own_scores, n_sel, M, score_stride, labels, R, label_stride, id_base, sel, Nq, keys, rank_dist, rank_idx, missing);
    VFR_CHECK_LAUNCH("gt_best_keys_kernel");
    return VFR_OK;
}

int vfr_gt_labels_u8(const int32_t *times, const int32_t *nannot, const int32_t *n_own, int64_t Nq, int A,
                     const double *thresholds_host, int R, int strict, int Mmax, uint8_t *labels, vfr_stream_t stream)
{
    VFR_REQUIRE(Nq >= 0 && A >= 0 && Mmax >= 0 && R > 0 && thresholds_host, VFR_EINVAL, "vfr_gt_labels_u8: bad argument");
    VFR_REQUIRE(R <= vfr::MAX_LABEL_THR, VFR_EUNSUPPORTED, "vfr_gt_labels_u8: R=%d > %d thresholds", R, vfr::MAX_LABEL_THR);
    if (Nq == 0 || Mmax == 0) return VFR_OK;
    VFR_REQUIRE(times && nannot && n_own && labels, VFR_EINVAL, "vfr_gt_labels_u8: null pointer");
    vfr::LabelThr t{};
    for (int r = 0; r < R; ++r) t.t[r] = thresholds_host[r];
    vfr::ProfScope prof(vfr::SITE_EXCHANGE, vfr::as_stream(stream));
    hipLaunchKernelGGL(vfr::gt_labels_kernel, dim3((unsigned)vfr::cdiv(Nq * Mmax, 256)), dim3(256), 0, vfr::as_stream(stream),
                       times, nannot, n_own, Nq, A, R, t, strict, Mmax, labels);
    VFR_CHECK_LAUNCH("gt_labels_kernel");
    return VFR_OK;
}

}  // extern "C"

namespace vfr {
// Device self-check of what the pre-filter's margins (and every MFMA GEMM's bit-exactness) rest on: one
// v_mfma_f32_16x16x4_f32 is four fp32 fmas, k ascending, on top of its accumulator -- no wider intermediate, no other order,
// denormal operands and results kept.  A 16 x 16 tile over K is computed by the matrix pipe and, element by element, as an
// explicit fmaf chain on the vector ALU (`reversed`: the chain run k-descending -- what a failing check looks like); the
// number of elements whose bits differ goes to *mismatches.
__global__ __launch_bounds__(64) void mfma_selfcheck_kernel(const float *__restrict__ A, const float *__restrict__ B, int K, int reversed,
                                                            int *__restrict__ mismatches)
{
    const int lane = threadIdx.x, l15 = lane & 15, lq = lane >> 4;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s = 0; s < K / 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[l15 * K + 4 * s + lq], B[l15 * K + 4 * s + lq], acc, 0, 0, 0);
    int bad = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * lq + r, j = l15;
        float ref = 0.0f;
        for (int kk = 0; kk < K; ++kk) {
            const int k = reversed ? K - 1 - kk : kk;
            ref = __builtin_fmaf(A[i * K + k], B[j * K + k], ref);
        }
        bad += __float_as_uint(ref) != __float_as_uint(acc[r]) ? 1 : 0;
    }
    if (bad) atomicAdd(mismatches, bad);
}
static int run_smallq(const float *Q, int64_t Nq, const float *V, const int32_t *clip_offsets, const int64_t *moment_offsets, int Nv,
                      int total_clips, int max_clips, int D, float eps, int64_t id_base, int k, float *out_dist, int64_t *out_idx, int num_rank,
                      const float *rank_dist, const int64_t *rank_idx, int64_t *count_lt, void *workspace, vfr_stream_t stream)
{
    hipStream_t st = as_stream(stream);
    SmallqWs w = carve_smallq(workspace, Nq, total_clips, k);
    {
        ProfScope prof(SITE_SCORE_FUSED, st);
        const dim3 grid((unsigned)cdiv(total_clips, 256));
        // eight queries per pass over the bank (sixteen were measured: 0.574 against 0.542 ms at 64 queries, 0.194 against 0.184 at
        // 16 -- the kernel is bound by its 3 VALU issues per (query, clip, dimension), not by the row reads, and the wider
        // form's registers cost a wave of occupancy)
        {
            const dim3 g8(grid.x, (unsigned)cdiv(Nq, 8));
            if (Nq == 1)      hipLaunchKernelGGL(smallq_dist_kernel<1>, grid, dim3(256), 0, st, Q, (int)Nq, V, total_clips, D, eps, w.dist);
            else if (Nq == 2) hipLaunchKernelGGL(smallq_dist_kernel<2>, grid, dim3(256), 0, st, Q, (int)Nq, V, total_clips, D, eps, w.dist);
            else if (Nq <= 4) hipLaunchKernelGGL(smallq_dist_kernel<4>, grid, dim3(256), 0, st, Q, (int)Nq, V, total_clips, D, eps, w.dist);
            else              hipLaunchKernelGGL(smallq_dist_kernel<8>, g8, dim3(256), 0, st, Q, (int)Nq, V, total_clips, D, eps, w.dist);
        }
        // top-k by video selection (smallq_select_kernel) unless switched off: then the key array + selection tree
        const bool select = k > 0 && opt_score_smallq_select();
        // unwritten key slots (the padding past the last moment) must read as empty: any value >= KEY_EMPTY does
        if (k > 0 && !select && hipMemsetAsync(w.keys, 0xFF, (size_t)Nq * w.Mpad * 8, st) != hipSuccess)
            return fail(VFR_EHIP, "vfr_score_topk_mfma: hipMemsetAsync failed");
        if (select && opt_score_smallq_rank() > 0 && Nq >= opt_score_smallq_rank()) {
            // video-selection form: no key array -- rank counts and the per-video smallest distance with lane = video
            const dim3 rgrid((unsigned)cdiv(Nv, 64), (unsigned)Nq);
#define VFR_SQR(NTV, NRV) hipLaunchKernelGGL((smallq_rank_kernel<NTV, NRV>), rgrid, dim3(64), 0, st, w.dist, (int)Nq, clip_offsets, moment_offsets, Nv, total_clips, \
                                             id_base, num_rank, rank_dist, rank_idx, count_lt, w.dmin)
            if (max_clips <= 6) { if (num_rank <= 2) VFR_SQR(6, 2); else VFR_SQR(6, MAX_RANK); }
            else                { if (num_rank <= 2) VFR_SQR(SMALLQ_CLIPS_MAX, 2); else VFR_SQR(SMALLQ_CLIPS_MAX, MAX_RANK); }
#undef VFR_SQR
        } else
        if (k > 0 || num_rank > 0)
            hipLaunchKernelGGL(smallq_moments_kernel, dim3((unsigned)cdiv(Nv, SQ_VIDEOS), (unsigned)Nq), dim3(256), 0, st, w.dist, (int)Nq, clip_offsets,
                               moment_offsets, Nv, total_clips, id_base, (k > 0 && !select) ? w.keys : nullptr, w.Mpad, num_rank, rank_dist, rank_idx,
                               count_lt, select ? w.dmin : nullptr);
        VFR_CHECK_LAUNCH("smallq kernels");
    }
    {
        const bool select = k > 0 && opt_score_smallq_select();
        if (select) {
            ProfScope prof2(SITE_TOPK_MERGE, st);
            if (kpl_for(k) == 4)
                hipLaunchKernelGGL(smallq_select_kernel<4>, dim3((unsigned)Nq), dim3(SQS_THREADS), 0, st, w.dist, w.dmin, clip_offsets, moment_offsets,
                                   Nv, total_clips, id_base, k, w.keys, w.Mpad, out_dist, out_idx);
            else
                hipLaunchKernelGGL(smallq_select_kernel<8>, dim3((unsigned)Nq), dim3(SQS_THREADS), 0, st, w.dist, w.dmin, clip_offsets, moment_offsets,
                                   Nv, total_clips, id_base, k, w.keys, w.Mpad, out_dist, out_idx);
            VFR_CHECK_LAUNCH("smallq_select_kernel");
            return VFR_OK;
        }
    }
    if (k == 0) return VFR_OK;
    // selection tree: ranges of SQ_F1 lists of k keys -> lists of k, then SQ_F lists at a time until one list per query is left
    ProfScope prof(SITE_TOPK_MERGE, st);
    const int kpl = kpl_for(k);
    const unsigned long long *in = w.keys;
    int64_t Pin = w.P0;
    int F = SQ_F1;
    unsigned long long *bufs[2] = {w.bufa, w.bufb};
    for (int level = 0;; ++level) {
        const int64_t Pout = cdiv(Pin, F), n_vq = Nq * Pout;
        const bool root = Pout == 1;
        unsigned long long *out = root ? nullptr : bufs[level & 1];
        const dim3 grid((unsigned)cdiv(n_vq, 4));
        if (kpl == 4) hipLaunchKernelGGL(topk_tree_kernel<4>, grid, dim3(256), 0, st, in, Pin, F, Pout, n_vq, k, out, root ? out_dist : nullptr, root ? out_idx : nullptr);
        else          hipLaunchKernelGGL(topk_tree_kernel<8>, grid, dim3(256), 0, st, in, Pin, F, Pout, n_vq, k, out, root ? out_dist : nullptr, root ? out_idx : nullptr);
        if (root) break;
        in = out; Pin = Pout; F = Pin <= 2 * SQ_F ? (int)Pin : SQ_F;
    }
    VFR_CHECK_LAUNCH("topk_tree_kernel");
    return VFR_OK;
}
}  // namespace vfr

extern "C" int vfr_mfma_selfcheck(const float *A, const float *B, int K, int reversed, int *mismatches, vfr_stream_t stream)
{
    VFR_REQUIRE(A && B && mismatches && K > 0 && (K % 4) == 0, VFR_EINVAL, "vfr_mfma_selfcheck: bad argument (K a multiple of 4)");
    hipStream_t st = vfr::as_stream(stream);
    if (hipMemsetAsync(mismatches, 0, sizeof(int), st) != hipSuccess) return vfr::fail(VFR_EHIP, "vfr_mfma_selfcheck: hipMemsetAsync failed");
    hipLaunchKernelGGL(vfr::mfma_selfcheck_kernel, dim3(1), dim3(64), 0, st, A, B, K, reversed, mismatches);
    VFR_CHECK_LAUNCH("mfma_selfcheck_kernel");
    return VFR_OK;
}
