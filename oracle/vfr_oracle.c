/*
 * vfr_oracle.c -- CPU restatement of the reference's cross-modal scoring hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker.  The product path is libvfr.so (HIP, gfx950) and fails loudly without it.
 *
 * Parity status: PINNED.  Every function here is checked against golden vectors produced by
 * importing the reference itself (tools/gen_golden.py -> tests/golden/ *.npz): embeddings and
 * distances within 1e-4 (observed <= 2e-6), top-k moment indices equal wherever the reference's
 * own fp32 gap between neighbours exceeds that tolerance, metric dicts equal.
 *
 * Why a restatement and not a transliteration: the reference's arithmetic executes inside torch /
 * numpy vendor kernels whose fp32 summation order is unspecified (SIMD width dependent), so "the
 * reference's bits" are not reproducible even by the reference on another CPU (hard
 * part 1).  This file therefore fixes ONE canonical fp32 evaluation order per operation -- the
 * order the gfx950 kernels use -- so that HIP == oracle bit-for-bit, and oracle == reference to
 * 1e-4.  The canonical rules, used everywhere below:
 *
 *   R1  every dot product / contraction is a single k-ascending chain acc = fmaf(a_k, b_k, acc)
 *       (this is exactly what the gfx950 fp32 MFMAs -- 32x32x2, 16x16x4 -- compute; no split-K, no pairwise tree);
 *   R2  biases are added after the chain, unless a chain is documented to start from a C-in;
 *   R3  row reductions that must be parallel on the GPU (sum of squares over 4096 features) use
 *       the fixed "64 lanes x strided float4 chunks + xor butterfly 32..1" tree (tree64 below);
 *   R4  no FMA contraction by the compiler (-ffp-contract=off); every fused op is an explicit fmaf;
 *   R5  exp/sigmoid/tanh are the polynomial forms below (IEEE ops only), sqrt and / are IEEE.
 *
 * Reference citations are relative to /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VFO_EXPORT __attribute__((visibility("default")))

static int g_threads = 1;

VFO_EXPORT int vfo_version(void) { return 1; }
VFO_EXPORT void vfo_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
VFO_EXPORT int vfo_get_threads(void) { return g_threads; }

/* ------------------------------------------------------------------------------------------ */
/* R5: canonical transcendental forms (torch uses Sleef/libm: <= 1e-7 abs apart from these)   */
/* ------------------------------------------------------------------------------------------ */
static inline float vfo_expf(float x)
{
    /* clamp keeps 2^n and the result normal: no denormals anywhere on either side */
    x = fminf(fmaxf(x, -80.0f), 80.0f);
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)((int32_t)n + 127) << 23;
    return y * s.f;
}
static inline float vfo_sigmoidf(float x) { return 1.0f / (1.0f + vfo_expf(-x)); }
static inline float vfo_tanhf(float x)
{
    float ax = fabsf(x);
    float e = vfo_expf(2.0f * ax);
    float t = 1.0f - 2.0f / (e + 1.0f);
    return copysignf(t, x);
}

/* elementwise probe so tests can pin the GPU's math against this file bit-for-bit */
VFO_EXPORT void vfo_math_f32(int op, const float *x, const float *y, float *out, long n)
{
    for (long i = 0; i < n; ++i) {
        switch (op) {
        case 0: out[i] = vfo_expf(x[i]); break;
        case 1: out[i] = vfo_sigmoidf(x[i]); break;
        case 2: out[i] = vfo_tanhf(x[i]); break;
        case 3: out[i] = x[i] / y[i]; break;
        case 4: out[i] = sqrtf(x[i]); break;
        case 5: out[i] = fmaf(x[i], y[i], x[i]); break;
        default: out[i] = 0.0f;
        }
    }
}

/* R3: fixed reduction tree for sum of squares over a long row: lane l of 64 owns the float4 chunks
 * l, l+64, l+128, ... (elements 4*chunk .. 4*chunk+3 in order), then an xor butterfly 32..1. */
static float tree64_sumsq(const float *x, int n)
{
    float p[64];
    for (int l = 0; l < 64; ++l) {
        float acc = 0.0f;
        for (long chunk = l; chunk * 4 < n; chunk += 64)
            for (int c = 0; c < 4; ++c) {
                long j = chunk * 4 + c;
                if (j < n) acc = fmaf(x[j], x[j], acc);
            }
        p[l] = acc;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        float q[64];
        for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
        memcpy(p, q, sizeof p);
    }
    return p[0];
}

/* ------------------------------------------------------------------------------------------ */
/* R1: chained "NT" GEMM  out[m][j] = Cin[m][j] (or 0) ; acc = fmaf(A[m][k], W[j][k], acc)     */
/* W is given transposed (Wt[k][j]) so the j loop vectorises; each acc[j] stays a k-chain.      */
/* ------------------------------------------------------------------------------------------ */
static float *transpose_nk(const float *W, int N, int K)
{
    float *Wt = (float *)malloc((size_t)N * K * sizeof(float));
    for (int j = 0; j < N; ++j)
        for (int k = 0; k < K; ++k) Wt[(size_t)k * N + j] = W[(size_t)j * K + k];
    return Wt;
}

static void chain_gemm(const float *A, long lda, long M, int K, const float *restrict Wt, int N,
                       const float *Cin, long ldc, float *out, long ldo)
{
    enum { MB = 8 };
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (long m0 = 0; m0 < M; m0 += MB) {
        int mb = (int)((M - m0) < MB ? (M - m0) : MB);
        float *acc = (float *)malloc((size_t)MB * N * sizeof(float));
        for (int r = 0; r < mb; ++r)
            for (int j = 0; j < N; ++j) acc[r * N + j] = Cin ? Cin[(m0 + r) * ldc + j] : 0.0f;
        for (int k = 0; k < K; ++k) {
            const float *restrict w = Wt + (size_t)k * N;
            for (int r = 0; r < mb; ++r) {
                float a = A[(m0 + r) * lda + k];
                float *restrict ac = acc + r * N;
                for (int j = 0; j < N; ++j) ac[j] = fmaf(a, w[j], ac[j]);
            }
        }
        for (int r = 0; r < mb; ++r) memcpy(out + (m0 + r) * ldo, acc + r * N, (size_t)N * sizeof(float));
        free(acc);
    }
}

/* generic Linear: chain + bias (+ReLU).  nn.Linear call sites: model/models.py:22-24,31,47;
 * torchvision VGG classifier[0], classifier[3] (get_rgb_features.py:126). */
VFO_EXPORT void vfo_linear(const float *A, long M, int K, const float *W, const float *b, int N, int relu, float *out)
{
    float *Wt = transpose_nk(W, N, K);
    chain_gemm(A, K, M, K, Wt, N, NULL, 0, out, N);
    free(Wt);
    for (long m = 0; m < M; ++m)
        for (int j = 0; j < N; ++j) {
            float v = out[m * N + j] + (b ? b[j] : 0.0f);
            out[m * N + j] = (relu && !(v > 0.0f)) ? 0.0f : v;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* a15  utils.generate_moments (model/utils.py:71-75): singles (j,j) first, then combinations   */
/* ------------------------------------------------------------------------------------------ */
VFO_EXPORT int vfo_num_moments(int n) { return n * (n + 1) / 2; }
VFO_EXPORT void vfo_generate_moments(int n, int32_t *se)
{
    int m = 0;
    for (int j = 0; j < n; ++j) { se[2 * m] = j; se[2 * m + 1] = j; ++m; }
    for (int s = 0; s < n; ++s)
        for (int e = s + 1; e < n; ++e) { se[2 * m] = s; se[2 * m + 1] = e; ++m; }
}
/* local moment index of span (s,e) in the order above */
static inline int moment_index(int n, int s, int e)
{
    if (s == e) return s;
    return n + (s * (2 * n - s - 1)) / 2 + (e - s - 1);
}

/* ------------------------------------------------------------------------------------------ */
/* a3  CustomDataset.load_video_features, non-prep branch (model/data.py:163-181)               */
/* frames [T,F] -> seg [ceil(T/seg_len), F], ctx [F];  mode 0 = avg (np.mean), 1 = max          */
/* mean = (f0 + f1 + ... ) / count in fp32 frame order; norm = sqrt(tree64 sum of squares);     */
/* out = pooled / (norm + 1e-5)                                                                 */
/* ------------------------------------------------------------------------------------------ */
static void pool_rows(const float *frames, int t0, int t1, int F, int mode, float *out)
{
    for (int j = 0; j < F; ++j) {
        float acc = frames[(size_t)t0 * F + j];
        for (int t = t0 + 1; t < t1; ++t) {
            float v = frames[(size_t)t * F + j];
            acc = mode ? fmaxf(acc, v) : acc + v;
        }
        out[j] = mode ? acc : acc / (float)(t1 - t0);
    }
    float nrm = sqrtf(tree64_sumsq(out, F)) + 1e-5f;
    for (int j = 0; j < F; ++j) out[j] = out[j] / nrm;
}
VFO_EXPORT int vfo_segment_pool_norm(const float *frames, int T, int F, int seg_len, int mode, float *seg, float *ctx)
{
    int nseg = (T + seg_len - 1) / seg_len;
    for (int i = 0; i < nseg; ++i) {
        int t0 = i * seg_len, t1 = t0 + seg_len < T ? t0 + seg_len : T;
        pool_rows(frames, t0, t1, F, mode, seg + (size_t)i * F);
    }
    pool_rows(frames, 0, T, F, mode, ctx);
    return nseg;
}

/* ------------------------------------------------------------------------------------------ */
/* a4 + a7  make_visual_features (model/data.py:204-213) + CALModel visual branch               */
/* (model/models.py:21-26,55-56).  Row t of video v is [seg_t | ctx_v | t/n | (t+1)/n]; the     */
/* 8194-wide concat is never formed.  Canonical factored order (R1, R2):                        */
/*   S  = chain_{k<F} seg[k]*W1[j][k]            (from 0)                                       */
/*   Cx = chain_{k<F} ctx[k]*W1[j][F+k]          (from 0, once per video)                       */
/*   Te = fmaf(te1, W1[j][2F+1], fmaf(te0, W1[j][2F], 0))                                       */
/*   h  = relu(((S + Cx) + Te) + b1[j]);   out[d] = chain_{j<hid} h[j]*W2[d][j] + b2[d]         */
/* ------------------------------------------------------------------------------------------ */
VFO_EXPORT void vfo_visual_mlp(const float *seg, const float *ctx, const int32_t *clip_off, int Nv, int F,
                               const float *W1, const float *b1, const float *W2, const float *b2,
                               int hid, int D, float *out)
{
    long C = clip_off[Nv];
    int K1 = 2 * F + 2;
    float *Wseg_t = (float *)malloc((size_t)F * hid * sizeof(float));
    float *Wctx_t = (float *)malloc((size_t)F * hid * sizeof(float));
    for (int j = 0; j < hid; ++j)
        for (int k = 0; k < F; ++k) {
            Wseg_t[(size_t)k * hid + j] = W1[(size_t)j * K1 + k];
            Wctx_t[(size_t)k * hid + j] = W1[(size_t)j * K1 + F + k];
        }
    float *S = (float *)malloc((size_t)C * hid * sizeof(float));
    float *Cx = (float *)malloc((size_t)Nv * hid * sizeof(float));
    chain_gemm(seg, F, C, F, Wseg_t, hid, NULL, 0, S, hid);
    chain_gemm(ctx, F, Nv, F, Wctx_t, hid, NULL, 0, Cx, hid);
    free(Wseg_t); free(Wctx_t);
    for (int v = 0; v < Nv; ++v) {
        int c0 = clip_off[v], n = clip_off[v + 1] - c0;
        for (int t = 0; t < n; ++t) {
            float te0 = (float)t / (float)n, te1 = (float)(t + 1) / (float)n;
            float *h = S + (size_t)(c0 + t) * hid;
            for (int j = 0; j < hid; ++j) {
                float te = fmaf(te1, W1[(size_t)j * K1 + 2 * F + 1], fmaf(te0, W1[(size_t)j * K1 + 2 * F], 0.0f));
                float x = ((h[j] + Cx[(size_t)v * hid + j]) + te) + b1[j];
                h[j] = x > 0.0f ? x : 0.0f;
            }
        }
    }
    float *W2t = transpose_nk(W2, D, hid);
    chain_gemm(S, hid, C, hid, W2t, D, NULL, 0, out, D);
    free(W2t);
    for (long r = 0; r < C; ++r)
        for (int d = 0; d < D; ++d) out[r * D + d] = out[r * D + d] + b2[d];
    free(S); free(Cx);
}

/* ------------------------------------------------------------------------------------------ */
/* a8  CALModel GloVe branch (model/models.py:61-66): Embedding -> [optional unit-norm x        */
/* learnable length, :62-64] -> BiLSTM(H), h0=c0=0, ALL T steps incl. pads (Q6) -> Linear.      */
/* torch.nn.LSTM gate order i,f,g,o.  Canonical order (one chain over the concatenated          */
/* [x_t | h] operand, the form the fused MFMA step kernel computes):                            */
/*   acc     = chain_{k<E} x_t[k]*Wih[j][k]   (from 0)   then continues                          */
/*             chain_{k<H} h[k]*Whh[j][k]                                                       */
/*   gate[j] = acc + (bih[j] + bhh[j])                                                          */
/*   c' = fmaf(f, c, i*g);  h' = o * tanh(c')                                                   */
/*   out[d]  = chain_{k<2H} [h_fwd|h_bwd][k]*Wfc[d][k] + bfc[d]                                 */
/* ------------------------------------------------------------------------------------------ */
static void lstm_dir(const float *X /*[B,T,E]*/, long B, int T, int E, int H, const float *Wih, const float *Whh,
                     const float *bih, const float *bhh, int reverse, float *hout /*[B, ld]*/, long ld)
{
    int G = 4 * H, K = E + H;
    float *Wcat = (float *)malloc((size_t)G * K * sizeof(float));     /* rows [Wih[j] | Whh[j]] */
    for (int j = 0; j < G; ++j) {
        memcpy(Wcat + (size_t)j * K, Wih + (size_t)j * E, (size_t)E * sizeof(float));
        memcpy(Wcat + (size_t)j * K + E, Whh + (size_t)j * H, (size_t)H * sizeof(float));
    }
    float *Wt = transpose_nk(Wcat, G, K);
    free(Wcat);
    float *A = (float *)malloc((size_t)B * K * sizeof(float));
    float *h = (float *)calloc((size_t)B * H, sizeof(float));
    float *c = (float *)calloc((size_t)B * H, sizeof(float));
    float *gates = (float *)malloc((size_t)B * G * sizeof(float));
    for (int step = 0; step < T; ++step) {
        int t = reverse ? T - 1 - step : step;
        for (long b = 0; b < B; ++b) {
            memcpy(A + b * K, X + ((size_t)b * T + t) * E, (size_t)E * sizeof(float));
            memcpy(A + b * K + E, h + b * H, (size_t)H * sizeof(float));
        }
        chain_gemm(A, K, B, K, Wt, G, NULL, 0, gates, G);
        for (long b = 0; b < B; ++b)
            for (int j = 0; j < H; ++j) {
                const float *g4 = gates + b * G;
                float ig = vfo_sigmoidf(g4[j] + (bih[j] + bhh[j]));
                float fg = vfo_sigmoidf(g4[H + j] + (bih[H + j] + bhh[H + j]));
                float gg = vfo_tanhf(g4[2 * H + j] + (bih[2 * H + j] + bhh[2 * H + j]));
                float og = vfo_sigmoidf(g4[3 * H + j] + (bih[3 * H + j] + bhh[3 * H + j]));
                float cn = fmaf(fg, c[b * H + j], ig * gg);
                c[b * H + j] = cn;
                h[b * H + j] = og * vfo_tanhf(cn);
            }
    }
    for (long b = 0; b < B; ++b) memcpy(hout + b * ld, h + b * H, (size_t)H * sizeof(float));
    free(Wt); free(A); free(h); free(c); free(gates);
}

VFO_EXPORT void vfo_embed(const int64_t *tokens, long B, int T, const float *emb, const float *len_tab, int E, float *X)
{
    for (long r = 0; r < B * T; ++r) {
        const float *e = emb + (size_t)tokens[r] * E;
        float *x = X + r * E;
        if (!len_tab) { memcpy(x, e, (size_t)E * sizeof(float)); continue; }
        /* models.py:64: embedded / (||embedded|| + 1e-5) * length ; norm = sqrt(k-chain sum sq) */
        float acc = 0.0f;
        for (int k = 0; k < E; ++k) acc = fmaf(e[k], e[k], acc);
        float nrm = sqrtf(acc) + 1e-5f, len = len_tab[tokens[r]];
        for (int k = 0; k < E; ++k) x[k] = (e[k] / nrm) * len;
    }
}

VFO_EXPORT void vfo_bilstm_final(const int64_t *tokens, long B, int T, const float *emb, const float *len_tab,
                                 const float *Wih_f, const float *Whh_f, const float *bih_f, const float *bhh_f,
                                 const float *Wih_b, const float *Whh_b, const float *bih_b, const float *bhh_b,
                                 int E, int H, const float *Wfc, const float *bfc, int D, float *out)
{
    float *X = (float *)malloc((size_t)B * T * E * sizeof(float));
    vfo_embed(tokens, B, T, emb, len_tab, E, X);
    float *hcat = (float *)malloc((size_t)B * 2 * H * sizeof(float));
    lstm_dir(X, B, T, E, H, Wih_f, Whh_f, bih_f, bhh_f, 0, hcat, 2L * H);
    lstm_dir(X, B, T, E, H, Wih_b, Whh_b, bih_b, bhh_b, 1, hcat + H, 2L * H);
    vfo_linear(hcat, B, 2 * H, Wfc, bfc, D, 0, out);
    free(X); free(hcat);
}

/* ------------------------------------------------------------------------------------------ */
/* a10  scoring core (model/evaluate.py:49-58, evaluate_single.py:48-53, main.py:148-157)       */
/*   dist[c] = F.pairwise_distance(v_c, q) = || (v_c - q) + eps ||_2  (eps on the DIFFERENCE,   */
/*             Q8):  d_k = (v[k] - q[k]) + eps;  acc = fmaf(d_k, d_k, acc) k-ascending; sqrtf   */
/*   score(s,e) = (dist[s] + dist[s+1] + ... + dist[e]) / (e - s + 1)   (index_select().mean()) */
/* ------------------------------------------------------------------------------------------ */
static inline float clip_dist(const float *v, const float *q, int D, float eps)
{
    float acc = 0.0f;
    for (int k = 0; k < D; ++k) {
        float d = (v[k] - q[k]) + eps;
        acc = fmaf(d, d, acc);
    }
    return sqrtf(acc);
}
VFO_EXPORT void vfo_clip_distances(const float *Q, long Nq, const float *V, long C, int D, float eps, float *dist)
{
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (long q = 0; q < Nq; ++q)
        for (long c = 0; c < C; ++c) dist[q * C + c] = clip_dist(V + c * D, Q + q * D, D, eps);
}
/* ------------------------------------------------------------------------------------------ */
/* f2  Trainer.ranking_loss (model/main.py:214-232), the loss of train_epoch / test_epoch:       */
/*   per sample i:  c_x = mean over the rows r of set x with mask[r] == i of                    */
/*                        F.pairwise_distance(x[r], lang[i])   (x = posit | intra | inter)       */
/*   loss = sum_i relu(c_posit - c_intra + b) + lamb * relu(c_posit - c_inter + b)               */
/* canonical order: clip_dist chain per row, sequential sum in row order, IEEE divide by the     */
/* count (0/0 = NaN like torch's mean of an empty selection), hinge terms, sequential sum over i.*/
/* per_sample [S,8] = c_posit, c_intra, c_inter, t1, t2, n_posit, n_intra, loss_i                 */
/* ------------------------------------------------------------------------------------------ */
VFO_EXPORT float vfo_ranking_loss(const float *posit, const float *intra, const float *inter, const float *lang,
                                  const int32_t *maskp, const int32_t *maskn, long P, long Nn, int S, int D, float b,
                                  float lamb, float eps, float *per_sample)
{
    float loss = 0.0f;
    for (int i = 0; i < S; ++i) {
        float sp = 0.0f, sn = 0.0f, si = 0.0f;
        long np_ = 0, nn_ = 0;
        for (long r = 0; r < P; ++r)
            if (maskp[r] == i) {
                sp = sp + clip_dist(posit + r * D, lang + (long)i * D, D, eps);
                si = si + clip_dist(inter + r * D, lang + (long)i * D, D, eps);
                ++np_;
            }
        for (long r = 0; r < Nn; ++r)
            if (maskn[r] == i) { sn = sn + clip_dist(intra + r * D, lang + (long)i * D, D, eps); ++nn_; }
        const float cp = sp / (float)np_, cn = sn / (float)nn_, ci = si / (float)np_;
        const float t1 = (cp - cn) + b, t2 = (cp - ci) + b;
        const float h1 = t1 > 0.0f ? t1 : (t1 != t1 ? t1 : 0.0f), h2 = t2 > 0.0f ? t2 : (t2 != t2 ? t2 : 0.0f);
        const float li = h1 + lamb * h2;
        loss = loss + li;
        if (per_sample) {
            float *o = per_sample + (long)i * 8;
            o[0] = cp; o[1] = cn; o[2] = ci; o[3] = t1; o[4] = t2; o[5] = (float)np_; o[6] = (float)nn_; o[7] = li;
        }
    }
    return loss;
}
/* all moments of one video for one query, written in generate_moments order */
static inline void video_moments(const float *dist, int n, float *out)
{
    for (int s = 0; s < n; ++s) {
        float sum = dist[s];
        out[s] = sum / 1.0f;
        for (int e = s + 1; e < n; ++e) {
            sum = sum + dist[e];
            out[moment_index(n, s, e)] = sum / (float)(e - s + 1);
        }
    }
}
VFO_EXPORT void vfo_moment_offsets(const int32_t *clip_off, int Nv, int64_t *mom_off)
{
    mom_off[0] = 0;
    for (int v = 0; v < Nv; ++v) {
        int n = clip_off[v + 1] - clip_off[v];
        mom_off[v + 1] = mom_off[v] + (int64_t)n * (n + 1) / 2;
    }
}
VFO_EXPORT void vfo_score_moments(const float *Q, long Nq, const float *V, const int32_t *clip_off, int Nv, int D,
                                  float eps, float *scores)
{
    int64_t *mo = (int64_t *)malloc((size_t)(Nv + 1) * sizeof(int64_t));
    vfo_moment_offsets(clip_off, Nv, mo);
    int64_t SM = mo[Nv];
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (long q = 0; q < Nq; ++q) {
        float dist[1024];
        for (int v = 0; v < Nv; ++v) {
            int c0 = clip_off[v], n = clip_off[v + 1] - c0;
            for (int c = 0; c < n; ++c) dist[c] = clip_dist(V + (size_t)(c0 + c) * D, Q + q * D, D, eps);
            video_moments(dist, n, scores + q * SM + mo[v]);
        }
    }
    free(mo);
}
/* each query against ONE video (its own): evaluate_single.py:48-53.  scores [Nq, Mmax], pad +inf */
VFO_EXPORT void vfo_score_own(const float *Q, long Nq, const float *V, const int32_t *clip_off, const int32_t *own,
                              int D, float eps, int Mmax, float *scores)
{
    for (long q = 0; q < Nq; ++q) {
        float dist[1024];
        int v = own[q], c0 = clip_off[v], n = clip_off[v + 1] - c0;
        for (int m = 0; m < Mmax; ++m) scores[q * Mmax + m] = INFINITY;
        for (int c = 0; c < n; ++c) dist[c] = clip_dist(V + (size_t)(c0 + c) * D, Q + q * D, D, eps);
        video_moments(dist, n, scores + q * Mmax);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a12  global ranking (model/evaluate.py:67-80).  np.argsort's order on exact fp32 ties is     */
/* unspecified (unstable introsort); canonical order here is (distance, global moment id)       */
/* ascending, i.e. the stable argsort.  Distances are >= 0 so the fp32 bit pattern is           */
/* order-preserving and key = dist_bits << 32 | id compares lexicographically.                  */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t make_key(float d, uint32_t id)
{
    union { float f; uint32_t u; } c; c.f = d;
    return ((uint64_t)c.u << 32) | id;
}
static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}
VFO_EXPORT void vfo_score_topk(const float *Q, long Nq, const float *V, const int32_t *clip_off, int Nv, int D,
                               float eps, int k, float *out_dist, int64_t *out_idx)
{
    int64_t *mo = (int64_t *)malloc((size_t)(Nv + 1) * sizeof(int64_t));
    vfo_moment_offsets(clip_off, Nv, mo);
    int64_t SM = mo[Nv];
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (long q = 0; q < Nq; ++q) {
        uint64_t *keys = (uint64_t *)malloc((size_t)SM * sizeof(uint64_t));
        float dist[1024], sc[1024 * 8];
        for (int v = 0; v < Nv; ++v) {
            int c0 = clip_off[v], n = clip_off[v + 1] - c0, M = n * (n + 1) / 2;
            for (int c = 0; c < n; ++c) dist[c] = clip_dist(V + (size_t)(c0 + c) * D, Q + q * D, D, eps);
            float *buf = M <= 1024 * 8 ? sc : (float *)malloc((size_t)M * sizeof(float));
            video_moments(dist, n, buf);
            for (int m = 0; m < M; ++m) keys[mo[v] + m] = make_key(buf[m], (uint32_t)(mo[v] + m));
            if (buf != sc) free(buf);
        }
        qsort(keys, (size_t)SM, sizeof(uint64_t), cmp_u64);
        for (int i = 0; i < k; ++i) {
            if (i < SM) {
                union { uint32_t u; float f; } c; c.u = (uint32_t)(keys[i] >> 32);
                out_dist[q * k + i] = c.f;
                out_idx[q * k + i] = (int64_t)(keys[i] & 0xffffffffu);
            } else {
                out_dist[q * k + i] = INFINITY;
                out_idx[q * k + i] = -1;
            }
        }
        free(keys);
    }
    free(mo);
}
/* rank of a given (distance, id) in the canonical order = number of moments sorting before it.
 * This is evaluate.py:77's MR (0-based index of the first GT-positive moment) when (dstar,istar)
 * is the best GT-positive moment. */
VFO_EXPORT void vfo_rank_of(const float *Q, long Nq, const float *V, const int32_t *clip_off, int Nv, int D,
                            float eps, const float *dstar, const int64_t *istar, int64_t *count_lt)
{
    int64_t *mo = (int64_t *)malloc((size_t)(Nv + 1) * sizeof(int64_t));
    vfo_moment_offsets(clip_off, Nv, mo);
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (long q = 0; q < Nq; ++q) {
        float dist[1024];
        uint64_t kstar = make_key(dstar[q], (uint32_t)istar[q]);
        int64_t cnt = 0;
        for (int v = 0; v < Nv; ++v) {
            int c0 = clip_off[v], n = clip_off[v + 1] - c0;
            for (int c = 0; c < n; ++c) dist[c] = clip_dist(V + (size_t)(c0 + c) * D, Q + q * D, D, eps);
            for (int s = 0; s < n; ++s) {
                float sum = dist[s];
                for (int e = s; e < n; ++e) {
                    if (e > s) sum = sum + dist[e];
                    float sc = sum / (float)(e - s + 1);
                    cnt += make_key(sc, (uint32_t)(mo[v] + moment_index(n, s, e))) < kstar;
                }
            }
        }
        count_lt[q] = cnt;
    }
    free(mo);
}

/* ------------------------------------------------------------------------------------------ */
/* a1  DiDeMoDataset normalise tail (get_rgb_features.py:64-69):                                */
/*     THWC u8 -> TCHW f32,  ((x / 255) - mean[c]) / std[c]   (three fp32 roundings)            */
/* ------------------------------------------------------------------------------------------ */
static const float IMNET_MEAN[3] = {0.485f, 0.456f, 0.406f};
static const float IMNET_STD[3] = {0.229f, 0.224f, 0.225f};
VFO_EXPORT void vfo_frames_normalize(const uint8_t *thwc, int T, int H, int W, float *tchw)
{
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < 3; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float v = (float)thwc[(((size_t)t * H + y) * W + x) * 3 + c];
                    v = v / 255.0f;
                    v = v - IMNET_MEAN[c];
                    tchw[(((size_t)t * 3 + c) * H + y) * W + x] = v / IMNET_STD[c];
                }
}

/* ------------------------------------------------------------------------------------------ */
/* a2  VGG19 "E" feature stack up to fc7 (get_rgb_features.py:122-126; torchvision vgg19,       */
/* version unpinned, absent here: arithmetic restated from the public configuration and pinned  */
/* against torch.nn.functional in tests/test_oracle_vgg_torch.py).                              */
/*   conv3x3 pad 1:  one chain per output over k = (ky*3 + kx)*Cin + ci ascending -- tap-major, */
/*   channel-minor, the order an NHWC implicit GEMM consumes; zero padding contributes           */
/*   fmaf(0,w,acc) = acc; + bias; ReLU.  Weights keep the reference layout [Cout,Cin,3,3].      */
/* ------------------------------------------------------------------------------------------ */
VFO_EXPORT void vfo_conv3x3_relu(const float *x, int B, int Cin, int H, int W, const float *w, const float *b,
                                 int Cout, float *y)
{
#pragma omp parallel for collapse(2) schedule(static) num_threads(g_threads)
    for (int n = 0; n < B; ++n)
        for (int co = 0; co < Cout; ++co) {
            float *acc = (float *)calloc((size_t)H * W, sizeof(float));
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx)
                    for (int ci = 0; ci < Cin; ++ci) {
                        float wv = w[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx];
                        const float *xp = x + ((size_t)n * Cin + ci) * H * W;
                        for (int oy = 0; oy < H; ++oy) {
                            int iy = oy + ky - 1;
                            if (iy < 0 || iy >= H) continue;
                            for (int ox = 0; ox < W; ++ox) {
                                int ix = ox + kx - 1;
                                if (ix < 0 || ix >= W) continue;
                                acc[oy * W + ox] = fmaf(xp[iy * W + ix], wv, acc[oy * W + ox]);
                            }
                        }
                    }
            float *yp = y + ((size_t)n * Cout + co) * H * W;
            for (int i = 0; i < H * W; ++i) {
                float v = acc[i] + b[co];
                yp[i] = v > 0.0f ? v : 0.0f;
            }
            free(acc);
        }
}
VFO_EXPORT void vfo_maxpool2(const float *x, int B, int C, int H, int W, float *y)
{
    int Ho = H / 2, Wo = W / 2;
    for (size_t p = 0; p < (size_t)B * C; ++p)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const float *xp = x + p * H * W + (size_t)(2 * oy) * W + 2 * ox;
                y[p * Ho * Wo + (size_t)oy * Wo + ox] = fmaxf(fmaxf(xp[0], xp[1]), fmaxf(xp[W], xp[W + 1]));
            }
}
/* AdaptiveAvgPool2d((7,7)): bin i covers [floor(i*H/7), ceil((i+1)*H/7)); row-major sum / count.
 * Identity when H = W = 7 (224x224 input). */
VFO_EXPORT void vfo_adaptive_avgpool7(const float *x, int B, int C, int H, int W, float *y)
{
    for (size_t p = 0; p < (size_t)B * C; ++p)
        for (int oy = 0; oy < 7; ++oy)
            for (int ox = 0; ox < 7; ++ox) {
                int y0 = (oy * H) / 7, y1 = ((oy + 1) * H + 6) / 7;
                int x0 = (ox * W) / 7, x1 = ((ox + 1) * W + 6) / 7;
                float acc = 0.0f;
                for (int iy = y0; iy < y1; ++iy)
                    for (int ix = x0; ix < x1; ++ix) acc = acc + x[p * H * W + (size_t)iy * W + ix];
                y[p * 49 + oy * 7 + ox] = acc / (float)((y1 - y0) * (x1 - x0));
            }
}

/* ------------------------------------------------------------------------------------------ */
/* f4  ResNet-152 up to the global average pool (get_rgb_features.py:127-131: torchvision      */
/* resnet152, children()[:-1] = conv1, bn1, relu, maxpool, layer1..4, avgpool -> [T, 2048, 1, 1];*/
/* torchvision is absent here and its version is unpinned: arithmetic restated from the public  */
/* architecture -- Bottleneck blocks (1x1, 3x3 carrying the stride, 1x1 x4; 1x1 downsample on   */
/* the first block of a layer), counts 3/8/36/3 -- and pinned against a torch.nn composition    */
/* (tests/golden G12).                                                                          */
/*   eval-mode BatchNorm folded into the convolution: invstd = 1 / sqrt(var + eps),             */
/*   alpha = gamma * invstd, w' = w * alpha (per output channel), beta = fma(-mean, alpha, b);  */
/*   conv: one chain per output over k = (ky*KW + kx)*Cin + ci ascending (tap-major,            */
/*   channel-minor -- the order an NHWC implicit GEMM consumes), padding contributes nothing;   */
/*   then v = acc + beta, v = v + identity (when the block adds its input), ReLU.               */
/* ------------------------------------------------------------------------------------------ */
VFO_EXPORT void vfo_bn_fold(const float *w, const float *bn /* [4][Cout]: gamma, b, mean, var */, int Cout, int K, float eps,
                            float *wf, float *beta)
{
    for (int co = 0; co < Cout; ++co) {
        const float invstd = 1.0f / sqrtf(bn[3 * Cout + co] + eps);
        const float alpha = bn[co] * invstd;
        for (int k = 0; k < K; ++k) wf[(size_t)co * K + k] = w[(size_t)co * K + k] * alpha;
        beta[co] = fmaf(-bn[2 * Cout + co], alpha, bn[Cout + co]);
    }
}
VFO_EXPORT void vfo_conv2d(const float *x, int B, int Cin, int H, int W, const float *wf /* [Cout,Cin,KH,KW] folded */,
                           const float *beta, int Cout, int KH, int KW, int stride, int pad, const float *res, int relu, float *y)
{
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static) num_threads(g_threads)
    for (int n = 0; n < B; ++n)
        for (int co = 0; co < Cout; ++co) {
            float *acc = (float *)calloc((size_t)Ho * Wo, sizeof(float));
            for (int ky = 0; ky < KH; ++ky)
                for (int kx = 0; kx < KW; ++kx)
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float wv = wf[(((size_t)co * Cin + ci) * KH + ky) * KW + kx];
                        const float *xp = x + ((size_t)n * Cin + ci) * H * W;
                        for (int oy = 0; oy < Ho; ++oy) {
                            const int iy = oy * stride + ky - pad;
                            if (iy < 0 || iy >= H) continue;
                            for (int ox = 0; ox < Wo; ++ox) {
                                const int ix = ox * stride + kx - pad;
                                if (ix < 0 || ix >= W) continue;
                                acc[oy * Wo + ox] = fmaf(xp[iy * W + ix], wv, acc[oy * Wo + ox]);
                            }
                        }
                    }
            float *yp = y + ((size_t)n * Cout + co) * Ho * Wo;
            const float *rp = res ? res + ((size_t)n * Cout + co) * Ho * Wo : NULL;
            for (int i = 0; i < Ho * Wo; ++i) {
                float v = acc[i] + beta[co];
                if (rp) v = v + rp[i];
                yp[i] = (relu && !(v > 0.0f)) ? 0.0f : v;
            }
            free(acc);
        }
}
/* MaxPool2d(kernel 3, stride 2, padding 1): the padding never wins (torch pads with -inf) */
VFO_EXPORT void vfo_maxpool3s2(const float *x, int B, int C, int H, int W, float *y)
{
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    for (size_t p = 0; p < (size_t)B * C; ++p)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                float m = -INFINITY;
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
                        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                        m = fmaxf(m, x[p * H * W + (size_t)iy * W + ix]);
                    }
                y[p * Ho * Wo + (size_t)oy * Wo + ox] = m;
            }
}
/* AdaptiveAvgPool2d((1,1)): row-major sum of the plane / (H*W) */
VFO_EXPORT void vfo_global_avgpool(const float *x, int B, int C, int H, int W, float *y)
{
    for (size_t p = 0; p < (size_t)B * C; ++p) {
        float acc = 0.0f;
        for (int i = 0; i < H * W; ++i) acc = acc + x[p * H * W + i];
        y[p] = acc / (float)(H * W);
    }
}
