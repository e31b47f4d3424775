/*
 * vfr.h -- C ABI of libvfr.so: the MI355X (gfx950) cross-modal scoring hot path of
 * video-fragments-retrieval.  Plain pointers and sizes only; no torch / C++ types.
 *
 * The reference has no FFI layer: its operator surface is Python (nn.Module calls and two
 * evaluate() functions) and all arithmetic runs in implicit torch / torchvision / numpy vendor
 * kernels.  Each entry point below replaces one such implicit call site; the citation names the
 * reference file:line (relative to the reference repo root) whose arithmetic it takes over.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; row-major, fp32 unless typed
 *   - the caller owns all memory, workspaces included (vfr_*_workspace_bytes tells how much);
 *     the library never allocates or frees device memory and keeps no pointer after return
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     no entry point synchronises with the host
 *   - return 0 on success, a negative VFR_E* code otherwise; vfr_last_error() gives the message
 *     (thread-local).  No exception, exit or abort crosses the boundary.
 *   - no model state: weights are passed per call (no hidden model handle), so load_state_dict()/.to()
 *     on the Python side keep working.  Compute entry points are re-entrant: any number of host threads
 *     may call them concurrently (each on its own stream or sharing one).  The process-wide state is:
 *     the tuning options (vfr_set_option: PROCESS-GLOBAL atomic ints, read at the start of a call -- a
 *     knob for tests and sweeps, every setting gives the same bits; do NOT change one while calls are in
 *     flight on other threads: a call may then see the old value for one of its launches and the new
 *     one for the next, which is harmless for the results but not a per-call setting), the fault word
 *     registered with vfr_set_fault_word, and the launch-site profiler (off by default; event pairs
 *     are owned by the calling scope, the totals are folded under a mutex)
 *   - numerics: every contraction is one k-ascending fp32 fma chain (what the fp32 MFMAs
 *     compute), so results are bit-identical to oracle/vfr_oracle.c on any input
 */
#ifndef VFR_H
#define VFR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library itself is built with -fvisibility=hidden */
#endif

#define VFR_OK 0
#define VFR_EINVAL (-1)       /* null pointer, negative size, misaligned buffer                 */
#define VFR_EUNSUPPORTED (-2) /* shape outside what the kernels are built for (message says)    */
#define VFR_EHIP (-3)         /* a HIP runtime call failed; hipGetErrorString in the message    */
#define VFR_EWORKSPACE (-4)   /* workspace smaller than vfr_*_workspace_bytes                   */

typedef void *vfr_stream_t; /* hipStream_t */

int vfr_version(void);
const char *vfr_last_error(void);

/* ---- a3  clip pooling: CustomDataset.load_video_features, model/data.py:163-181 -------------
 * frames [T,F] -> seg [ceil(T/seg_len), F], ctx [F].  mode 0 = avg (np.mean), 1 = max (np.max);
 * pooled row / (||pooled row||_2 + 1e-5).  F must be a multiple of 4.                           */
int vfr_segment_pool_norm_f32(const float *frames, int T, int F, int seg_len, int mode, float *seg, float *ctx,
                              vfr_stream_t stream);
/* many videos at once: frames [sum T, F], frame_offsets [Nv+1] (int32, device),
 * seg_offsets [Nv+1] = prefix sums of ceil(T_v/seg_len), total_segments = its last entry (host
 * copy); seg [total_segments, F], ctx [Nv, F]                                                   */
int vfr_segment_pool_norm_batch_f32(const float *frames, const int32_t *frame_offsets, const int32_t *seg_offsets,
                                    int Nv, int total_segments, int F, int seg_len, int mode, float *seg,
                                    float *ctx, vfr_stream_t stream);

/* ---- a4 + a7  clip encoder: make_visual_features model/data.py:204-213 + CALModel visual
 * branch model/models.py:21-26,55-56.  Row t of video v is [seg_t | ctx_v | t/n | (t+1)/n] x
 * W1[hid, 2F+2] -> ReLU -> W2[D, hid]; the 2F+2 concat is never materialised and the context
 * half of W1 is applied once per video.  clip_offsets [Nv+1] int32; out [total_clips, D].       */
size_t vfr_visual_mlp_workspace_bytes(int total_clips, int Nv, int F, int hid);
int vfr_visual_mlp_f32(const float *seg, const float *ctx, const int32_t *clip_offsets, int Nv, int total_clips,
                       int F, const float *W1, const float *b1, const float *W2, const float *b2, int hid, int D,
                       float *out, void *workspace, size_t workspace_bytes, vfr_stream_t stream);

/* ---- generic Linear (+ReLU): nn.Linear at model/models.py:31,59 (BERT branch) and the VGG
 * classifier[0], classifier[3] (get_rgb_features.py:126).  out[M,N] = A[M,K] W[N,K]^T + b      */
int vfr_linear_f32(const float *A, int64_t M, int K, const float *W, const float *b, int N, int relu, float *out,
                   vfr_stream_t stream);

/* ---- a8  query encoder: CALModel GloVe branch model/models.py:61-66 (+ init_hidden :50-52):
 * Embedding gather [-> unit-norm x learnable length when len_tab != NULL, :62-64] -> BiLSTM(H),
 * h0 = c0 = 0, all T steps including pads -> h_n [fwd|bwd] -> Linear(2H, D).
 * tokens [B,T] int64; emb [vocab,E]; W_ih [4H,E], W_hh [4H,H], b_* [4H] (gate order i,f,g,o)
 * for the forward (_f) and reverse (_b) directions; Wfc [D,2H]; out [B,D].
 * The workspace also holds the per-vocabulary input-projection table (emb x W_ih^T, both directions) that the
 * recurrent steps start their gate chains from when vocab <= 32768 -- hence `vocab` in its size.
 * B <= 2 at a shape whose 32-column weight slices fit a CU's LDS (E + H <= 1170, H <= 1024, 2 * ceil(H / 8) <= CUs), and
 * 3 <= B <= 32 at E = 100, H = 1000 (weights resident in registers as MFMA fragments): the
 * whole sequence and the Linear run as ONE launch of 2 * ceil(H / 8) workgroups that wait on each other's h every step, so
 * ALL of them must be resident: do not run two such calls concurrently on one device (two streams / processes can starve
 * each other of CUs -- the host checks the grid against the device's CU count, it cannot see what else is running).  The
 * waits are bounded, and a give-up is REPAIRED inside the same call: behind every such launch a rescue kernel is enqueued
 * that returns at once in the normal case and otherwise re-encodes the batch without any cross-workgroup dependency (one
 * workgroup per query; milliseconds) -- `out` never holds NaN from a give-up and the return code stays VFR_OK because the
 * result is right.  The event is reported through the fault word: vfr_set_fault_word / vfr_poll_faults below.
 * vfr_set_option("lstm_persist", 0) selects one launch per step instead.                                            */
size_t vfr_bilstm_workspace_bytes(int64_t B, int T, int E, int H, int vocab);
int vfr_bilstm_final_f32(const int64_t *tokens, int64_t B, int T, const float *emb, int vocab, const float *len_tab,
                         const float *Wih_f, const float *Whh_f, const float *bih_f, const float *bhh_f,
                         const float *Wih_b, const float *Whh_b, const float *bih_b, const float *bhh_b, int E,
                         int H, const float *Wfc, const float *bfc, int D, float *out, void *workspace,
                         size_t workspace_bytes, vfr_stream_t stream);

/* Fault word: kernels cannot return codes, and no entry point synchronises.  A caller that wants to HEAR about device-side
 * recoveries registers one device-accessible HOST word (page-locked: hipHostMalloc / torch pin_memory; process-wide, any
 * device; NULL unregisters; it must outlive every call made while registered).  Kernels raise bits in it with system-scope
 * atomics; vfr_poll_faults() (host only, no synchronisation: poll after the caller's own sync point) returns and clears the
 * bits raised since the last poll and leaves the explanation in vfr_last_error().
 *   VFR_FAULT_SEQ_RESCUED  a single-launch sequence encoder gave up and the batch was re-encoded by the rescue kernel     */
#define VFR_FAULT_SEQ_RESCUED 1
int vfr_set_fault_word(uint32_t *word_host);
int vfr_poll_faults(void);

/* ---- a10  scoring core: model/evaluate.py:49-58 (same code evaluate_single.py:48-53,
 * main.py:148-157).  dist[c] = ||(V[c] - q) + eps||_2 (F.pairwise_distance), score of moment
 * (s,e) = mean(dist[s..e]); moments of a video in utils.generate_moments order
 * (model/utils.py:71-75), videos concatenated in clip_offsets order.
 * moment_offsets [Nv+1] int64 = prefix sums of n_v(n_v+1)/2.                                    */
/* dense: scores [Nq, moment_offsets[Nv]]                                                        */
int vfr_score_moments_f32(const float *Q, int64_t Nq, const float *V, const int32_t *clip_offsets,
                          const int64_t *moment_offsets, int Nv, int max_clips, int64_t total_moments, int D,
                          float eps, float *scores, vfr_stream_t stream);
/* each query against ONE video own[q] (evaluate_single.py:48-53): scores [Nq, Mmax], tail +inf  */
int vfr_score_own_f32(const float *Q, int64_t Nq, const float *V, const int32_t *clip_offsets, const int32_t *own,
                      int max_clips, int D, float eps, int Mmax, float *scores, vfr_stream_t stream);

/* ---- a12  ranking: np.argsort over all moments of all videos + R@k / MR, evaluate.py:67-80.
 * Fused scoring + selection: never materialises the [Nq, sum M] score matrix.  Order is
 * (distance, global moment id) ascending (the stable argsort).
 *   k > 0 : out_dist [Nq,k] / out_idx [Nq,k] (int64 global moment ids) = the k best, padded
 *           with (+inf, -1) when fewer than k moments exist
 *   num_rank > 0 (<= 4) : rank_dist / rank_idx / count_lt are [num_rank, Nq]; count_lt[r][q] (int64)
 *           += number of moments that sort strictly before (rank_dist[r][q], rank_idx[r][q]) --
 *           evaluate.py:77's MR when that pair is the best ground-truth-positive moment of IoU
 *           threshold r.  Counts are ADDED (zero the array first, or chain shards).
 * thr_seed (nullable, [Nq] packed keys (fp32 bits of distance << 32 | id)): an upper bound on each query's k-th best
 *           key, known to the caller (the merged sample of all shards, 8e).  Only keys below it are collected and the
 *           internal sample pre-pass is skipped; results are exact for any valid bound.
 * id_base is added to every emitted / compared moment id (shard offset for multi-GPU, 8e).
 * min_clips / max_clips: smallest / largest clip count in the bank (min_clips 0 = unknown; when
 * min == max the kernels drop their per-video length guards).
 * Limits: max_clips <= 64, k <= 448, id_base + total moments < 2^32.                           */
size_t vfr_score_topk_workspace_bytes(int64_t Nq, int Nv, int k);
int vfr_score_topk_f32(const float *Q, int64_t Nq, const float *V, const int32_t *clip_offsets,
                       const int64_t *moment_offsets, int Nv, int total_clips, int min_clips, int max_clips, int D,
                       float eps, int64_t id_base, int k, float *out_dist, int64_t *out_idx, int num_rank,
                       const float *rank_dist, const int64_t *rank_idx, int64_t *count_lt, const int64_t *thr_seed,
                       void *workspace, size_t workspace_bytes, vfr_stream_t stream);
/* The same pass with the clip distances taken from the GEMM form |v|^2 + |q|^2 - 2 v.q on the matrix cores
 * (model/evaluate.py:53-58 as north_star's "MFMA batched query x clip GEMM"; csrc/score_mfma.h):
 *   dtype VFR_MFMA_F32  fp32 MFMA pre-filter with a rigorous error margin; whatever the margin cannot decide is
 *                       re-scored with the exact chain, so every output is BIT-IDENTICAL to vfr_score_topk_f32
 *                       (shapes the pre-filter is not built for are simply handed to vfr_score_topk_f32; 1-8 queries -- a
 *                       serving request -- take a path of their own with lanes = clips / videos, equally exact);
 *   dtype VFR_MFMA_BF16 bf16 operands, fp32 accumulate: count_lt from the approximate
 *                       distances, top-k = exact re-rank of the k + 28 best approximate candidates.  Needs D = 100,
 *                       max_clips <= 21, num_rank in {0, 2}, k <= 253; VFR_EUNSUPPORTED otherwise.
 * Arguments as vfr_score_topk_f32.
 * dtype | VFR_MFMA_BANK_READY: the caller states that `workspace` was last used by a pre-filter call (one for which
 * vfr_score_topk_mfma_prefilter returned 1) on this bank and has not been written since: the bank-side products (mean,
 * centred rows, norms; they sit at the front of the workspace at offsets that depend on total_clips alone) are reused
 * instead of recomputed.  For serving many query batches against one resident bank (~0.1 ms per call at 210 000 clips).
 * CHECKED ON THE DEVICE: every call hashes all of V and the clip offsets (64-bit position-weighted sum, ~20 us at 84 MB)
 * and compares {hash, total_clips, Nv, D, eps, bf16 copy present} with the signature stored next to the products; on any
 * difference the products are recomputed inside the same call (the pre-pass kernels are always enqueued and return at once
 * when the signature matches -- no host decision, no synchronisation).  A bank rewritten through a raw pointer therefore
 * costs a recomputation, never a wrong result.  (The hash is a position-weighted 64-bit sum: it catches accidental rewrites,
 * it is not collision-resistant against an adversary who chooses the bank.)  NOT checked: that the product region of `workspace` itself is intact --
 * that part stays the caller's statement (do not set the flag for a workspace other code has used in between).
 * vfr_score_topk_mfma_prefilter: 1 if a call with these shapes runs the pre-filter (and therefore leaves the bank-side
 * products in its workspace), 0 if it is handed to the exact kernels.                            */
#define VFR_MFMA_F32 0
#define VFR_MFMA_BF16 1
#define VFR_MFMA_BANK_READY 0x100
size_t vfr_score_topk_mfma_workspace_bytes(int64_t Nq, int Nv, int total_clips, int k);
int vfr_score_topk_mfma_prefilter(int64_t Nq, int Nv, int total_clips, int max_clips, int D, int num_rank, int k, int dtype);
int vfr_score_topk_mfma(const float *Q, int64_t Nq, const float *V, const int32_t *clip_offsets,
                        const int64_t *moment_offsets, int Nv, int total_clips, int min_clips, int max_clips, int D,
                        float eps, int64_t id_base, int k, float *out_dist, int64_t *out_idx, int num_rank,
                        const float *rank_dist, const int64_t *rank_idx, int64_t *count_lt, const int64_t *thr_seed,
                        int dtype, void *workspace, size_t workspace_bytes, vfr_stream_t stream);
/* measurement hook (SYNCHRONISES the stream): what the last f32 call on `workspace` (same Nq, Nv, total_clips, k) left to
 * the exact kernels.  stats_host[4] = {query groups of 64, groups handed to the exact fallback, (query, video) pairs
 * re-scored exactly, pairs the (video, group) bitmap can mark = Nq * Nv}.                                             */
int vfr_score_topk_mfma_stats(const void *workspace, int64_t Nq, int Nv, int total_clips, int k, int64_t *stats_host,
                              vfr_stream_t stream);
/* Device self-check of the arithmetic model the pre-filter's margins and the MFMA GEMMs' bit-exactness assume: a
 * v_mfma_f32_16x16x4_f32 = four fp32 fmas, k ascending, on its accumulator (denormals kept).  A, B [16, K] device rows
 * (K % 4 == 0; make them adversarial: wide exponent range, cancellation, denormals): C = A B^T by the matrix pipe against an
 * explicit fmaf chain per element; *mismatches (device int) = elements whose bits differ -- 0 on a conforming device.
 * reversed != 0 runs the reference chain k-descending (what a failing check looks like).  The Python layer runs it once per
 * process before the first "mfma" / "bf16" scoring call and falls back to the exact kernels if it fails.          */
int vfr_mfma_selfcheck(const float *A, const float *B, int K, int reversed, int *mismatches, vfr_stream_t stream);
/* merge G per-shard top-k lists (after the RCCL all-gather): part_dist/part_idx
 * [G, Nq, k] -> out [Nq, k], same (distance, id) order.                                        */
int vfr_topk_merge_f32(const float *part_dist, const int64_t *part_idx, int G, int64_t Nq, int k, float *out_dist,
                       int64_t *out_idx, vfr_stream_t stream);
/* The same exchange with the lists in the form the ranks send them: one int64 key per entry,
 * (fp32 bits of the distance << 32) | moment id, whose signed order is the (distance, id) order;
 * VFR_KEY_EMPTY = (+inf, 0xffffffff) marks an unused slot.  pack: (dist, idx) [n] -> keys [n]
 * (idx < 0 -> empty).  merge: part_keys [G, Nq, k] -> out_dist/out_idx [Nq, k] (both or neither)
 * and/or out_keys [Nq, k] (nullable) -- the merged list as keys, whose column k-1 is the
 * threshold seed of the next pass and which can be merged again without repacking.  _strided: slot
 * g starts at part_keys + g * slot_stride (>= Nq * k int64 elements) -- the lists as they lie inside
 * the rows of a packed exchange buffer, merged in place.                                          */
#define VFR_KEY_EMPTY 0x7F800000FFFFFFFFll
int vfr_topk_pack_keys(const float *dist, const int64_t *idx, int64_t n, int64_t *keys, vfr_stream_t stream);
int vfr_topk_merge_keys(const int64_t *part_keys, int G, int64_t Nq, int k, float *out_dist, int64_t *out_idx,
                        int64_t *out_keys, vfr_stream_t stream);
int vfr_topk_merge_keys_strided(const int64_t *part_keys, int64_t slot_stride, int G, int64_t Nq, int k, float *out_dist,
                                int64_t *out_idx, int64_t *out_keys, vfr_stream_t stream);
/* a12  position of the first ground-truth-positive moment (model/evaluate.py:67-77,
 * np.where(labels[order])[0][0]) needs that moment's key: keys[r, sel[s]] = min over m < M with
 * labels[r, s, m] != 0 of (own_scores[s, m], id_base[s] + m); every other entry of keys [R, Nq]
 * (queries whose video lives on another rank, queries without a positive) = VFR_KEY_EMPTY.
 * own_scores [n_sel, score_stride] from vfr_score_own_f32, labels uint8 [R, n_sel, label_stride]. */
int vfr_gt_best_keys_f32(const float *own_scores, int64_t n_sel, int M, int score_stride, const uint8_t *labels, int R,
                         int label_stride, const int64_t *id_base, const int64_t *sel, int64_t Nq, int64_t *keys,
                         vfr_stream_t stream);
/* The same, plus what a single-shard caller does with the keys next, in the same two launches (a serving request is a chain
 * of short dependent launches: every one saved is ~10 us of its latency): rank_dist / rank_idx [R, Nq] (nullable, together) =
 * the keys unpacked into the rank-key arguments of vfr_score_topk_* ((+inf, 0xffffffff) where there is none); count_zero
 * [R, Nq] int64 (nullable) is zeroed (the count_lt argument those entry points ADD to); *missing (device int, nullable) = 1 iff
 * some (threshold, selected query) has no positive moment, else 0 -- model/evaluate.py:77's IndexError condition.          */
int vfr_gt_rank_keys_f32(const float *own_scores, int64_t n_sel, int M, int score_stride, const uint8_t *labels, int R,
                         int label_stride, const int64_t *id_base, const int64_t *sel, int64_t Nq, int64_t *keys,
                         float *rank_dist, int64_t *rank_idx, int64_t *count_zero, int *missing, vfr_stream_t stream);

/* a11  ground-truth labels (model/evaluate.py:59-62 with utils.get_iou, model/utils.py:78-82; main.py:161 uses >=):
 * labels[r][q][m] = 1 iff >= 2 annotators of query q have IoU > thr[r] (strict != 0) or >= thr[r] with local moment m
 * (utils.generate_moments order) of the query's own video of n_own[q] clips; IoU = float64 intersection / union of the
 * inclusive clip spans.  times int32 [Nq, A, 2] (rows >= nannot[q] unused), thresholds_host: R <= 16 doubles (HOST).
 * labels uint8 [R, Nq, Mmax], zero for m >= n(n+1)/2.                                           */
int vfr_gt_labels_u8(const int32_t *times, const int32_t *nannot, const int32_t *n_own, int64_t Nq, int A,
                     const double *thresholds_host, int R, int strict, int Mmax, uint8_t *labels, vfr_stream_t stream);

/* ---- a1  frame normalisation: DiDeMoDataset.__getitem__ tail, get_rgb_features.py:64-69
 * THWC uint8 -> TCHW fp32, ((x/255) - mean[c]) / std[c] with the ImageNet constants (:34-35)    */
int vfr_frames_normalize_f32(const uint8_t *frames_thwc, int T, int H, int W, float *out_tchw, vfr_stream_t stream);

/* ---- a2  VGG-19 "E" up to fc7: torchvision vgg19.features/avgpool/classifier[0..4],
 * get_rgb_features.py:122-126,145-147.  Layer primitives (NCHW fp32):                           */
int vfr_conv3x3_relu_f32(const float *x, int B, int Cin, int H, int W, const float *w /*[Cout,Cin,3,3]*/,
                         const float *b, int Cout, float *y, vfr_stream_t stream);
int vfr_maxpool2_f32(const float *x, int B, int C, int H, int W, float *y, vfr_stream_t stream);
int vfr_adaptive_avgpool7_f32(const float *x, int B, int C, int H, int W, float *y, vfr_stream_t stream);
/* whole stack: frames [T,H,W,3] u8 -> out [T, fc_dim].  cfg_host: ncfg ints, >0 = conv width,
 * 0 = maxpool (VGG-19: 64,64,0,128,128,0,256x4,0,512x4,0,512x4,0).  conv_w/conv_b: HOST arrays
 * of DEVICE pointers, one per conv.  fc6 [fc_dim, C_last*49], fc7 [fc_dim, fc_dim].             */
size_t vfr_vgg_fc7_workspace_bytes(int T, int H, int W, const int *cfg_host, int ncfg, int fc_dim);
int vfr_vgg_fc7_f32(const uint8_t *frames_thwc, int T, int H, int W, const int *cfg_host, int ncfg,
                    const float *const *conv_w_host, const float *const *conv_b_host, const float *fc6_w,
                    const float *fc6_b, const float *fc7_w, const float *fc7_b, int fc_dim, float *out,
                    void *workspace, size_t workspace_bytes, vfr_stream_t stream);

/* ---- f4  ResNet-152 variant of the extractor: torchvision resnet152 without its fc head, get_rgb_features.py:127-131
 * (conv1 / bn1 / relu / maxpool, Bottleneck layers, global average pool) -> out [T, 32 * width] (2048 at width 64).
 * blocks_host [4] = Bottleneck blocks per layer ({3, 8, 36, 3}); width = 64.  conv_w_host / bn_host: HOST arrays of DEVICE
 * pointers, one per convolution in EXECUTION order -- the stem, then per block conv1 (1x1), conv2 (3x3, carries the
 * stride), conv3 (1x1) and, for the first block of a layer, the 1x1 downsample convolution: conv weight [Cout, Cin, k, k]
 * (torchvision layout) and its BatchNorm as one packed [4, Cout] tensor (weight, bias, running_mean, running_var).  Eval-mode
 * BatchNorm is folded into the convolution (csrc/resnet.hip); bn_eps = 1e-5 for torchvision's modules.                    */
size_t vfr_resnet_pool_workspace_bytes(int T, int H, int W, const int *blocks_host, int width);
int vfr_resnet_pool_f32(const uint8_t *frames_thwc, int T, int H, int W, const int *blocks_host, int width,
                        const float *const *conv_w_host, const float *const *bn_host, float bn_eps, float *out, void *workspace,
                        size_t workspace_bytes, vfr_stream_t stream);
/* The same with the BatchNorm folding and tap-major repacking of the 155 convolutions done ONCE per model instead of once per
 * call (0.6 ms of a 36 ms video): vfr_resnet_fold_f32 writes the folded weights and betas into a caller-owned buffer of
 * vfr_resnet_folded_bytes; vfr_resnet_pool_folded_f32 runs the stack on it (same workspace size, same bits).                */
size_t vfr_resnet_folded_bytes(const int *blocks_host, int width);
int vfr_resnet_fold_f32(const int *blocks_host, int width, const float *const *conv_w_host, const float *const *bn_host, float bn_eps,
                        void *folded, size_t folded_bytes, vfr_stream_t stream);
int vfr_resnet_pool_folded_f32(const uint8_t *frames_thwc, int T, int H, int W, const int *blocks_host, int width, const void *folded,
                               size_t folded_bytes, float *out, void *workspace, size_t workspace_bytes, vfr_stream_t stream);

/* ---- f2  training / test loss: Trainer.ranking_loss, model/main.py:214-232 (called from train_epoch :63 and
 * test_epoch :102).  posit, inter [P,D], intra [Nn,D], lang [S,D]; maskp [P], maskn [Nn] int64 sample ids (rows whose id
 * is outside [0,S) belong to no sample).  loss[0] = sum_i relu(c_posit - c_intra + b) + lamb*relu(c_posit - c_inter + b),
 * c_x = mean pairwise_distance(x rows of sample i, lang[i]).  The optional normalize_loss row scaling (:219-223) is the
 * caller's (it is plain differentiable arithmetic on the inputs).  The workspace written by the forward call (row distances,
 * per-sample terms) is what the backward call reads; grad_loss is the upstream scalar gradient on the device.            */
size_t vfr_ranking_loss_workspace_bytes(int64_t P, int64_t Nn, int S);
int vfr_ranking_loss_f32(const float *posit, const float *intra, const float *inter, const float *lang, const int64_t *maskp,
                         const int64_t *maskn, int64_t P, int64_t Nn, int S, int D, float b, float lamb, float eps,
                         float *loss, void *workspace, size_t workspace_bytes, vfr_stream_t stream);
int vfr_ranking_loss_grad_f32(const float *posit, const float *intra, const float *inter, const float *lang,
                              const int64_t *maskp, const int64_t *maskn, int64_t P, int64_t Nn, int S, int D, float lamb,
                              float eps, const float *grad_loss, const void *workspace, float *grad_posit, float *grad_intra,
                              float *grad_inter, float *grad_lang, vfr_stream_t stream);

/* ---- f2 (second half)  the encoders' backward: loss.backward() through CALModel.forward, model/main.py:58-67 with
 * model/models.py:21-26 (visual_fc) and :40-47,61-66 (lstm, lang_fc).  The gradient contractions are vfr_linear_f32 on
 * transposed operands (dX = dY W, dW = dY^T X); these are the pieces around them (train.py wires them into autograd):
 *   transpose      out [cols, rows] = in [rows, cols]^T
 *   colsum         out[c] = sum_r in[r][c] in a fixed order (16 row phases, then the phases; bias gradients; deterministic)
 *   relu_backward  out = act > 0 ? grad : 0
 *   lstm_cell_forward   one step of nn.LSTM's cell (gates i, f, g, o): pre [B,4H] = h_prev W_hh^T + b_hh, the input part
 *                  x_t W_ih^T + b_ih of batch row b at xproj + b * x_stride; writes the ACTIVATED gates [B,4H], c, h [B,H]
 *   lstm_cell_backward  one step of BPTT: (dh [B,H], dc [B,H] in/out) + the step's gates and cell states -> dpre [B,4H]
 *   bilstm_train_forward   the whole recurrence of both directions over dense inputs X [B,T,E] (E, H multiples of 4): T
 *                  launches of the inference path's fused step, which here also stores the activated gates.  Buffers are
 *                  [direction][step]-major: gates [2][T][B][4H], cs / hs [2][T+1][B][H] (slot 0 = the zero initial state,
 *                  written here; step s of the reverse direction read time T-1-s); h_n = hs[d][T].
 *   bilstm_train_backward  BPTT for both directions: gout [B,2H] = gradient of [h_fwd | h_bwd]; WhhT_d [H,4H] = W_hh,d^T;
 *                  dpre [2][T][B][4H] = gradient of every step's gate pre-activations (the weight gradients are GEMMs over
 *                  it).  Per step one cell-backward launch and one GEMM grid for both directions, K cut into ranges whose
 *                  partial products the next cell-backward adds in a fixed order (deterministic).                          */
int vfr_transpose_f32(const float *in, int64_t rows, int64_t cols, float *out, vfr_stream_t stream);
int vfr_colsum_f32(const float *in, int64_t rows, int cols, float *out, vfr_stream_t stream);
int vfr_relu_backward_f32(const float *grad, const float *act, int64_t n, float *out, vfr_stream_t stream);
int vfr_lstm_cell_forward_f32(const float *pre, const float *xproj, int64_t x_stride, const float *c_prev, int64_t B, int H,
                              float *gates, float *c, float *h, vfr_stream_t stream);
int vfr_lstm_cell_backward_f32(const float *dh, float *dc, const float *gates, const float *c_prev, const float *c_cur, int64_t B,
                               int H, float *dpre, vfr_stream_t stream);
int vfr_bilstm_train_forward_f32(const float *X, int64_t B, int T, int E, int H, const float *Wih_f, const float *Whh_f,
                                 const float *bih_f, const float *bhh_f, const float *Wih_b, const float *Whh_b,
                                 const float *bih_b, const float *bhh_b, float *gates, float *cs, float *hs, vfr_stream_t stream);
size_t vfr_bilstm_train_backward_workspace_bytes(int64_t B, int H);
int vfr_bilstm_train_backward_f32(const float *gout, const float *gates, const float *cs, const float *WhhT_f, const float *WhhT_b,
                                  int64_t B, int T, int H, float *dpre, void *workspace, size_t workspace_bytes, vfr_stream_t stream);

/* ---- parity probe: elementwise canonical math (0 exp, 1 sigmoid, 2 tanh, 3 x/y, 4 sqrt,
 * 5 fma(x,y,x)) so tests can pin the device's transcendental forms against the oracle's.        */
int vfr_math_f32(int op, const float *x, const float *y, float *out, int64_t n, vfr_stream_t stream);

/* ---- tuning / introspection (tests and bench only) -------------------------------------------
 * vfr_set_option("gemm", 0|1): 0 = LDS-tiled VALU chain kernels, 1 = MFMA kernels (default).
 * Both produce identical bits; the switch exists so tests can prove that on the device.
 * "score_fast" 0|1 (the fused D=100 scorer vs the generic one), "score_split" 0|1 (top-k and rank keys in separate
 * launches), "lstm_tile" 0|1|2|3 (fused LSTM step tile rows: automatic | 64 | 128 | 32), "gemm_pp" 0|1 (experimental ping-pong
 * schedule of the large MFMA GEMMs, default 0), "score_pre_b" N (videos in the top-k threshold ladder's stage B; 0 = Nv/16
 * capped at 640), "score_tasks" N (wave-tasks the fused scorer's plan aims for; 0 = automatic), "lstm_skip0" 1|0 (the first LSTM step
 * skips its recurrent segment because h_0 = 0 | runs it), "score_smallq" N (vfr_score_topk_mfma, f32: batches of up to N <= 64 queries against banks of <= 21 clips per video are scored with lanes = clips / videos, the top-k by video selection, default 64; 0: always the fused kernels), "lstm_small" N (batches of up to N <= 4 queries take the vector-chain LSTM step, default 2; 0: always the MFMA tiles), "score_mfma_min" N (vfr_score_topk_mfma hands banks of fewer than N videos to the exact kernels, default 128), "gemm_small" 0|64 (GEMMs of under 384 128-row tiles: 32-row tiles |
 * 64-row tiles), "lstm_xcd" 1|0 (XCD-aware workgroup order of the fused LSTM step | launch order), "lstm_persist" 1|0 (up to 32
 * queries: the whole BiLSTM sequence in ONE launch -- one or two queries: weight slices resident in LDS, vector chains; above
 * "lstm_persist_min" (default 2) at the model's shape: weights resident in registers as MFMA fragments -- with h handed between
 * workgroups as tagged granules | one launch per step), "score_smallq_select" 1|0 (few-queries top-k by video selection | key
 * array + selection tree), "vgg_fuse_pool" 1|0 (a 2x2 max-pool behind a VGG convolution runs in that convolution's epilogue |
 * its own kernel), "vgg_direct1" 1|0 (first VGG convolution as the direct kernel | the implicit-GEMM MFMA kernel), "vgg_halo" 1|0 (the VGG stack on
 * halo-padded activations where its shape allows: convolution loader without tap masks | unpadded), "lstm_fast" 1|0 (the
 * select-free instantiation of the table-start LSTM step where the launch qualifies | always the general form),
 * "lstm_multi" 0|1 (EXPERIMENT, default 0: one launch per LSTM step | all T steps of both directions in one launch -- ordered task
 * lists per XCD group, completion counters, agent-scope state traffic; with "lstm_tile" 2: 128-row tiles; a give-up is repaired by a
 * rescue kernel and reported through the fault word like the sequence kernels'),
 * "score_kth_seed" 1|0 (stage A of the top-k threshold ladder takes its seed by bisection on the candidates' score bits | by the merge kernel),
 * "score_smallq_rank" N (few-queries path with video selection: from N queries on -- default 8 -- the rank counts run with lane = video,
 * straight-line triangle, 3-instruction exact quotient; 0: always the 16-threads-per-video moment kernel),
 * "score_defer" N (vfr_score_topk_mfma, f32: whole-video early-out of the rank half of the moment triangle when at most N lanes of
 * a wave are left undecided by the video's smallest / largest clip distance -- those are re-counted exactly; default 8, -1: off),
 * "score_sort" 1|0|2 (the pre-filter pass with rank keys runs on the batch sorted by difficulty from 1024 queries x 4096 videos
 * on | caller's order | sorted whatever the size), "score_hist" 1|0 (the main top-k launch tightens its threshold from a histogram
 * of the candidates found so far | stage B's threshold throughout) -- same bits either way.  Options are process-global (see Conventions).        */
int vfr_set_option(const char *name, int value);
int vfr_get_option(const char *name);
/* vfr_set_option("profile", 1): every instrumented launch is bracketed by two hipEvents recorded on
 * the launch stream.  After the caller has synchronised the device, vfr_profile_read(site, ...) folds
 * the finished pairs and returns the site's accumulated device time and launch count (reset != 0
 * clears all sites afterwards).  Sites are 1 .. vfr_profile_sites()-1, named by
 * vfr_profile_site_name().  This is how bench.py measures the dominant kernel's launch duration.   */
int vfr_profile_sites(void);
const char *vfr_profile_site_name(int site);
int vfr_profile_read(int site, double *total_ms, int64_t *launches, int reset);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* VFR_H */
