// a3: 25-frame pooling + L2 normalisation of fc7 frame features (model/data.py:163-181).
//
// HBM-bound streaming kernel: one 64-lane wave per output row (a segment row or a video's context
// row); lane l owns the float4 chunks l, l+64, ... of the row, so every frame read is a fully
// coalesced 1 KiB wave access.  The sum of squares uses the oracle's R3 tree (per-lane chain over its
// chunks, then an xor butterfly through wave shuffles), which makes the result bit-identical to the
// CPU restatement.
#include "vfr_common.h"

namespace vfr {

__global__ __launch_bounds__(256) void segment_pool_norm_kernel(const float *__restrict__ frames,
                                                                const int32_t *__restrict__ frame_off,
                                                                const int32_t *__restrict__ seg_off, int Nv, int F,
                                                                int seg_len, int mode, int single_T,
                                                                float *__restrict__ seg, float *__restrict__ ctx)
{
    const int lane = threadIdx.x & 63;
    const int64_t S = seg_off ? seg_off[Nv] : (single_T + seg_len - 1) / seg_len;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= S + Nv) return;

    int v;
    int64_t t0, t1;
    float *out;
    if (row < S) {
        if (seg_off) {   // largest v with seg_off[v] <= row
            int lo = 0, hi = Nv;
            while (hi - lo > 1) {
                int mid = (lo + hi) >> 1;
                if (seg_off[mid] <= row) lo = mid; else hi = mid;
            }
            v = lo;
        } else v = 0;
        int64_t fbeg = frame_off ? frame_off[v] : 0, fend = frame_off ? frame_off[v + 1] : single_T;
        int i = (int)(row - (seg_off ? seg_off[v] : 0));
        t0 = fbeg + (int64_t)i * seg_len;
        t1 = t0 + seg_len < fend ? t0 + seg_len : fend;
        out = seg + row * F;
    } else {
        v = (int)(row - S);
        t0 = frame_off ? frame_off[v] : 0;
        t1 = frame_off ? frame_off[v + 1] : single_T;
        out = ctx + (int64_t)v * F;
    }
    const float cnt = (float)(t1 - t0);
    const int nchunk = F >> 2;
    float ss = 0.0f;
    for (int chunk = lane; chunk < nchunk; chunk += 64) {
        const float4 *src = reinterpret_cast<const float4 *>(frames + t0 * F) + chunk;
        float4 acc = *src;
        for (int64_t t = t0 + 1; t < t1; ++t) {
            src += nchunk;
            float4 x = *src;
            if (mode) {
                acc.x = fmaxf(acc.x, x.x); acc.y = fmaxf(acc.y, x.y); acc.z = fmaxf(acc.z, x.z); acc.w = fmaxf(acc.w, x.w);
            } else {
                acc.x = acc.x + x.x; acc.y = acc.y + x.y; acc.z = acc.z + x.z; acc.w = acc.w + x.w;
            }
        }
        if (!mode) { acc.x = acc.x / cnt; acc.y = acc.y / cnt; acc.z = acc.z / cnt; acc.w = acc.w / cnt; }
        ss = __builtin_fmaf(acc.x, acc.x, ss);
        ss = __builtin_fmaf(acc.y, acc.y, ss);
        ss = __builtin_fmaf(acc.z, acc.z, ss);
        ss = __builtin_fmaf(acc.w, acc.w, ss);
        reinterpret_cast<float4 *>(out)[chunk] = acc;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss = ss + __shfl_xor(ss, off, 64);
    const float nrm = __builtin_sqrtf(ss) + 1e-5f;
    for (int chunk = lane; chunk < nchunk; chunk += 64) {   // same lane wrote these: program order suffices
        float4 a = reinterpret_cast<float4 *>(out)[chunk];
        a.x = a.x / nrm; a.y = a.y / nrm; a.z = a.z / nrm; a.w = a.w / nrm;
        reinterpret_cast<float4 *>(out)[chunk] = a;
    }
}


// ------------------------------------------------------------------------------------------------
// Single-pass form for batches (F <= 4096): one 1024-thread workgroup per video, thread = one float4 column chunk.
// Every frame element is read ONCE: the same load feeds the running segment pool and the whole-video (context)
// pool, both in frame order (the canonical sums).  At a segment end the row's sum of squares is taken with the
// oracle's R3 tree through LDS: lane l of wave 0 chains the chunks l, l+64, ... in order, then the xor butterfly.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float row_norm_tree64(float4 v, int chunk, int nchunk, float4 *sh, float *sh_norm)
{
    __syncthreads();                                   // previous use of sh / sh_norm finished
    if (chunk < nchunk) sh[chunk] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        float ss = 0.0f;
        for (int c = threadIdx.x; c < nchunk; c += 64) {
            const float4 a = sh[c];
            ss = __builtin_fmaf(a.x, a.x, ss);
            ss = __builtin_fmaf(a.y, a.y, ss);
            ss = __builtin_fmaf(a.z, a.z, ss);
            ss = __builtin_fmaf(a.w, a.w, ss);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ss = ss + __shfl_xor(ss, off, 64);
        if (threadIdx.x == 0) *sh_norm = __builtin_sqrtf(ss) + 1e-5f;
    }
    __syncthreads();
    return *sh_norm;
}

__global__ __launch_bounds__(1024) void segment_pool_norm_video_kernel(const float *__restrict__ frames,
                                                                      const int32_t *__restrict__ frame_off,
                                                                      const int32_t *__restrict__ seg_off, int F,
                                                                      int seg_len, int mode, float *__restrict__ seg,
                                                                      float *__restrict__ ctx)
{
    __shared__ float4 sh[1024];
    __shared__ float sh_norm;
    const int v = blockIdx.x, chunk = threadIdx.x, nchunk = F >> 2;
    const bool live = chunk < nchunk;
    const int64_t fbeg = frame_off[v], fend = frame_off[v + 1];
    int64_t srow = seg_off[v];
    const float4 *src = reinterpret_cast<const float4 *>(frames) + fbeg * nchunk + (live ? chunk : 0);
    float4 call = make_float4(0.f, 0.f, 0.f, 0.f), cseg = call;
    int in_seg = 0;
    // frames of a segment are requested U at a time (one 16-byte load per thread and frame: with a single load in flight per
    // wave the kernel ran at 4.1-4.4 TB/s, now 4.2-4.5; the accumulation stays in frame order.  A barrier-free two-pass form --
    // streaming pools, then a row-normalisation pass -- was measured slower: 3.8-4.0 TB/s)
    constexpr int U = 5;
    for (int64_t t = fbeg; t < fend;) {
        const int64_t left = seg_len - in_seg, seg_end = t + left < fend ? t + left : fend;
        while (t < seg_end) {
            const int nb = seg_end - t < U ? (int)(seg_end - t) : U;
            float4 x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = live ? src[(int64_t)(u < nb ? u : nb - 1) * nchunk] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u < nb) {
                    if (t + u == fbeg) call = x[u];
                    else if (mode) { call.x = fmaxf(call.x, x[u].x); call.y = fmaxf(call.y, x[u].y); call.z = fmaxf(call.z, x[u].z); call.w = fmaxf(call.w, x[u].w); }
                    else { call.x = call.x + x[u].x; call.y = call.y + x[u].y; call.z = call.z + x[u].z; call.w = call.w + x[u].w; }
                    if (in_seg + u == 0) cseg = x[u];
                    else if (mode) { cseg.x = fmaxf(cseg.x, x[u].x); cseg.y = fmaxf(cseg.y, x[u].y); cseg.z = fmaxf(cseg.z, x[u].z); cseg.w = fmaxf(cseg.w, x[u].w); }
                    else { cseg.x = cseg.x + x[u].x; cseg.y = cseg.y + x[u].y; cseg.z = cseg.z + x[u].z; cseg.w = cseg.w + x[u].w; }
                }
            }
            t += nb; src += (int64_t)nb * nchunk; in_seg += nb;
        }
        {                                               // segment complete (uniform across the workgroup)
            if (!mode) { const float c = (float)in_seg; cseg.x = cseg.x / c; cseg.y = cseg.y / c; cseg.z = cseg.z / c; cseg.w = cseg.w / c; }
            const float nrm = row_norm_tree64(cseg, chunk, nchunk, sh, &sh_norm);
            if (live) {
                float4 o;
                o.x = cseg.x / nrm; o.y = cseg.y / nrm; o.z = cseg.z / nrm; o.w = cseg.w / nrm;
                reinterpret_cast<float4 *>(seg + srow * F)[chunk] = o;
            }
            ++srow;
            in_seg = 0;
        }
    }
    if (fend > fbeg) {
        if (!mode) { const float c = (float)(fend - fbeg); call.x = call.x / c; call.y = call.y / c; call.z = call.z / c; call.w = call.w / c; }
        const float nrm = row_norm_tree64(call, chunk, nchunk, sh, &sh_norm);
        if (live) {
            float4 o;
            o.x = call.x / nrm; o.y = call.y / nrm; o.z = call.z / nrm; o.w = call.w / nrm;
            reinterpret_cast<float4 *>(ctx + (int64_t)v * F)[chunk] = o;
        }
    }
}

static int launch_pool(const float *frames, const int32_t *frame_off, const int32_t *seg_off, int Nv, int64_t rows,
                       int F, int seg_len, int mode, int single_T, float *seg, float *ctx, hipStream_t st)
{
    VFR_REQUIRE(F > 0 && (F & 3) == 0, VFR_EUNSUPPORTED, "segment_pool_norm: F=%d must be a multiple of 4", F);
    VFR_REQUIRE((((uintptr_t)frames | (uintptr_t)seg | (uintptr_t)ctx) & 15) == 0, VFR_EINVAL,
                "segment_pool_norm: buffers must be 16-byte aligned");
    if (rows == 0) return VFR_OK;
    hipLaunchKernelGGL(segment_pool_norm_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st, frames, frame_off,
                       seg_off, Nv, F, seg_len, mode, single_T, seg, ctx);
    VFR_CHECK_LAUNCH("segment_pool_norm_kernel");
    return VFR_OK;
}

}  // namespace vfr

extern "C" {

int vfr_segment_pool_norm_f32(const float *frames, int T, int F, int seg_len, int mode, float *seg, float *ctx,
                              vfr_stream_t stream)
{
    VFR_REQUIRE(frames && seg && ctx && T > 0 && seg_len > 0 && (mode == 0 || mode == 1), VFR_EINVAL,
                "vfr_segment_pool_norm_f32: bad argument");
    int64_t rows = (T + seg_len - 1) / seg_len + 1;
    return vfr::launch_pool(frames, nullptr, nullptr, 1, rows, F, seg_len, mode, T, seg, ctx, vfr::as_stream(stream));
}

int vfr_segment_pool_norm_batch_f32(const float *frames, const int32_t *frame_offsets, const int32_t *seg_offsets,
                                    int Nv, int total_segments, int F, int seg_len, int mode, float *seg, float *ctx,
                                    vfr_stream_t stream)
{
    VFR_REQUIRE(frames && frame_offsets && seg_offsets && seg && ctx && Nv >= 0 && total_segments >= 0 &&
                    seg_len > 0 && (mode == 0 || mode == 1),
                VFR_EINVAL, "vfr_segment_pool_norm_batch_f32: bad argument");
    if (Nv == 0) return VFR_OK;
    if (F <= 4096 && (F & 3) == 0 && ((((uintptr_t)frames | (uintptr_t)seg | (uintptr_t)ctx) & 15) == 0)) {
        vfr::ProfScope prof(vfr::SITE_POOL, vfr::as_stream(stream));
        hipLaunchKernelGGL(vfr::segment_pool_norm_video_kernel, dim3((unsigned)Nv), dim3(1024), 0, vfr::as_stream(stream),
                           frames, frame_offsets, seg_offsets, F, seg_len, mode, seg, ctx);
        VFR_CHECK_LAUNCH("segment_pool_norm_video_kernel");
        return VFR_OK;
    }
    return vfr::launch_pool(frames, frame_offsets, seg_offsets, Nv, (int64_t)total_segments + Nv, F, seg_len, mode, 0,
                            seg, ctx, vfr::as_stream(stream));
}

}  // extern "C"
