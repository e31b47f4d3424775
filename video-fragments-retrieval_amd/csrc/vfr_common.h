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
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- chain GEMM "NT": out[m][n] = epi( Cin[m][n] | 0  (+)  sum_k A[m][k] * W[n][k] ), one k-ascending
// fp32 fma chain per output (oracle rule R1).  Epilogue flags:
enum : int {
    EPI_NONE = 0,
    EPI_BIAS = 1,      // + bias[n] after the chain
    EPI_BIAS2 = 2,     // + (bias[n] + bias2[n])   (LSTM: b_ih + b_hh)
    EPI_RELU = 4,      // max(x, 0) last
};
struct GemmArgs {
    const float *A; int64_t lda;
    const float *W; int64_t ldw;
    const float *Cin; int64_t ldc;     // nullable
    const float *bias, *bias2;         // nullable
    float *out; int64_t ldo;
    int64_t M; int N; int K;
    int epi;
};
int gemm_nt(const GemmArgs &g, hipStream_t st);

}  // namespace vfr
