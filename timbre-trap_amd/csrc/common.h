// Shared helpers for the gfx950 kernels of libttrap_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/ttrap.h"

#define TT_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t _e = hipGetLastError();                  \
        if (_e != hipSuccess) return (int)_e;               \
    } while (0)

#define TT_HIP(call)                                        \
    do {                                                    \
        hipError_t _e = (call);                             \
        if (_e != hipSuccess) return (int)_e;               \
    } while (0)

static inline hipStream_t tt_stream(void* s) { return (hipStream_t)s; }

// hipFuncSetAttribute acts on the CURRENT device's copy of a kernel: every launch helper remembers, per device, that it
// has raised the dynamic-LDS limit of its kernel (one bit per device ordinal, set with an atomic OR so that threads
// driving different devices may race harmlessly -- the attribute call is idempotent).
struct AttrOnce {
    unsigned long long done[4] = {0ull, 0ull, 0ull, 0ull};
    int pending() const {                      // device ordinal that still needs the attribute, or -1
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        dev &= 255;
        return ((__atomic_load_n(&done[dev >> 6], __ATOMIC_RELAXED) >> (dev & 63)) & 1ull) ? -1 : dev;
    }
    void mark(int dev) { __atomic_fetch_or(&done[dev >> 6], 1ull << (dev & 63), __ATOMIC_RELAXED); }
};

// Run-time switches (environment).  tt_switch: the documented ones, which the tests exercise -- always read.
// tt_tune: knobs of the A/B scripts under tools/ (workgroups per CU, alternative tile shapes, ablations, measured-and-dropped
// variants): compiled OUT of the shipped library, where they return their default; a library built with -DTTRAP_EXPERIMENTAL
// (tools/build_variant.sh <name> -DTTRAP_EXPERIMENTAL, selected with TTRAP_LIB) reads them.
static inline int tt_switch(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#ifdef TTRAP_EXPERIMENTAL
static inline int tt_tune(const char* name, int dflt) { return tt_switch(name, dflt); }
static inline bool tt_tune_set(const char* name) { return getenv(name) != nullptr; }
#else
static inline int tt_tune(const char*, int dflt) { return dflt; }
static inline bool tt_tune_set(const char*) { return false; }
#endif

// Persistent kernels launch min(work items, tt_cus() * workgroups per CU) workgroups.  256 CUs on MI355X;
// tt_set_cu_limit (include/ttrap.h) lowers the figure so that small shapes run the multi-tile loops (tests, tuning).
extern int g_tt_cu_limit;                       // defined in losses.hip
static inline int tt_cus() { return g_tt_cu_limit; }

// Static loss scale of the fp16 backward (tt_set_loss_scale, include/ttrap.h): the factor S carried by every 16-bit activation
// gradient the CALLING THREAD hands to the backward entry points from now on.  Read on the host at launch time and passed to the
// kernels by value, so it is ordered with the stream like any other argument; thread-local, so concurrent callers (autograd worker
// threads of different devices) do not see each other's setting.  1 (the default) is the plain path.
extern thread_local float g_tt_loss_scale;      // defined in losses.hip
static inline float tt_loss_scale() { return g_tt_loss_scale; }
static inline float tt_loss_unscale() { return 1.0f / g_tt_loss_scale; }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) {   // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// ELU and its derivative through the output, exact fp32 path (the arithmetic every CPU-restatement parity test pins): compare + select like torch -- a > 0 gives a
// itself (v_med3(a, exp(a) - 1, 0), tried in round 3, returns exp(a) - 1 for 0 < a < 3e-4 where the fp32 difference falls below a:
// up to 1.2e-7 off) and a NaN stays a NaN (the median and v_min forms return 0 / 1 for a NaN operand, which would hide a diverged
// weight from the loss).
__device__ __forceinline__ float elu1(float a) { return a > 0.f ? a : __expf(a) - 1.f; }
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.f ? 1.f : y + 1.f; }

// 64-lane wave reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
