// Shared helpers for the gfx950 kernels of libttrap_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
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

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) {   // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

__device__ __forceinline__ float elu1(float a) { return a > 0.f ? a : (__expf(a) - 1.f); }
// derivative of ELU expressed through its output y
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.f ? 1.f : (y + 1.f); }

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
