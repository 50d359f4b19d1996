// Shared pieces of the 16-bit channel-innermost kernels (conv_wide_bf16.hip, conv_level_bf16.hip, conv_stride_bf16.hip,
// latent_bf16.hip, conv_edge_bf16.hip) for gfx950.
//
// ELEMENT TYPE.  These five sources are compiled TWICE (timbre_trap/_hip.py): as written with e16 = bf16 -- the "bf16 MFMA conv
// path" of BASELINE config[2] -- and with -DTT_F16 with e16 = fp16, the dtype the reference's own train step runs in
// (experiments/train.py:415: torch.autocast('cuda') defaults to float16): the same kernels, the same layouts (every operand is 16
// bits), v_mfma_f32_*_f16 instead of *_bf16 at the same rate, 11 instead of 8 significant bits in every stored activation and
// operand, a range of 6e-5 .. 65504 (normal) instead of fp32's.  The second build exports every entry point with the suffix _h
// (e16_names.h); include/ttrap.h declares both sets.
#pragma once
#include "e16_names.h"
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#if defined(TT_F16)
typedef _Float16 e16;
#else
typedef __bf16 e16;
#endif
typedef e16 e16x8 __attribute__((ext_vector_type(8)));
typedef e16 e16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256;                        // threads per workgroup (4 waves)

__device__ float4 g_wzero16;                   // DMA source of out-of-image pieces

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same with the non-temporal hint (cache-policy bit 1 = nt on gfx940+): for tensors a backward kernel reads for the LAST time -- the block input x and
// the hidden activation h1, both written by the forward pass tens of milliseconds earlier -- so that they do not displace the gradients, which the previous
// kernel wrote a moment ago and the next one reads at once, from the L2 / memory-side cache (round 6; TT_BWD_NT_X / TT_BWD_NT_H1).  Measured
// (profiles/r06_cache_hints_ab.txt): x streamed 50.43 / 50.13 / 50.68 -> 50.13 / 49.99 / 50.17 ms per step; h1 streamed AS WELL 50.41 / 50.65 / 50.60 -- its
// tiles carry a halo that the neighbouring tiles re-read, and a streamed line is gone by then: h1 stays an ordinary load.
#ifndef TT_BWD_NT_X
#define TT_BWD_NT_X 1
#endif
#ifndef TT_BWD_NT_H1
#define TT_BWD_NT_H1 0
#endif
__device__ __forceinline__ void glds16_nt(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}
__device__ __forceinline__ void glds16_x(const void* gsrc, void* lds_wave_base) { if (TT_BWD_NT_X) glds16_nt(gsrc, lds_wave_base); else glds16(gsrc, lds_wave_base); }
__device__ __forceinline__ void glds16_h1(const void* gsrc, void* lds_wave_base) { if (TT_BWD_NT_H1) glds16_nt(gsrc, lds_wave_base); else glds16(gsrc, lds_wave_base); }
// A (16 x 32) . B (32 x 16): lane l holds row / column l & 15 and k = 8 (l >> 4) + j; D: column l & 15, rows 4 (l >> 4) + r
#if defined(TT_F16)
__device__ __forceinline__ f32x4 mma32(e16x8 a, e16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
#else
__device__ __forceinline__ f32x4 mma32(e16x8 a, e16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#endif
// K = 16: lane l holds k = 4 (l >> 4) + j
// and sixteen independent 4 x 4 x 4 products per wave (block = lane / 4): the lane-per-pixel form of the narrow levels
#if defined(TT_F16)
__device__ __forceinline__ f32x4 mma16(s16x4 a, s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(e16x4, a), __builtin_bit_cast(e16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma4(s16x4 a, s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(e16x4, a), __builtin_bit_cast(e16x4, b), c, 0, 0, 0);
}
#else
__device__ __forceinline__ f32x4 mma16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mma4(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0); }
#endif

__device__ __forceinline__ int xcd_order(int v, int n) {          // see conv_mfma.hip: one contiguous eighth of the raster per XCD
    const int per = n >> 3;
    return v < (per << 3) ? (v & 7) * per + (v >> 3) : v;
}
// (Round 6, measured and dropped: the persistent tile walks of CONSECUTIVE kernels in OPPOSITE directions -- the residual blocks of a level alternating by
//  dilation, k_wrb_dxw against k_wrb_bwd_a -- so that a consumer starts with the tiles its producer wrote last and finds them in the 256 MB memory-side
//  cache (a 64-clip tensor is 283 MB): 50.47 / 50.02 / 49.82 / 50.03 ms per step without, 50.07 / 49.80 / 49.98 / 50.35 with -- nothing;
//  profiles/r06_cache_hints_ab.txt.)
// ELU and its derivative with as few vector instructions as the values allow (the bf16 kernels are bound by VALU issue:
// PMC of round 3, profiles/r03_*): with e = exp(a) - 1, ELU(a) = a for a > 0 (the smaller of a and e, above 0) and e for a <= 0
// (the larger, at most 0) = the median of (a, e, 0): one v_med3_f32 instead of compare + select.  Two caveats, both handled here:
//   * in fp32, e < a for 0 < a < ~3e-4 (the difference exp(a) - 1 is quantised to 2^-23), so the median returns e there: at most
//     1.2e-7 below a -- far inside the bf16 rounding of the stored result (2^-9 relative), NOT bitwise the select form;
//   * v_med3 / v_min drop NaNs (med3(NaN, NaN, 0) = 0): a NaN pre-activation (diverged weight, inf - inf) would be zeroed at the next
//     activation and never reach the loss.  The forward form therefore adds a * 0 (one full-rate v_fma: +-0 for finite a, NaN for
//     NaN and inf), so NaN / inf surface in the outputs and losses like in torch's ELU.  The derivative forms (backward only) keep
//     the min: by then the forward has already put the NaN into the loss and into dy.
// ELU'(a) = min(exp(a), 1); expressed through the output h = ELU(a): min(h + 1, 1).
__device__ __forceinline__ float elu_f(float a) { return __builtin_fmaf(a, 0.f, __builtin_amdgcn_fmed3f(a, __expf(a) - 1.f, 0.f)); }
// The median alone (NaN -> 0), for the residual-block forward kernels, which do not need the per-element term: a non-finite INPUT
// reaches their output through the residual add (y = ELU(..) + x), and non-finite PARAMETERS are found once per workgroup when the
// weights are loaded (params_poisoned below) and turn the whole output into NaN.  Two full-rate instructions per element saved.
__device__ __forceinline__ float elu_res(float a) { return __builtin_amdgcn_fmed3f(a, __expf(a) - 1.f, 0.f); }
// The OUTPUT activation of the residual-block forward (after the 1x1 product).  In the fp16 build a third source of non-finite values
// exists that neither the residual add nor the parameter check covers: a FINITE hidden activation above 65504 becomes inf when it is
// stored / fed to the 1x1 product as fp16, the product then holds -inf or inf - inf = NaN, and the median would turn those into -1 / 0:
// a silently wrong finite y where torch under fp16 autocast shows inf / NaN in the loss.  The fp16 build therefore keeps the
// NaN-propagating form here (one v_fma per element); bf16 has fp32's range and keeps the median.
#if defined(TT_F16)
__device__ __forceinline__ float elu_out(float a) { return elu_f(a); }
#else
__device__ __forceinline__ float elu_out(float a) { return elu_res(a); }
#endif
// v * 0 summed over the values a lane loads: +-0 for finite parameters, NaN as soon as one is NaN or inf
__device__ __forceinline__ float poison_acc(float acc, float v) { return __builtin_fmaf(v, 0.f, acc); }
__device__ __forceinline__ bool params_poisoned(float acc) {      // any lane of the wave (every wave loads all parameters)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    return !(acc == 0.f);
}
__device__ __forceinline__ float elu_dpre(float a) { return __builtin_fminf(__expf(a), 1.f); }
__device__ __forceinline__ float elu_dout(float h) { return __builtin_fminf(h + 1.f, 1.f); }

// ds_read_b64_tr_b16 (gfx950 transpose read), per 16-lane group: lane 4j + q supplies the address of 4 consecutive bf16
// (row j, columns 4q..4q+3); lane i receives column i of rows 0..3.  With rows = pixels and columns = 16 channels that is
// the K = pixels operand of the 16-row MFMAs straight from a channel-innermost LDS image.
__device__ __forceinline__ s16x4 lds_tr16(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// 32-bit LDS address of a pointer into shared memory (operand of hand-written ds_* instructions)
__device__ __forceinline__ unsigned lds_addr(const unsigned char* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(p);
}

// (Round 5 tried a bare s_barrier behind lgkmcnt(0) at the top of the persistent tile loops instead of __syncthreads(), whose fence the
// compiler lowers to s_waitcnt vmcnt(0) -- a wait for the previous tile's global stores -- in front of the barrier: no effect on the
// train step (53.42 vs 53.43 ms), slightly worse on inference (33.2 vs 32.8 ms); profiles/r05_barrier_fastp_ab.txt.  k_nrb_bwd_fused
// keeps its own bare barriers, where they carry the deferred wait for the x tile.)

// shape of the partial-sum reduce kernels: a block of 1024 threads = REL consecutive dump elements x RSL slices of the contributors
// (each thread keeps eight loads in flight; 16 x 64 puts 600+ blocks on the chip where 64 x 16 left a third of the CUs idle)
constexpr int REL = 16, RSL = 64;

inline int grid_for(int ntiles, int lds_bytes, int max_per_cu) {
    int per = lds_bytes > 0 ? (160 * 1024) / lds_bytes : max_per_cu;
    if (per > max_per_cu) per = max_per_cu;
    if (per < 1) per = 1;
    const int cap = tt_cus() * per;
    return ntiles < cap ? ntiles : cap;
}

template <class K> int raise_lds(K kernel, int bytes, AttrOnce& once) {
    const int dev = once.pending();
    if (dev >= 0) {
        TT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        once.mark(dev);
    }
    return 0;
}

}  // namespace
