// ResidualConv2dBlock for the narrow levels (C = 4, 8) of the Timbre-Trap autoencoder on the gfx950 vector ALUs.
// Two implementations of the forward / data-gradient pass: k_small_lds (default: input tile staged by LDS-DMA, taps read
// from LDS with immediate offsets) and k_small (taps straight from global memory; any T / alignment).
//
// At C <= 8 the block moves 2*C*4 bytes per pixel for 20*C*C flops: 10-20 flop/B, below the fp32 ridge of the
// chip (~25 flop/B).  These levels are HBM/latency-bound, a 16-row MFMA tile would be 25-50 % empty, and a
// barrier-per-tile pipeline exposes memory latency.  So: one thread per pixel, all C output channels in
// registers, the 9*C input taps read straight from global memory (coalesced along time, the 3x3 reuse is served
// by the vector L1), weights broadcast from an LDS image ([ci][tap][co], one ds_read_b128 per 4 output channels),
// no barrier in the main loop, 8 waves per SIMD.
//
//   k_small<C,D,0>  y = ELU(W2 . ELU(W1 (*) x + b1) + b2) + x                 (reference modules.py:755-777)
//   k_small<C,D,1>  dx = dy + W1^T (*) dA1                                    (data gradient, flipped weights)
//   k_small_bwd_a   recompute + pointwise chain -> dA1; db1, db2, dW2 in registers, reduced once per workgroup
//   (dW1 stays on the MFMA weight-gradient kernel of conv_mfma.hip)
#include <cstdlib>
#include "common.h"
#include "conv_small.h"

namespace {

constexpr int ROWS_PER_BLOCK = 16;
#ifndef SMALL_RPT
#define SMALL_RPT 2
#endif
#ifndef SMALL_RPT8
#define SMALL_RPT8 2
#endif      // 4 waves x 4 row passes

template <int C>
struct SW {
    static constexpr int W1 = 0;                    // [ci][tap][co]
    static constexpr int W2 = W1 + C * 9 * C;       // [c_in][co2]
    static constexpr int W2T = W2 + C * C;          // [co2][c_in]
    static constexpr int B1 = W2T + C * C;
    static constexpr int B2 = B1 + C;
    static constexpr int FLOATS = B2 + C;
};

template <int C>
__device__ __forceinline__ void build_images(float* lds, const float* __restrict__ w1, const float* __restrict__ b1,
                                             const float* __restrict__ w2, const float* __restrict__ b2, bool flip) {
    using S = SW<C>;
    for (int i = threadIdx.x; i < C * 9 * C; i += blockDim.x) {
        const int co = i % C, tap = (i / C) % 9, ci = i / (9 * C);
        lds[S::W1 + i] = flip ? w1[(ci * C + co) * 9 + (8 - tap)] : w1[(co * C + ci) * 9 + tap];
    }
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
        const int a = i / C, b = i % C;
        lds[S::W2 + i] = w2 ? w2[b * C + a] : 0.f;       // [c_in = a][co2 = b]
        lds[S::W2T + i] = w2 ? w2[a * C + b] : 0.f;      // [co2 = a][c_in = b]
    }
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        lds[S::B1 + i] = b1 ? b1[i] : 0.f;
        lds[S::B2 + i] = b2 ? b2[i] : 0.f;
    }
    __syncthreads();
}

// 3x3 dilated conv of one pixel: acc[co] += sum W1s[ci][tap][co] * x[ci][h + (kh-1)D][t + (kw-1)D].
// The channel loop stays a run-time loop and each kernel row starts with a compiler memory fence: otherwise the
// loop-invariant LDS weight reads are all hoisted into registers (9*C*C of them) and the kernel spills.
template <int C, int D>
__device__ __forceinline__ void conv_pixel(const float* __restrict__ xb, const float* W1s, long plane, int H, int T, int h,
                                           int t, float (&acc)[C]) {
    // tap addresses relative to the channel plane (clamped) and validity, shared by all channels
    int off[9];
    bool ok[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int hh = h + (kh - 1) * D, tt = t + (kw - 1) * D;
            ok[kh * 3 + kw] = hh >= 0 && hh < H && tt >= 0 && tt < T;
            off[kh * 3 + kw] = ok[kh * 3 + kw] ? hh * T + tt : h * T + t;
        }
    // the nine taps of channel ci + 1 are in flight while channel ci is multiplied: one exposed memory latency per
    // pixel instead of one per kernel row
    float cur[9], nxt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cur[k] = xb[off[k]];
#pragma unroll 1
    for (int ci = 0; ci < C; ++ci) {
        const float* xn = xb + (ci + 1 < C ? ci + 1 : ci) * plane;
#pragma unroll
        for (int k = 0; k < 9; ++k) nxt[k] = xn[off[k]];
        const float* wc = W1s + ci * 9 * C;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float xv = ok[kh * 3 + kw] ? cur[kh * 3 + kw] : 0.f;
                const float* wl = wc + (kh * 3 + kw) * C;
#pragma unroll
                for (int co = 0; co < C; ++co) acc[co] = fmaf(xv, wl[co], acc[co]);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) cur[k] = nxt[k];
    }
}

template <int C, int D, int MODE>
__global__ __launch_bounds__(256) void k_small(const float* __restrict__ x, const float* __restrict__ w1,
                                               const float* __restrict__ b1, const float* __restrict__ w2,
                                               const float* __restrict__ b2, const float* __restrict__ res,
                                               float* __restrict__ y, float* __restrict__ h1out, int B, int H, int T) {
    using S = SW<C>;
    __shared__ float lds[S::FLOATS];
    build_images<C>(lds, w1, MODE == 0 ? b1 : nullptr, MODE == 0 ? w2 : nullptr, MODE == 0 ? b2 : nullptr, MODE == 1);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tx;
    const int b = blockIdx.z;
    const long plane = (long)H * T;
    const float* xb = x + (long)b * C * plane;
    if (t >= T) return;
#pragma unroll 1
    for (int pass = 0; pass < ROWS_PER_BLOCK / 4; ++pass) {
        const int h = blockIdx.y * ROWS_PER_BLOCK + pass * 4 + ty;
        if (h >= H) break;
        float acc[C];
#pragma unroll
        for (int co = 0; co < C; ++co) acc[co] = lds[S::B1 + co];
        conv_pixel<C, D>(xb, lds + S::W1, plane, H, T, h, t, acc);
        const long o = (long)b * C * plane + (long)h * T + t;
        if (MODE == 0) {
            float a2[C];
#pragma unroll
            for (int co = 0; co < C; ++co) a2[co] = lds[S::B2 + co];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float hv = elu1(acc[c]);
                if (h1out) h1out[o + c * plane] = hv;
                const float* wl = lds + S::W2 + c * C;
#pragma unroll
                for (int co = 0; co < C; ++co) a2[co] = fmaf(hv, wl[co], a2[co]);
            }
#pragma unroll
            for (int co = 0; co < C; ++co) y[o + co * plane] = elu1(a2[co]) + x[o + co * plane];
        } else {
#pragma unroll
            for (int co = 0; co < C; ++co) y[o + co * plane] = acc[co] + res[o + co * plane];
        }
    }
}

typedef float f32x4s __attribute__((ext_vector_type(4)));
// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4 x 4 x 1 outer products per wave instruction.  Lane l belongs to block l / 4;
// A = one value per lane (row l % 4 of its block), B = one value per lane (column l % 4), result register r of lane l =
// D[block l / 4][row r][column l % 4] (tools/probes/mfma4x4_probe.cpp).  With  A = W[co = l % 4][k]  (the same four weights in
// every block) and  B = the lane's own pixel value for input k,  register r of lane l accumulates output channel r of pixel l:
// exactly the layout of the thread-per-pixel loop  acc[co] = fmaf(x, w[co], acc[co]),  one FMA per product in the same order
// (bit-identical results), but issued on the MATRIX pipe at the vector FMA rate (measured 59.6 vs 53.9 TMAC/s chip-wide) -- the
// vector ALUs keep ELU, addressing and the epilogue, and the weight fetch shrinks to one ds_read_b32 per four channels.
__device__ __forceinline__ f32x4s mfma4(float a, float b, f32x4s c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// ---- LDS-tiled variant ---------------------------------------------------------------------------------------------
// Same arithmetic, input tile (all C channels, 8 rows + halo, 64 columns + halo) brought in by LDS-DMA and double
// buffered like the matrix-core kernels.  A thread still owns one pixel, but its 9 C taps are ds_read_b32 with immediate
// offsets from ONE base address: no per-tap address arithmetic, no validity selects (the halo is zero-filled by the
// DMA), no global loads in the inner loop, and the residual comes from the tile's centre instead of a second global read.
__device__ float4 g_zero16_small;

__device__ __forceinline__ void glds16s(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int xcd_tile_s(int v, int ntiles) {
    const int per = ntiles >> 3;
    return v < (per << 3) ? (v & 7) * per + (v >> 3) : v;
}

// RPT = rows per thread (tile = 8 RPT rows x 64 columns): two rows halve the weight reads per pixel and shrink the halo
// share of the tile; NBUF = 2 double-buffers the DMA (C = 8 with 16-row tiles keeps one buffer and relies on the second
// workgroup of the CU to cover its DMA wait).
template <int C, int D, int RPT, int NBUF>
struct SL {
    static constexpr int TR = 8 * RPT, XR = TR + 2 * D, XCP = 72, PLANE = XR * XCP;
    static constexpr int NQ = C * PLANE / 4, NP = (NQ + 63) / 64;
    static constexpr int BUF = NP * 256;
    static constexpr int LDS_BYTES = (NBUF * BUF + SW<C>::FLOATS) * 4;
};

#ifndef SMALL_FWD4_TR
#define SMALL_FWD4_TR 16  // rows of the four-pixels-per-lane forward tile
#endif
#ifndef SMALL_LB8
#define SMALL_LB8 1      // waves per SIMD the C = 8 forward kernel is compiled for.  Measured: 6 (80 registers, a handful spilled,
                         // three workgroups per CU) 0.75-1.05 ms and 5 0.73-0.78 ms against 0.49-0.54 ms uncapped -- the spills cost
                         // far more than the third workgroup hides
#endif
template <int C, int D, int MODE, int RPT, int NBUF, bool MF>
__global__ __launch_bounds__(512, (C == 8 && MODE == 0 && MF ? SMALL_LB8 : 1)) void k_small_lds(const float* __restrict__ x, const float* __restrict__ w1,
                                                   const float* __restrict__ b1, const float* __restrict__ w2,
                                                   const float* __restrict__ b2, const float* __restrict__ res,
                                                   float* __restrict__ y, float* __restrict__ h1out, int B, int H, int T) {
    using S = SW<C>;
    using L = SL<C, D, RPT, NBUF>;
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    float* xs = lds_dyn;
    float* wimg = lds_dyn + NBUF * L::BUF;
    build_images<C>(wimg, w1, MODE == 0 ? b1 : nullptr, MODE == 0 ? w2 : nullptr, MODE == 0 ? b2 : nullptr, MODE == 1);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_h = (H + L::TR - 1) / L::TR, tiles_t = (T + 63) / 64;
    const int ntiles = B * tiles_h * tiles_t;
    const long plane = (long)H * T;
    const float* zero = reinterpret_cast<const float*>(&g_zero16_small);

    auto issue = [&](int v, int buf) {
        int tt = xcd_tile_s(v, ntiles);
        const int tx = tt % tiles_t; tt /= tiles_t;
        const int ty = tt % tiles_h;
        const int b = tt / tiles_h, row0 = ty * L::TR - D, col0 = tx * 64 - 4;
        const float* xb = x + (long)b * C * plane;
        float* dst = xs + buf * L::BUF;
#pragma unroll
        for (int jj = 0; jj < (L::NP + 7) / 8; ++jj) {
            const int j = wave + 8 * jj;
            if (j < L::NP) {
                const int q = j * 64 + lane;
                const int ci = q / (L::PLANE / 4);
                const int rem = q - ci * (L::PLANE / 4);
                const int r = rem / 18, c4 = rem - r * 18;
                const int h = row0 + r, t = col0 + 4 * c4;
                const bool ok = q < L::NQ && h >= 0 && h < H && t >= 0 && t < T;
                glds16s(ok ? xb + (ci * (int)plane + h * T + t) : zero, dst + j * 256);
            }
        }
    };

    int v = blockIdx.x;
    if (v >= ntiles) return;
    int buf = 0;
    if (NBUF == 2) issue(v, 0);
    for (; v < ntiles; v += gridDim.x) {
        if (NBUF == 1) { __syncthreads(); issue(v, 0); }              // everyone is done with the previous tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (NBUF == 2 && v + (int)gridDim.x < ntiles) issue(v + (int)gridDim.x, buf ^ 1);
        int tt = xcd_tile_s(v, ntiles);
        const int tx = tt % tiles_t; tt /= tiles_t;
        const int ty = tt % tiles_h;
        const int b = tt / tiles_h, h0 = ty * L::TR + wave, t = tx * 64 + lane;
        // tap (kh, kw) of channel ci for row r of this thread: + ci*PLANE + (8 r + kh*D)*XCP + kw*D
        const float* xt = xs + buf * L::BUF + wave * L::XCP + (4 - D) + lane;
        float acc[RPT][C];
        if constexpr (MF) {
            // matrix-pipe form (mfma4 above): lane = pixel, accumulator register = output channel, one group of four channels
            // per instruction; weights arrive one ds_read_b32 per group (four distinct addresses per wave: a broadcast)
            constexpr int NG = C / 4;
            const int l4 = lane & 3;
            f32x4s av[RPT][NG];
#pragma unroll
            for (int r = 0; r < RPT; ++r)
#pragma unroll
                for (int gq = 0; gq < NG; ++gq)
#pragma unroll
                    for (int q = 0; q < 4; ++q) av[r][gq][q] = wimg[S::B1 + 4 * gq + q];
#pragma unroll 1
            for (int ci = 0; ci < C; ++ci) {
                const float* xc = xt + ci * L::PLANE;
                const float* wc = wimg + S::W1 + ci * 9 * C + l4;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        float w[NG];
#pragma unroll
                        for (int gq = 0; gq < NG; ++gq) w[gq] = wc[(kh * 3 + kw) * C + 4 * gq];
#pragma unroll
                        for (int r = 0; r < RPT; ++r) {
                            const float xv = xc[(8 * r + kh * D) * L::XCP + kw * D];
#pragma unroll
                            for (int gq = 0; gq < NG; ++gq) av[r][gq] = mfma4(w[gq], xv, av[r][gq]);
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RPT; ++r)
#pragma unroll
                for (int co = 0; co < C; ++co) acc[r][co] = av[r][co >> 2][co & 3];
        } else {
#pragma unroll
        for (int r = 0; r < RPT; ++r)
#pragma unroll
            for (int co = 0; co < C; ++co) acc[r][co] = wimg[S::B1 + co];
#pragma unroll 1
        for (int ci = 0; ci < C; ++ci) {
            const float* xc = xt + ci * L::PLANE;
            const float* wc = wimg + S::W1 + ci * 9 * C;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float* wl = wc + (kh * 3 + kw) * C;
                    float w[C];
#pragma unroll
                    for (int co = 0; co < C; ++co) w[co] = wl[co];
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        const float xv = xc[(8 * r + kh * D) * L::XCP + kw * D];
#pragma unroll
                        for (int co = 0; co < C; ++co) acc[r][co] = fmaf(xv, w[co], acc[r][co]);
                    }
                }
            }
        }
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int h = h0 + 8 * r;
            if (h < H && t < T) {
                const long o = (long)b * C * plane + (long)h * T + t;
                if (MODE == 0) {
                    float a2[C];
#pragma unroll
                    for (int co = 0; co < C; ++co) a2[co] = wimg[S::B2 + co];
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float hv = elu1(acc[r][c]);
                        if (h1out) h1out[o + c * plane] = hv;
                        const float* wl = wimg + S::W2 + c * C;
#pragma unroll
                        for (int co = 0; co < C; ++co) a2[co] = fmaf(hv, wl[co], a2[co]);
                    }
#pragma unroll
                    for (int co = 0; co < C; ++co) y[o + co * plane] = elu1(a2[co]) + xt[co * L::PLANE + (8 * r + D) * L::XCP + D];
                } else {
#pragma unroll
                    for (int co = 0; co < C; ++co) y[o + co * plane] = acc[r][co] + res[o + co * plane];
                }
            }
        }
        if (NBUF == 2) buf ^= 1;
    }
}

// ---- forward, four pixels per lane -------------------------------------------------------------------------------------------------
// k_small_lds (MODE 0) is bound by LDS INSTRUCTIONS (profiles/r02_pmc_narrow_levels.txt: SQ_LDS_CMD_FIFO_FULL for a third of its busy
// time): one 4-byte tap read per pixel, input channel and tap, one weight read per tap and group of four output channels, 4-byte
// stores.  Here a lane owns FOUR consecutive frames of one row: the taps of a (channel, kernel row) come from three aligned 16-byte
// reads of the tile row (columns 4q - 4 .. 4q + 7 cover 4q - D .. 4q + 3 + D) and are picked out of registers, a weight read serves
// four pixels, the residual and every store are 16 bytes.  Same products in the same order on the same matrix instruction (mfma4:
// one FMA per product), so y and h1 are BIT-IDENTICAL to k_small_lds -- tests/test_gpu_conv.py pins that.
// Tile 16 rows x 64 frames, 256 threads (16 rows x 16 quads), one LDS buffer: three (C = 8) / four (C = 4) workgroups per CU cover each
// other's staging.
template <int C, int D, int TR_>
struct SL4 {
    static constexpr int TR = TR_, XR = TR + 2 * D, XCP = 72, PLANE = XR * XCP;
    static constexpr int NTH = TR * 16, NW = NTH / 64;
    static constexpr int NQ = C * PLANE / 4, NP = (NQ + 63) / 64;
    static constexpr int BUF = NP * 256;
    static constexpr int LDS_BYTES = (BUF + SW<C>::FLOATS) * 4;
};

template <int C, int D, int TR>
__global__ __launch_bounds__((TR * 16), 3) void k_small_fwd4(const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ y,
                                                       float* __restrict__ h1out, int B, int H, int T) {
    using S = SW<C>;
    using L = SL4<C, D, TR>;
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    float* xs = lds_dyn;
    float* wimg = lds_dyn + L::BUF;
    build_images<C>(wimg, w1, b1, w2, b2, false);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = threadIdx.x & 15, row = threadIdx.x >> 4, l4 = lane & 3;
    const int tiles_h = (H + L::TR - 1) / L::TR, tiles_t = (T + 63) / 64;
    const int ntiles = B * tiles_h * tiles_t;
    const long plane = (long)H * T;
    const float* zero = reinterpret_cast<const float*>(&g_zero16_small);
    constexpr int NG = C / 4;

    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        int tt = xcd_tile_s(v, ntiles);
        const int tx = tt % tiles_t; tt /= tiles_t;
        const int ty = tt % tiles_h;
        const int b = tt / tiles_h, row0 = ty * L::TR - D, col0 = tx * 64 - 4;
        const float* xb = x + (long)b * C * plane;
        __syncthreads();                                         // everyone is done with the previous tile
#pragma unroll
        for (int jj = 0; jj < (L::NP + L::NW - 1) / L::NW; ++jj) {
            const int j = wave + L::NW * jj;
            if (j < L::NP) {
                const int p = j * 64 + lane;
                const int ci = p / (L::PLANE / 4);
                const int rem = p - ci * (L::PLANE / 4);
                const int r = rem / 18, c4 = rem - r * 18;
                const int h = row0 + r, t = col0 + 4 * c4;
                const bool ok = p < L::NQ && h >= 0 && h < H && t >= 0 && t < T;
                glds16s(ok ? xb + (ci * (int)plane + h * T + t) : zero, xs + j * 256);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        const int h = ty * L::TR + row, t4 = tx * 64 + 4 * q;
        // window of tile row (row + kh D), columns 4q .. 4q + 11 of the tile = frames t4 - 4 .. t4 + 7
        const float* xt = xs + row * L::XCP + 4 * q;
        f32x4s av[4][NG];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int gq = 0; gq < NG; ++gq)
#pragma unroll
                for (int c = 0; c < 4; ++c) av[p][gq][c] = wimg[S::B1 + 4 * gq + c];
#pragma unroll 1
        for (int ci = 0; ci < C; ++ci) {
            const float* xc = xt + ci * L::PLANE;
            const float* wc = wimg + S::W1 + ci * 9 * C + l4;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                asm volatile("" ::: "memory");
                const float4 q0 = *reinterpret_cast<const float4*>(xc + kh * D * L::XCP);
                const float4 q1 = *reinterpret_cast<const float4*>(xc + kh * D * L::XCP + 4);
                const float4 q2 = *reinterpret_cast<const float4*>(xc + kh * D * L::XCP + 8);
                const float win[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    float w[NG];
#pragma unroll
                    for (int gq = 0; gq < NG; ++gq) w[gq] = wc[(kh * 3 + kw) * C + 4 * gq];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float xv = win[4 + p + (kw - 1) * D];
#pragma unroll
                        for (int gq = 0; gq < NG; ++gq) av[p][gq] = mfma4(w[gq], xv, av[p][gq]);
                    }
                }
            }
        }
        if (h < H && t4 < T) {
            const long o = (long)b * C * plane + (long)h * T + t4;
            float a2[4][C];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int co = 0; co < C; ++co) a2[p][co] = wimg[S::B2 + co];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float hv[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) hv[p] = elu1(av[p][c >> 2][c & 3]);
                if (h1out) *reinterpret_cast<float4*>(h1out + o + c * plane) = float4{hv[0], hv[1], hv[2], hv[3]};
                const float* wl = wimg + S::W2 + c * C;
#pragma unroll
                for (int co = 0; co < C; ++co) {
                    const float wv = wl[co];
#pragma unroll
                    for (int p = 0; p < 4; ++p) a2[p][co] = fmaf(hv[p], wv, a2[p][co]);
                }
            }
#pragma unroll
            for (int co = 0; co < C; ++co) {
                const float4 xr = *reinterpret_cast<const float4*>(xt + co * L::PLANE + D * L::XCP + 4);
                *reinterpret_cast<float4*>(y + o + co * plane) =
                    float4{elu1(a2[0][co]) + xr.x, elu1(a2[1][co]) + xr.y, elu1(a2[2][co]) + xr.z, elu1(a2[3][co]) + xr.w};
            }
        }
    }
}

// recompute + pointwise chain; persistent workgroups accumulate db1, db2, dW2 in registers
template <int C, int D, bool RECOMP>
__global__ __launch_bounds__(256) void k_small_bwd_a(const float* __restrict__ x, const float* __restrict__ h1in,
                                                     const float* __restrict__ dy,
                                                     const float* __restrict__ w1, const float* __restrict__ b1,
                                                     const float* __restrict__ w2, const float* __restrict__ b2,
                                                     float* __restrict__ da1, float* __restrict__ db1, float* __restrict__ dw2,
                                                     float* __restrict__ db2, int B, int H, int T) {
    using S = SW<C>;
    __shared__ float lds[S::FLOATS];
    __shared__ float red[C * C + 2 * C];
    for (int i = threadIdx.x; i < C * C + 2 * C; i += 256) red[i] = 0.f;
    build_images<C>(lds, w1, b1, w2, b2, false);        // ends with a workgroup barrier
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long plane = (long)H * T;
    const int tiles_t = (T + 63) / 64, tiles_h = (H + 3) / 4;
    const int ntiles = B * tiles_h * tiles_t;
    float aw2[C][C], ab1[C], ab2[C];
#pragma unroll
    for (int a = 0; a < C; ++a) {
        ab1[a] = 0.f; ab2[a] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) aw2[a][c] = 0.f;
    }
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int q = tile;
        const int tt = q % tiles_t; q /= tiles_t;
        const int th = q % tiles_h;
        const int b = q / tiles_h;
        const int t = tt * 64 + tx, h = th * 4 + ty;
        if (t >= T || h >= H) continue;
        const long o = (long)b * C * plane + (long)h * T + t;
        float h1[C];
        if (RECOMP) {
            const float* xb = x + (long)b * C * plane;
#pragma unroll
            for (int co = 0; co < C; ++co) h1[co] = lds[S::B1 + co];
            conv_pixel<C, D>(xb, lds + S::W1, plane, H, T, h, t, h1);
#pragma unroll
            for (int co = 0; co < C; ++co) h1[co] = elu1(h1[co]);
        } else {
#pragma unroll
            for (int co = 0; co < C; ++co) h1[co] = h1in[o + co * plane];
        }
        float a2[C];
#pragma unroll
        for (int co = 0; co < C; ++co) a2[co] = lds[S::B2 + co];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float* wl = lds + S::W2 + c * C;
#pragma unroll
            for (int co = 0; co < C; ++co) a2[co] = fmaf(h1[c], wl[co], a2[co]);
        }
        asm volatile("" ::: "memory");
        float d1[C];
#pragma unroll
        for (int c = 0; c < C; ++c) d1[c] = 0.f;
#pragma unroll
        for (int co = 0; co < C; ++co) {
            const float gd = dy[o + co * plane] * elu_grad_from_out(elu1(a2[co]));     // dA2
            ab2[co] += gd;
            const float* wl = lds + S::W2T + co * C;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                d1[c] = fmaf(gd, wl[c], d1[c]);
                aw2[co][c] = fmaf(gd, h1[c], aw2[co][c]);
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float gd = d1[c] * elu_grad_from_out(h1[c]);
            ab1[c] += gd;
            da1[o + c * plane] = gd;
        }
    }
    // one reduction per workgroup: wave shuffle -> LDS -> global atomics
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int a = 0; a < C; ++a) {
        const float s1 = wave_sum(ab1[a]), s2 = wave_sum(ab2[a]);
        if (lane == 0) { atomicAdd(&red[C * C + a], s1); atomicAdd(&red[C * C + C + a], s2); }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float s = wave_sum(aw2[a][c]);
            if (lane == 0) atomicAdd(&red[a * C + c], s);
        }
    }
    __syncthreads();
    if (threadIdx.x < C * C) atomicAdd(dw2 + threadIdx.x, red[threadIdx.x]);
    if (threadIdx.x < C) {
        atomicAdd(db1 + threadIdx.x, red[C * C + threadIdx.x]);
        atomicAdd(db2 + threadIdx.x, red[C * C + C + threadIdx.x]);
    }
}

// ---- fused backward of one narrow residual block ---------------------------------------------------------------------
// One pass instead of three (pointwise chain -> dA1 in HBM -> data gradient -> weight gradient):
//   stage   h1 and dy tiles WITH halo (rows h0 - D .. h0 + TR + D - 1, columns t0 - 4 .. t0 + 67) and the x tile with a row
//           halo (columns t0 .. t0 + 63) by LDS-DMA; out-of-image pieces come from a zero source;
//   phase 1 every pixel of the halo tile: a2 = W2 h1 + b2, dA2 = dy ELU'(a2), dH1 = W2^T dA2, dA1 = dH1 ELU'(h1), written
//           IN PLACE over h1 (dy = 0 outside the image makes dA1 = 0 there: the zero padding the two gradients need);
//           db1, db2, dW2 accumulate in registers over the CENTRE pixels only (every pixel is some tile's centre once);
//   phase 2 data gradient on the vector ALUs from LDS taps (the forward loop with flipped weights, residual = dy from the
//           tile centre) and dW1 on the matrix cores with both operands packed as in k_wgrad3_pack (conv_mfma.hip):
//             A[(kw, co)][(h', t)] = dA1[co][h'][t - (kw-1)D]     (column-shifted rows of the dA1 tile)
//             B[(h', t)][(kh, ci)] = x[ci][h' + (kh-1)D][t]       (row-shifted rows of the x tile)
//           accumulated over every tile of the persistent workgroup, one partial image per workgroup at the end
//           (summed by k_wgrad3_pack_reduce).
// HBM traffic per block: h1, dy, x read once (+ halo, mostly L2 hits under the XCD-ordered tile walk), dx written once:
// 4 tensors instead of 8 (3 + 3 + 2) for the three-kernel path.
template <int C, int D, int RPT, int NW, int NBUF>
struct SF {
    static constexpr int TR = NW * RPT, XR = TR + 2 * D, NT = 64 * NW;
    static constexpr int PLANE = XR * 72 + 4;                 // h1 / dy tile plane pitch: 4 mod 32 (A-fragment reads spread over banks)
    static constexpr int NQH = C * PLANE / 4, NPH = (NQH + 63) / 64, HBUF = NPH * 256;
    static constexpr int QPLANE = XR * 64 + 4;                // x tile (row halo only)
    static constexpr int NQX = C * QPLANE / 4, NPX = (NQX + 63) / 64, XBUF = NPX * 256;
    static constexpr int M = 3 * C, MT = (M + 15) / 16, NC = MT * 16, IMG = MT * 16 * NC;
    static constexpr int SET = 2 * HBUF + XBUF;                // one staging set: h1, dy, x
    static constexpr int TILE_FLOATS = NBUF * SET;
    static constexpr int LDS_FLOATS = TILE_FLOATS + SW<C>::FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(IMG + C * C + 2 * C <= TILE_FLOATS, "final reduction reuses the tile area");
};

template <int C, int D, int RPT, int NW, int NBUF>
__global__ __launch_bounds__(64 * NW, (C <= 4 ? 4 : 2)) void k_small_bwd_fused(const float* __restrict__ x, const float* __restrict__ h1in,
                                                         const float* __restrict__ dy, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         float* __restrict__ dx, float* __restrict__ db1, float* __restrict__ dw2,
                                                         float* __restrict__ db2, float* __restrict__ scratch, int B, int H, int T,
                                                         int ablate) {
    using S = SW<C>;
    using L = SF<C, D, RPT, NW, NBUF>;
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    // staging set `buf`: h1 tile (then dA1 in place) | dy tile | x tile
    float* wimg = lds_dyn + L::TILE_FLOATS;    // W1 flipped [co][tap][ci], W2 [c][co2], W2T [co2][c], (b1 unused), b2
    build_images<C>(wimg, w1, nullptr, w2, b2, true);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, l15 = lane & 15;
    const int tiles_h = (H + L::TR - 1) / L::TR, tiles_t = (T + 63) / 64;
    const int ntiles = B * tiles_h * tiles_t;
    const long plane = (long)H * T;
    const float* zero = reinterpret_cast<const float*>(&g_zero16_small);

    // operand offsets of the packed weight gradient: A rows m = kw*C + co (dA1 tile), B columns n = kh*C + ci (x tile)
    int aoff[L::MT], boff[L::MT];
    bool aok[L::MT];
#pragma unroll
    for (int mt = 0; mt < L::MT; ++mt) {
        const int m = mt * 16 + l15;
        aok[mt] = m < L::M;
        const int kw = aok[mt] ? m / C : 0, co = aok[mt] ? m - kw * C : 0;
        aoff[mt] = co * L::PLANE + (D + wave) * 72 + 4 - (kw - 1) * D + g;          // + NW rr * 72 + 4 sk
        const int n = aok[mt] ? m : L::M - 1;
        const int kh = n / C, ci = n - kh * C;
        boff[mt] = ci * L::QPLANE + (wave + kh * D) * 64 + g;                       // + NW rr * 64 + 4 sk
    }
    f32x4s wacc[L::MT][L::MT];
#pragma unroll
    for (int a = 0; a < L::MT; ++a)
#pragma unroll
        for (int b = 0; b < L::MT; ++b) wacc[a][b] = f32x4s{0.f, 0.f, 0.f, 0.f};
    float aw2[C][C], ab1[C], ab2[C];
#pragma unroll
    for (int a = 0; a < C; ++a) {
        ab1[a] = 0.f; ab2[a] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) aw2[a][c] = 0.f;
    }

    auto stage = [&](int v, int buf) {
        float* hs = lds_dyn + buf * L::SET;
        float* ds = hs + L::HBUF;
        float* xs = hs + 2 * L::HBUF;
        int tt = xcd_tile_s(v, ntiles);
        const int tx = tt % tiles_t; tt /= tiles_t;
        const int ty = tt % tiles_h;
        const int b = tt / tiles_h, row0 = ty * L::TR - D, t0 = tx * 64;
        const long cb = (long)b * C * plane;
        constexpr int PQ = L::PLANE / 4;
#pragma unroll
        for (int jj = 0; jj < (L::NPH + NW - 1) / NW; ++jj) {
            const int j = wave + NW * jj;
            if (j < L::NPH) {
                const int q = j * 64 + lane;
                const int ci = q / PQ;
                const int rem = q - ci * PQ;
                const int r = rem / 18, c4 = rem - r * 18;
                const int h = row0 + r, t = t0 - 4 + 4 * c4;
                const bool ok = q < L::NQH && r < L::XR && h >= 0 && h < H && t >= 0 && t < T;
                const long o = cb + (ci * (int)plane + h * T + t);
                glds16s(ok ? h1in + o : zero, hs + j * 256);
                glds16s(ok ? dy + o : zero, ds + j * 256);
            }
        }
        constexpr int XQ = L::QPLANE / 4;
#pragma unroll
        for (int jj = 0; jj < (L::NPX + NW - 1) / NW; ++jj) {
            const int j = wave + NW * jj;
            if (j < L::NPX) {
                const int q = j * 64 + lane;
                const int ci = q / XQ;
                const int rem = q - ci * XQ;
                const int r = rem >> 4, c4 = rem & 15;
                const int h = row0 + r, t = t0 + 4 * c4;
                const bool ok = q < L::NQX && r < L::XR && h >= 0 && h < H && t < T;
                glds16s(ok ? x + cb + (ci * (int)plane + h * T + t) : zero, xs + j * 256);
            }
        }
    };

    int buf = 0;
    if (NBUF == 2 && (int)blockIdx.x < ntiles) stage(blockIdx.x, 0);
#pragma unroll 1
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        if (NBUF == 1) {
            __syncthreads();                               // everyone is done with the previous tile (and the weight images exist)
            if (!(ablate & 8) || v == (int)blockIdx.x) stage(v, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile has landed ...
        __syncthreads();                                   // ... for every wave, and everyone is done with the other set
        if (NBUF == 2 && v + (int)gridDim.x < ntiles && !(ablate & 8)) stage(v + (int)gridDim.x, buf ^ 1);
        float* hs = lds_dyn + buf * L::SET;
        float* ds = hs + L::HBUF;
        float* xs = hs + 2 * L::HBUF;
        // ---- phase 1: pointwise chain over the halo tile, dA1 in place of h1 ----
#pragma unroll 1
        for (int i = threadIdx.x; i < ((ablate & 1) ? 0 : L::XR * 72); i += L::NT) {
            const int r = i / 72, c = i - r * 72;
            const float centre = (r >= D && r < D + L::TR && c >= 4 && c < 68) ? 1.f : 0.f;
            float h1[C], a2[C];
#pragma unroll
            for (int k = 0; k < C; ++k) { h1[k] = hs[k * L::PLANE + i]; a2[k] = wimg[S::B2 + k]; }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const float* wl = wimg + S::W2 + k * C;
#pragma unroll
                for (int co = 0; co < C; ++co) a2[co] = fmaf(h1[k], wl[co], a2[co]);
            }
            asm volatile("" ::: "memory");
            float d1[C];
#pragma unroll
            for (int k = 0; k < C; ++k) d1[k] = 0.f;
#pragma unroll
            for (int co = 0; co < C; ++co) {
                const float gd = ds[co * L::PLANE + i] * elu_grad_from_out(elu1(a2[co]));     // dA2
                const float gc = gd * centre;
                ab2[co] += gc;
                const float* wl = wimg + S::W2T + co * C;
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    d1[k] = fmaf(gd, wl[k], d1[k]);
                    aw2[co][k] = fmaf(gc, h1[k], aw2[co][k]);
                }
            }
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const float gd = d1[k] * elu_grad_from_out(h1[k]);
                ab1[k] = fmaf(gd, centre, ab1[k]);
                hs[k * L::PLANE + i] = gd;
            }
        }
        __syncthreads();
        // ---- phase 2: dW1 on the matrix cores (k = 4 sk + g within the wave's rows), then the data gradient on the vector ALUs ----
        auto wgrad_steps = [&](int sk0, int nsk) {
#pragma unroll
            for (int rr = 0; rr < RPT; ++rr)
#pragma unroll
                for (int s_ = 0; s_ < nsk; ++s_) {
                    const int sk = sk0 + s_;
                    float av[L::MT], bv[L::MT];
#pragma unroll
                    for (int mt = 0; mt < L::MT; ++mt) {
                        av[mt] = aok[mt] ? hs[aoff[mt] + rr * NW * 72 + 4 * sk] : 0.f;
                        bv[mt] = xs[boff[mt] + rr * NW * 64 + 4 * sk];
                    }
#pragma unroll
                    for (int nt = 0; nt < L::MT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < L::MT; ++mt)
                            wacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], wacc[mt][nt], 0, 0, 0);
                }
        };
        {
            int tt = xcd_tile_s(v, ntiles);
            const int tx = tt % tiles_t; tt /= tiles_t;
            const int ty = tt % tiles_h;
            const int b = tt / tiles_h, h0 = ty * L::TR + wave, t = tx * 64 + lane;
            const float* xt = hs + wave * 72 + (4 - D) + lane;          // tap (kh, kw) of row rr: + (NW rr + kh D) 72 + kw D
            float acc[RPT][C];
#pragma unroll
            for (int r = 0; r < RPT; ++r)
#pragma unroll
                for (int k = 0; k < C; ++k) acc[r][k] = ds[k * L::PLANE + (D + wave + NW * r) * 72 + 4 + lane];
            // (issuing the k-steps inside the tap loop below, to run the matrix pipe in the shadow of the FMA stream, measured
            //  4 % SLOWER at C = 8 and is not used)
            if (!(ablate & 2)) wgrad_steps(0, 16);
#pragma unroll 1
            for (int co = 0; co < ((ablate & 4) ? 0 : C); ++co) {
                const float* xc = xt + co * L::PLANE;
                const float* wc = wimg + S::W1 + co * 9 * C;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float* wl = wc + (kh * 3 + kw) * C;
                        float w[C];
#pragma unroll
                        for (int k = 0; k < C; ++k) w[k] = wl[k];
#pragma unroll
                        for (int r = 0; r < RPT; ++r) {
                            const float xv = xc[(NW * r + kh * D) * 72 + kw * D];
#pragma unroll
                            for (int k = 0; k < C; ++k) acc[r][k] = fmaf(xv, w[k], acc[r][k]);
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int h = h0 + NW * r;
                if (h < H && t < T) {
                    const long o = (long)b * C * plane + (long)h * T + t;
#pragma unroll
                    for (int k = 0; k < C; ++k) dx[o + k * plane] = acc[r][k];
                }
            }
        }
        if (NBUF == 2) buf ^= 1;
    }
    // ---- reductions: dW1 partial image per workgroup; db1, db2, dW2 -> LDS -> one global atomic per element ----
    __syncthreads();
    float* red = lds_dyn;                                  // [IMG] image, then [C*C] dW2, [C] db1, [C] db2
    for (int i = threadIdx.x; i < L::IMG + C * C + 2 * C; i += L::NT) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < L::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < L::MT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&red[(mt * 16 + 4 * g + r) * L::NC + nt * 16 + l15], wacc[mt][nt][r]);
#pragma unroll
    for (int a = 0; a < C; ++a) {
        const float s1 = wave_sum(ab1[a]), s2 = wave_sum(ab2[a]);
        if (lane == 0) { atomicAdd(&red[L::IMG + C * C + a], s1); atomicAdd(&red[L::IMG + C * C + C + a], s2); }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float s = wave_sum(aw2[a][c]);
            if (lane == 0) atomicAdd(&red[L::IMG + a * C + c], s);
        }
    }
    __syncthreads();
    float* part = scratch + (long)blockIdx.x * L::IMG;
    for (int i = threadIdx.x; i < L::IMG; i += L::NT) part[i] = red[i];
    if (threadIdx.x < C * C) atomicAdd(dw2 + threadIdx.x, red[L::IMG + threadIdx.x]);
    if (threadIdx.x < C) {
        atomicAdd(db1 + threadIdx.x, red[L::IMG + C * C + threadIdx.x]);
        atomicAdd(db2 + threadIdx.x, red[L::IMG + C * C + C + threadIdx.x]);
    }
}

// dw[co][ci][kh][kw] += sum over workgroups of partial[(kw*C + co)][(kh*C + ci)]   (the layout of k_wgrad3_pack's images)
template <int C>
__global__ __launch_bounds__(256) void k_small_wgrad_reduce(const float* __restrict__ scratch, float* __restrict__ dw, int nblk) {
    constexpr int M = 3 * C, MT = (M + 15) / 16, NC = MT * 16, IMG = MT * 16 * NC;
    __shared__ float red[8][33];
    const int e = threadIdx.x & 31, pg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;                     // element of the C*C*9 gradient
    const bool ok = i < C * C * 9;
    float s0 = 0.f;
    if (ok) {
        const int kw = i % 3, kh = (i / 3) % 3, ci = (i / 9) % C, co = i / (9 * C);
        const float* p = scratch + (kw * C + co) * NC + kh * C + ci;
        for (int k = pg; k < nblk; k += 8) s0 += p[(long)k * IMG];
    }
    red[pg][e] = s0;
    __syncthreads();
    if (pg == 0 && ok)
        dw[i] += ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) + ((red[4][e] + red[5][e]) + (red[6][e] + red[7][e]));
}

inline bool fused_bwd_variant() { return getenv("TTRAP_SMALL_UNFUSED_BWD") == nullptr; }   // read per call: tests A/B the two paths in one process

template <int C, int D>
int launch_small_bwd_fused(const float* x, const float* h1, const float* dy, const float* w1, const float* w2, const float* b2,
                           float* dx, float* dw1, float* db1, float* dw2, float* db2, float* scratch, int B, int H, int T,
                           hipStream_t st) {
    // eight waves, two rows each, one staging set: two workgroups per CU at C = 4 (75 KB each), one at C = 8 (150 KB).
    // Measured alternative at C = 4: sixteen waves with both staging sets resident (next tile's DMA under this tile's
    // arithmetic) -- 0.94 ms against 0.89 ms per launch at the bench shape, so the kernel is not waiting on the DMA.
    constexpr int NW = 8, RPT = 2, NBUF = 1;
    using L = SF<C, D, RPT, NW, NBUF>;
    static_assert(L::LDS_BYTES <= 160 * 1024, "staging sets exceed the LDS");
    static AttrOnce attr;
    if (const int adev_ = attr.pending(); adev_ >= 0) {
        TT_HIP(hipFuncSetAttribute((const void*)k_small_bwd_fused<C, D, RPT, NW, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS_BYTES));
        attr.mark(adev_);
    }
    const int ntiles = B * ((H + L::TR - 1) / L::TR) * ((T + 63) / 64);
    int per_cu = (160 * 1024) / L::LDS_BYTES;
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    int grid = ntiles < tt_cus() * per_cu ? ntiles : tt_cus() * per_cu;
    if (grid > 512) grid = 512;                            // partial images in the caller's scratch (tt_wgrad_scratch_floats)
    const char* ab = tt_tune_set("TTRAP_FUSED_ABLATE") ? getenv("TTRAP_FUSED_ABLATE") : nullptr;          // measurement only: bit 0 / 1 / 2 / 3 = skip phase 1 / dW1 / dx / re-staging
    hipLaunchKernelGGL((k_small_bwd_fused<C, D, RPT, NW, NBUF>), dim3(grid), dim3(L::NT), L::LDS_BYTES, st, x, h1, dy, w1, w2, b2, dx, db1, dw2,
                       db2, scratch, B, H, T, ab ? atoi(ab) : 0);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_small_wgrad_reduce<C>), dim3((C * C * 9 + 31) / 32), dim3(256), 0, st, (const float*)scratch, dw1, grid);
    TT_LAUNCH_CHECK();
    return 0;
}

// the LDS-tiled kernels need LDS-DMA-able rows (T % 4 == 0, 16-byte aligned input); TTRAP_SMALL_GLOBAL=1 forces the
// thread-per-pixel kernels with global taps (kept for unaligned shapes and for A/B measurements)
inline bool lds_variant() { static const bool v = !tt_tune_set("TTRAP_SMALL_GLOBAL"); return v; }

inline bool small_mfma_variant() { return getenv("TTRAP_SMALL_VALU_FMA") == nullptr; }   // read per call (tests A/B both forms)

template <int C, int D, int MODE>
int launch_small_lds(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* res, float* y,
                     float* h1, int B, int H, int T, hipStream_t st) {
    constexpr int RPT = (C == 8 ? SMALL_RPT8 : SMALL_RPT), NBUF = (C == 8 && RPT == 2) ? 1 : 2;
    using L = SL<C, D, RPT, NBUF>;
    static AttrOnce attr;
    if (const int adev_ = attr.pending(); adev_ >= 0) {
        TT_HIP(hipFuncSetAttribute((const void*)k_small_lds<C, D, MODE, RPT, NBUF, false>, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS_BYTES));
        TT_HIP(hipFuncSetAttribute((const void*)k_small_lds<C, D, MODE, RPT, NBUF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS_BYTES));
        attr.mark(adev_);
    }
    const int ntiles = B * ((H + L::TR - 1) / L::TR) * ((T + 63) / 64);
    int per_cu = (160 * 1024) / L::LDS_BYTES;
    if (per_cu > 4) per_cu = 4;
    const int grid = ntiles < tt_cus() * per_cu ? ntiles : tt_cus() * per_cu;
    // Measured at the bench shapes (bit-identical results either way): C = 8 forward 0.48-0.53 ms on the matrix pipe against
    // 0.53-0.59 ms on the vector ALUs, C = 8 data gradient -0.07 ms at dilation 1 and 2 but +0.08 ms at dilation 3 (which
    // therefore keeps the vector form); C = 4 the same within noise (0.40 ms: that level is not arithmetic-bound) with a third
    // fewer registers.
    constexpr bool MF_OK = !(C == 8 && MODE == 1 && D == 3);
    if (MF_OK && small_mfma_variant())
        hipLaunchKernelGGL((k_small_lds<C, D, MODE, RPT, NBUF, true>), dim3(grid), dim3(512), L::LDS_BYTES, st, x, w1, b1, w2, b2, res, y, h1, B, H, T);
    else
        hipLaunchKernelGGL((k_small_lds<C, D, MODE, RPT, NBUF, false>), dim3(grid), dim3(512), L::LDS_BYTES, st, x, w1, b1, w2, b2, res, y, h1, B, H, T);
    TT_LAUNCH_CHECK();
    return 0;
}

template <int C, int D>
int fwd_t(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y, float* h1, int B, int H,
          int T, hipStream_t st) {
    if (lds_variant() && T % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        // four pixels per lane (k_small_fwd4; TTRAP_SMALL_FWD4=0: the one-pixel form, bit-identical) needs 16-byte aligned outputs too
        static const int four = tt_switch("TTRAP_SMALL_FWD4", 1);      // 0: never, 1: where it wins, 2: every shape (A/B)
        // taken where it wins (B 96 planes, ms per launch, one / four pixels per lane): C = 8 0.743 / 0.649, 0.658 / 0.571 at dilation 1, 2;
        // dilation 3 0.649 / 0.918 (its 22-row tile leaves room for two workgroups per CU only); C = 4 0.436 / 0.423, 0.433 / 0.417,
        // 0.440 / 0.507 -- the kernel is the SUM of staging, products and epilogue at three waves per SIMD, not bound by one of them
        if (four && (four == 2 || (C == 8 && D <= 2)) && small_mfma_variant() && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(h1)) & 15) == 0) {
            constexpr int TR4 = SMALL_FWD4_TR;
            using L = SL4<C, D, TR4>;
            static AttrOnce attr4;
            auto kern = k_small_fwd4<C, D, TR4>;
            if (const int dev_ = attr4.pending(); dev_ >= 0) {
                TT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS_BYTES));
                attr4.mark(dev_);
            }
            const int ntiles = B * ((H + L::TR - 1) / L::TR) * ((T + 63) / 64);
            int per_cu = (160 * 1024) / L::LDS_BYTES;
            if (per_cu > 8) per_cu = 8;
            const int grid = ntiles < tt_cus() * per_cu ? ntiles : tt_cus() * per_cu;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(L::NTH), L::LDS_BYTES, st, x, w1, b1, w2, b2, y, h1, B, H, T);
            TT_LAUNCH_CHECK();
            return 0;
        }
        return launch_small_lds<C, D, 0>(x, w1, b1, w2, b2, nullptr, y, h1, B, H, T, st);
    }
    dim3 grid((T + 63) / 64, (H + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, B);
    hipLaunchKernelGGL((k_small<C, D, 0>), grid, dim3(256), 0, st, x, w1, b1, w2, b2, (const float*)nullptr, y, h1, B, H, T);
    TT_LAUNCH_CHECK();
    return 0;
}

template <int C, int D>
int bwd_t(const float* x, const float* h1, const float* dy, const float* w1, const float* b1, const float* w2, const float* b2,
          float* dx, float* dw1, float* db1, float* dw2, float* db2, float* ws, float* scratch, int B, int H, int T,
          hipStream_t st) {
    const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    // measured at the bench shapes (64 clips): C = 4 fused 0.84-0.89 ms against 1.04-1.11 ms for the three kernels; at C = 8
    // one staging set fills the LDS (one workgroup of eight waves per CU, nothing to hide the phases behind) and the two
    // paths tie (1.23-1.34 against 1.23-1.25 ms), so C = 8 stays on the three-kernel path unless TTRAP_SMALL_FUSED_BWD_C8 is set
    const bool want_fused = C <= 4 ? fused_bwd_variant() : (fused_bwd_variant() && getenv("TTRAP_SMALL_FUSED_BWD_C8") != nullptr);
    if (h1 && want_fused && lds_variant() && T % 4 == 0 && al16(x) && al16(h1) && al16(dy)) {
        const int rc = launch_small_bwd_fused<C, D>(x, h1, dy, w1, w2, b2, dx, dw1, db1, dw2, db2, scratch, B, H, T, st);
        return rc ? rc : TT_SMALL_BWD_DID_DW1;
    }
    const int ntiles = B * ((H + 3) / 4) * ((T + 63) / 64);
    const int pgrid = ntiles < 8 * tt_cus() ? ntiles : 8 * tt_cus();
    if (h1)
        hipLaunchKernelGGL((k_small_bwd_a<C, D, false>), dim3(pgrid), dim3(256), 0, st, x, h1, dy, w1, b1, w2, b2, ws, db1, dw2,
                           db2, B, H, T);
    else
        hipLaunchKernelGGL((k_small_bwd_a<C, D, true>), dim3(pgrid), dim3(256), 0, st, x, h1, dy, w1, b1, w2, b2, ws, db1, dw2,
                           db2, B, H, T);
    TT_LAUNCH_CHECK();
    if (lds_variant() && T % 4 == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0)
        return launch_small_lds<C, D, 1>(ws, w1, nullptr, nullptr, nullptr, dy, dx, nullptr, B, H, T, st);
    dim3 grid((T + 63) / 64, (H + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, B);
    hipLaunchKernelGGL((k_small<C, D, 1>), grid, dim3(256), 0, st, (const float*)ws, w1, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, dy, dx, (float*)nullptr, B, H, T);
    TT_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int tt_small_rb_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y, float* h1,
                    int B, int C, int H, int T, int dilation, hipStream_t st) {
    switch (C * 10 + dilation) {
        case 41: return fwd_t<4, 1>(x, w1, b1, w2, b2, y, h1, B, H, T, st);
        case 42: return fwd_t<4, 2>(x, w1, b1, w2, b2, y, h1, B, H, T, st);
        case 43: return fwd_t<4, 3>(x, w1, b1, w2, b2, y, h1, B, H, T, st);
        case 81: return fwd_t<8, 1>(x, w1, b1, w2, b2, y, h1, B, H, T, st);
        case 82: return fwd_t<8, 2>(x, w1, b1, w2, b2, y, h1, B, H, T, st);
        case 83: return fwd_t<8, 3>(x, w1, b1, w2, b2, y, h1, B, H, T, st);
        default: return TT_E_UNSUPPORTED;
    }
}

int tt_small_rb_bwd(const float* x, const float* h1, const float* dy, const float* w1, const float* b1, const float* w2,
                    const float* b2, float* dx, float* dw1, float* db1, float* dw2, float* db2, float* ws, float* scratch, int B, int C, int H,
                    int T, int dilation, hipStream_t st) {
#define TT_SB(CC, DD) bwd_t<CC, DD>(x, h1, dy, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, ws, scratch, B, H, T, st)
    switch (C * 10 + dilation) {
        case 41: return TT_SB(4, 1);
        case 42: return TT_SB(4, 2);
        case 43: return TT_SB(4, 3);
        case 81: return TT_SB(8, 1);
        case 82: return TT_SB(8, 2);
        case 83: return TT_SB(8, 3);
        default: return TT_E_UNSUPPORTED;
    }
#undef TT_SB
}
