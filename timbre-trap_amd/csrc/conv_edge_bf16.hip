// The two 3x3 boundary convolutions of the autoencoder where fp32 planar tensors meet the bf16 channels-last interior
// (reference modules.py:433 Encoder.convin: Conv2d(2, C0, 3, 'same') + ELU;  :560 Decoder.convout: Conv2d(C0, 2, 3, 'same')),
// C0 = 4 (model_complexity 2):
//   k_cin_fwd    coefficients (B,2,H,T) fp32 planar -> ELU(conv + b) as cl16 (B,4,H,T)
//   k_cin_bwd    g = dy * ELU'(y) on the fly; dW (4,2,3,3), db; optionally dx (B,2,H,T) fp32 planar (the re-encoded transcription)
//   k_cout_fwd   cl16 (B,4,H,T) -> logits (B,2,H,T) fp32 planar = conv + b
//   k_cout_bwd   dy (B,2,H,T) fp32 planar -> dx cl16 (B,4,H,T); dW (2,4,3,3), db
// With 2 x 4 channels there is nothing for the matrix cores: a lane is a pixel, the 72 weights are wave-uniform scalars, the
// tile (16 rows x 64 frames + 1 halo) sits in LDS as fp32, and the arithmetic is fp32 throughout (only the cl16 tensors
// are bf16).  Weight / bias gradients: per-lane accumulators over the lane's pixels, reduced over the wave by shuffles and
// over the workgroup through LDS once at the end; one partial per workgroup, summed by k_edge_reduce.
#include "bf16_common.h"

namespace {

constexpr int ETH = 16, ETW = 64, ERW = ETW + 2, EROWS = ETH + 2;      // tile and its halo
constexpr int EPLANE = EROWS * ERW;
constexpr int NPARTW = 80;                                             // 72 weight + up to 4 bias partials, padded

__device__ __forceinline__ float gatef(float dy, float y) { return dy * elu_dout(y); }

struct ETile { int b, h0, t0; };
__device__ __forceinline__ ETile etile(int v, int tiles_h, int tiles_t, int ntiles) {
    int tile = xcd_order(v, ntiles);
    ETile r;
    r.t0 = (tile % tiles_t) * ETW; tile /= tiles_t;
    r.h0 = (tile % tiles_h) * ETH;
    r.b = tile / tiles_h;
    return r;
}

// planes [NP][EROWS][ERW] fp32 <- NP consecutive planes of a planar (B,NP,H,T) tensor, zero outside the image.
// All requests of the tile are issued first (unrolled, clamped addresses, no branches), the LDS writes follow: a loop of
// load-then-store pairs would pay one memory latency per iteration.
template <int NP>
__device__ __forceinline__ void stage_planar(float* lds, const float* src, int b, int h0, int t0, int H, int T, int tid) {
    constexpr int NIT = (NP * EPLANE + NT - 1) / NT;
    float v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * NT + tid;
        const int pl = i / EPLANE, rem = i - pl * EPLANE;
        const int row = rem / ERW, col = rem - row * ERW;
        const int h = h0 - 1 + row, t = t0 - 1 + col;
        const bool ok = i < NP * EPLANE && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
        const float q = src[ok ? (((long)b * NP + pl) * H + h) * T + t : 0];
        v[it] = ok ? q : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * NT + tid;
        if (i < NP * EPLANE) lds[i] = v[it];
    }
}
// planes [4][EROWS][ERW] fp32 <- cl16 (B,H,T,4), optionally gated by the saved output
template <bool GATE>
__device__ __forceinline__ void stage_cl4(float* lds, const __bf16* src, const __bf16* ysrc, int b, int h0, int t0, int H, int T, int tid) {
    constexpr int NIT = (EPLANE + NT - 1) / NT;
    bf16x4 q[NIT], yq[GATE ? NIT : 1];
    bool okv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * NT + tid;
        const int row = i / ERW, col = i - row * ERW;
        const int h = h0 - 1 + row, t = t0 - 1 + col;
        okv[it] = i < EPLANE && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
        const long off = okv[it] ? (((long)b * H + h) * T + t) * 4 : 0;
        q[it] = *reinterpret_cast<const bf16x4*>(src + off);
        if (GATE) yq[it] = *reinterpret_cast<const bf16x4*>(ysrc + off);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * NT + tid;
        if (i >= EPLANE) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = GATE ? gatef((float)q[it][c], (float)yq[it][c]) : (float)q[it][c];
            lds[c * EPLANE + i] = okv[it] ? v : 0.f;
        }
    }
}

// sum `acc[N]` over the lanes of the wave, then over the four waves; thread e < N of the workgroup ends with the total
template <int N>
__device__ __forceinline__ float wg_total(float (&acc)[N], float* red, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        float s = acc[e];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) red[wave * N + e] = s;
    }
    __syncthreads();
    return tid < N ? (red[tid] + red[N + tid]) + (red[2 * N + tid] + red[3 * N + tid]) : 0.f;
}

// ---- Encoder.convin ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_cin_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                 __bf16* __restrict__ y, int H, int T, int tiles_h, int tiles_t, int ntiles) {
    __shared__ float xs[2 * EPLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float wr[72], br[4];                                         // w[co][ci][kh][kw]: uniform
#pragma unroll
    for (int i = 0; i < 72; ++i) { wr[i] = w[i]; asm volatile("" : "+v"(wr[i])); }   // pinned in VGPRs: as wave-uniform values they
                                                                                 // were spilled to VGPR lanes and read back with one v_readlane per use
#pragma unroll
    for (int c = 0; c < 4; ++c) br[c] = bias[c];
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        const ETile tl = etile(v, tiles_h, tiles_t, ntiles);
        __syncthreads();
        stage_planar<2>(xs, x, tl.b, tl.h0, tl.t0, H, T, tid);
        __syncthreads();
        const int t = tl.t0 + lane;
        for (int r = wave; r < ETH; r += 4) {
            const int h = tl.h0 + r;
            if (h >= H) break;
            float acc[4] = {br[0], br[1], br[2], br[3]};
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float xv = xs[ci * EPLANE + (r + k / 3) * ERW + lane + k % 3];
#pragma unroll
                    for (int co = 0; co < 4; ++co) acc[co] = fmaf(wr[(co * 2 + ci) * 9 + k], xv, acc[co]);
                }
            bf16x4 o;
#pragma unroll
            for (int co = 0; co < 4; ++co) o[co] = (__bf16)elu_f(acc[co]);
            if (t < T) *reinterpret_cast<bf16x4*>(y + (((long)tl.b * H + h) * T + t) * 4) = o;
        }
    }
}

// dW[co][ci][k] = sum g[co][p] x[ci][p + k];  db[co] = sum g[co];  dx[ci][p] = sum_{co,k} W[co][ci][k] g[co][p - k]
template <bool DX>
__global__ __launch_bounds__(NT) void k_cin_bwd(const float* __restrict__ x, const __bf16* __restrict__ y, const __bf16* __restrict__ dy,
                                                 const float* __restrict__ w, float* __restrict__ dx, float* __restrict__ part,
                                                 int H, int T, int tiles_h, int tiles_t, int ntiles) {
    __shared__ float xs[2 * EPLANE];
    __shared__ float gs[4 * EPLANE];
    __shared__ float red[4 * 76];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float wr[72];
#pragma unroll
    for (int i = 0; i < 72; ++i) wr[i] = w[i];
    float acc[76];                                               // 72 dW + 4 db
#pragma unroll
    for (int i = 0; i < 76; ++i) acc[i] = 0.f;
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        const ETile tl = etile(v, tiles_h, tiles_t, ntiles);
        __syncthreads();
        stage_planar<2>(xs, x, tl.b, tl.h0, tl.t0, H, T, tid);
        stage_cl4<true>(gs, dy, y, tl.b, tl.h0, tl.t0, H, T, tid);
        __syncthreads();
        const int t = tl.t0 + lane;
        for (int r = wave; r < ETH; r += 4) {
            const int h = tl.h0 + r;
            if (h >= H) break;
            const int ctr = (r + 1) * ERW + lane + 1;
            float g[4];
#pragma unroll
            for (int co = 0; co < 4; ++co) { g[co] = t < T ? gs[co * EPLANE + ctr] : 0.f; acc[72 + co] += g[co]; }
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float xv = xs[ci * EPLANE + (r + k / 3) * ERW + lane + k % 3];
#pragma unroll
                    for (int co = 0; co < 4; ++co) acc[(co * 2 + ci) * 9 + k] = fmaf(g[co], xv, acc[(co * 2 + ci) * 9 + k]);
                }
            if constexpr (DX) {
                float d[2] = {0.f, 0.f};
#pragma unroll
                for (int co = 0; co < 4; ++co)
#pragma unroll
                    for (int k = 0; k < 9; ++k) {                // g at p - (k - centre) = rows r + 2 - k/3, cols lane + 2 - k%3
                        const float gv = gs[co * EPLANE + (r + 2 - k / 3) * ERW + lane + 2 - k % 3];
#pragma unroll
                        for (int ci = 0; ci < 2; ++ci) d[ci] = fmaf(wr[(co * 2 + ci) * 9 + k], gv, d[ci]);
                    }
                if (t < T) {
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci) dx[(((long)tl.b * 2 + ci) * H + h) * T + t] = d[ci];
                }
            }
        }
    }
    __syncthreads();
    const float tot = wg_total<76>(acc, red, tid);
    if (tid < 76) part[(long)blockIdx.x * NPARTW + tid] = tot;
}

// ---- Decoder.convout -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_cout_fwd(const __bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                  float* __restrict__ y, int H, int T, int tiles_h, int tiles_t, int ntiles) {
    __shared__ float xs[4 * EPLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float wr[72], br[2];                                         // w[co][ci][kh][kw], co < 2, ci < 4
#pragma unroll
    for (int i = 0; i < 72; ++i) { wr[i] = w[i]; asm volatile("" : "+v"(wr[i])); }   // pinned in VGPRs: as wave-uniform values they
                                                                                 // were spilled to VGPR lanes and read back with one v_readlane per use
    br[0] = bias[0]; br[1] = bias[1];
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        const ETile tl = etile(v, tiles_h, tiles_t, ntiles);
        __syncthreads();
        stage_cl4<false>(xs, x, nullptr, tl.b, tl.h0, tl.t0, H, T, tid);
        __syncthreads();
        const int t = tl.t0 + lane;
        for (int r = wave; r < ETH; r += 4) {
            const int h = tl.h0 + r;
            if (h >= H) break;
            float acc[2] = {br[0], br[1]};
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float xv = xs[ci * EPLANE + (r + k / 3) * ERW + lane + k % 3];
                    acc[0] = fmaf(wr[ci * 9 + k], xv, acc[0]);
                    acc[1] = fmaf(wr[(4 + ci) * 9 + k], xv, acc[1]);
                }
            if (t < T) {
                y[(((long)tl.b * 2 + 0) * H + h) * T + t] = acc[0];
                y[(((long)tl.b * 2 + 1) * H + h) * T + t] = acc[1];
            }
        }
    }
}

__global__ __launch_bounds__(NT) void k_cout_bwd(const __bf16* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w,
                                                  __bf16* __restrict__ dx, float* __restrict__ part, int H, int T, int tiles_h,
                                                  int tiles_t, int ntiles) {
    __shared__ float xs[4 * EPLANE];
    __shared__ float gs[2 * EPLANE];
    __shared__ float red[4 * 74];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float wr[72];
#pragma unroll
    for (int i = 0; i < 72; ++i) wr[i] = w[i];
    float acc[74];                                               // 72 dW + 2 db
#pragma unroll
    for (int i = 0; i < 74; ++i) acc[i] = 0.f;
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        const ETile tl = etile(v, tiles_h, tiles_t, ntiles);
        __syncthreads();
        stage_cl4<false>(xs, x, nullptr, tl.b, tl.h0, tl.t0, H, T, tid);
        stage_planar<2>(gs, dy, tl.b, tl.h0, tl.t0, H, T, tid);
        __syncthreads();
        const int t = tl.t0 + lane;
        for (int r = wave; r < ETH; r += 4) {
            const int h = tl.h0 + r;
            if (h >= H) break;
            const int ctr = (r + 1) * ERW + lane + 1;
            float g[2];
#pragma unroll
            for (int co = 0; co < 2; ++co) { g[co] = t < T ? gs[co * EPLANE + ctr] : 0.f; acc[72 + co] += g[co]; }
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float xv = xs[ci * EPLANE + (r + k / 3) * ERW + lane + k % 3];
                    acc[ci * 9 + k] = fmaf(g[0], xv, acc[ci * 9 + k]);
                    acc[(4 + ci) * 9 + k] = fmaf(g[1], xv, acc[(4 + ci) * 9 + k]);
                }
            float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int co = 0; co < 2; ++co)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float gv = gs[co * EPLANE + (r + 2 - k / 3) * ERW + lane + 2 - k % 3];
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci) d[ci] = fmaf(wr[(co * 4 + ci) * 9 + k], gv, d[ci]);
                }
            bf16x4 o;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) o[ci] = (__bf16)d[ci];
            if (t < T) *reinterpret_cast<bf16x4*>(dx + (((long)tl.b * H + h) * T + t) * 4) = o;
        }
    }
    __syncthreads();
    const float tot = wg_total<74>(acc, red, tid);
    if (tid < 74) part[(long)blockIdx.x * NPARTW + tid] = tot;
}

// dw[e] += sum over workgroups (e < 72), db[e - 72] likewise; 1024 threads = 64 elements x 16 slices
__global__ __launch_bounds__(1024) void k_edge_reduce(const float* __restrict__ part, int nwg, float* __restrict__ dw, float* __restrict__ db,
                                                       int nb) {
    __shared__ float red[16][64];
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    float s = 0.f;
    if (e < 72 + nb)
        for (int j = sl; j < nwg; j += 16) s += part[(long)j * NPARTW + e];
    red[sl][el] = s;
    __syncthreads();
    if (sl != 0 || e >= 72 + nb) return;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += red[i][el];
    if (e < 72) dw[e] += sum; else db[e - 72] += sum;
}

constexpr int EDGE_MAX_WG = 1024;

inline bool edge_ok(int B, int H, int T) { return B > 0 && H > 0 && T > 0 && (long)H * T * 4 < (1l << 31); }
inline void edge_tiles(int B, int H, int T, int& th, int& tt, int& n) { th = (H + ETH - 1) / ETH; tt = (T + ETW - 1) / ETW; n = B * th * tt; }
inline int edge_grid(int ntiles) { const int cap = 4 * tt_cus(); const int g = ntiles < cap ? ntiles : cap; return g < EDGE_MAX_WG ? g : EDGE_MAX_WG; }

}  // namespace

extern "C" {

int64_t tt_edge16_scratch_bytes(void) { return (int64_t)EDGE_MAX_WG * NPARTW * 4; }

int tt_convin16_fwd(const float* x, const float* w, const float* b, void* y, int B, int H, int T, void* stream) {
    if (!x || !w || !b || !y || !edge_ok(B, H, T)) return TT_E_BADARG;
    int th, tt, n; edge_tiles(B, H, T, th, tt, n);
    hipLaunchKernelGGL(k_cin_fwd, dim3(edge_grid(n)), dim3(NT), 0, tt_stream(stream), x, w, b, (__bf16*)y, H, T, th, tt, n);
    TT_LAUNCH_CHECK();
    return 0;
}

int tt_convin16_bwd(const float* x, const void* y, const void* dy, const float* w, float* dx, float* dw, float* db, void* ws, int B,
                    int H, int T, void* stream) {
    if (!x || !y || !dy || !w || !dw || !db || !ws || !edge_ok(B, H, T)) return TT_E_BADARG;
    int th, tt, n; edge_tiles(B, H, T, th, tt, n);
    const int grid = edge_grid(n);
    hipStream_t st = tt_stream(stream);
    if (dx) hipLaunchKernelGGL(k_cin_bwd<true>, dim3(grid), dim3(NT), 0, st, x, (const __bf16*)y, (const __bf16*)dy, w, dx, (float*)ws, H, T, th, tt, n);
    else hipLaunchKernelGGL(k_cin_bwd<false>, dim3(grid), dim3(NT), 0, st, x, (const __bf16*)y, (const __bf16*)dy, w, dx, (float*)ws, H, T, th, tt, n);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_edge_reduce, dim3(2), dim3(1024), 0, st, (const float*)ws, grid, dw, db, 4);
    TT_LAUNCH_CHECK();
    return 0;
}

int tt_convout16_fwd(const void* x, const float* w, const float* b, float* y, int B, int H, int T, void* stream) {
    if (!x || !w || !b || !y || !edge_ok(B, H, T)) return TT_E_BADARG;
    int th, tt, n; edge_tiles(B, H, T, th, tt, n);
    hipLaunchKernelGGL(k_cout_fwd, dim3(edge_grid(n)), dim3(NT), 0, tt_stream(stream), (const __bf16*)x, w, b, y, H, T, th, tt, n);
    TT_LAUNCH_CHECK();
    return 0;
}

int tt_convout16_bwd(const void* x, const float* dy, const float* w, void* dx, float* dw, float* db, void* ws, int B, int H, int T,
                     void* stream) {
    if (!x || !dy || !w || !dx || !dw || !db || !ws || !edge_ok(B, H, T)) return TT_E_BADARG;
    int th, tt, n; edge_tiles(B, H, T, th, tt, n);
    const int grid = edge_grid(n);
    hipStream_t st = tt_stream(stream);
    hipLaunchKernelGGL(k_cout_bwd, dim3(grid), dim3(NT), 0, st, (const __bf16*)x, dy, w, (__bf16*)dx, (float*)ws, H, T, th, tt, n);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_edge_reduce, dim3(2), dim3(1024), 0, st, (const float*)ws, grid, dw, db, 2);
    TT_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
