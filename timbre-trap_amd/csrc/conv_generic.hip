// General direct 2-D convolution (every layer shape of the Timbre-Trap autoencoder and, with
// swapped weight strides, every data gradient), its weight/bias gradient, and the small pointwise
// helpers of the backward pass.  Shape-agnostic reference path of the library: the hot
// ResidualConv2dBlock has its own fused kernels in conv_mfma.hip / conv_small.hip.
//
// Replaces torch.nn.Conv2d / ConvTranspose2d (+ ELU) forward and backward as used by reference
// timbre_trap/framework/modules.py:431, :446, :534, :543, :628, :687, :746, :751.
#include "common.h"

namespace {

struct ConvP {
    const float* x; const float* w; const float* bias; const float* res; float* y;
    int B, Cin, Hin, T, Cout, Hout, KH, KW, sh, dh, dw, ph, pw, transposed;
    long ws_co, ws_ci, ws_kh, ws_kw;
    int act;
};

// thread = one output column t of one output row (b, ho); CO_T output channels in registers.
template <int CO_T>
__global__ __launch_bounds__(256) void k_conv_generic(ConvP p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int ho = blockIdx.y, b = blockIdx.z;
    const bool tv = t < p.T;
    const long xplane = (long)p.Hin * p.T;
    const float* xb = p.x + (long)b * p.Cin * xplane;
    const long yplane = (long)p.Hout * p.T;

    // tap range along H (uniform over the block)
    int kh0 = 0, khstep = 1;
    if (p.transposed) {
        khstep = p.sh;                              // dil_h == 1 for every transposed use
        kh0 = (ho + p.ph) % p.sh;
    }
    for (int co0 = 0; co0 < p.Cout; co0 += CO_T) {
        float acc[CO_T];
#pragma unroll
        for (int c = 0; c < CO_T; ++c) acc[c] = (p.bias && co0 + c < p.Cout) ? p.bias[co0 + c] : 0.f;
        for (int kh = kh0; kh < p.KH; kh += khstep) {
            int hi;
            if (p.transposed) {
                const int num = ho + p.ph - kh * p.dh;
                if (num < 0) continue;
                hi = num / p.sh;
                if (hi * p.sh != num) continue;
            } else {
                hi = ho * p.sh + kh * p.dh - p.ph;
            }
            if (hi < 0 || hi >= p.Hin) continue;
            for (int kw = 0; kw < p.KW; ++kw) {
                const int ti = t + kw * p.dw - p.pw;
                const bool v = tv && ti >= 0 && ti < p.T;
                const float* xp = xb + (long)hi * p.T + (v ? ti : 0);
                const float* wp = p.w + kh * p.ws_kh + kw * p.ws_kw + co0 * p.ws_co;
                for (int ci = 0; ci < p.Cin; ++ci) {
                    const float xv = v ? xp[ci * xplane] : 0.f;
#pragma unroll
                    for (int c = 0; c < CO_T; ++c)
                        if (co0 + c < p.Cout) acc[c] = fmaf(xv, wp[ci * p.ws_ci + c * p.ws_co], acc[c]);
                }
            }
        }
        if (tv) {
#pragma unroll
            for (int c = 0; c < CO_T; ++c) {
                if (co0 + c >= p.Cout) break;
                float v = acc[c];
                if (p.act == TT_ACT_ELU) v = elu1(v);
                const long o = ((long)b * p.Cout + co0 + c) * yplane + (long)ho * p.T + t;
                if (p.res) v += p.res[o];
                p.y[o] = v;
            }
        }
    }
}

// Fast path for the two 3x3 'same' convolutions at the ends of the autoencoder (Encoder.convin 2 -> C0,
// Decoder.convout C0 -> 2, reference modules.py:431 and :543) and their data gradients: HBM-bound, one thread per
// pixel, weights broadcast from an LDS image [ci][tap][co] built through the signed strides of the general interface.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_conv3x3_small(ConvP p) {
    __shared__ float wl[CIN * 9 * COUT + COUT];
    for (int i = threadIdx.x; i < CIN * 9 * COUT; i += 256) {
        const int co = i % COUT, tap = (i / COUT) % 9, ci = i / (9 * COUT);
        wl[i] = p.w[co * p.ws_co + ci * p.ws_ci + (tap / 3) * p.ws_kh + (tap % 3) * p.ws_kw];
    }
    if (threadIdx.x < COUT) wl[CIN * 9 * COUT + threadIdx.x] = p.bias ? p.bias[threadIdx.x] : 0.f;
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tx, b = blockIdx.z;
    if (t >= p.T) return;
    const long plane = (long)p.Hin * p.T;
    const float* xb = p.x + (long)b * CIN * plane;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const int h = blockIdx.y * 16 + pass * 4 + ty;
        if (h >= p.Hin) break;
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = wl[CIN * 9 * COUT + co];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int hh = h + kh - 1;
                const bool hv = hh >= 0 && hh < p.Hin;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int tt = t + kw - 1;
                    float xv = 0.f;
                    if (hv && tt >= 0 && tt < p.T) xv = xb[ci * plane + (long)hh * p.T + tt];
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv, wl[(ci * 9 + kh * 3 + kw) * COUT + co], acc[co]);
                }
            }
        const long o = (long)b * COUT * plane + (long)h * p.T + t;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float v = acc[co];
            if (p.act == TT_ACT_ELU) v = elu1(v);
            if (p.res) v += p.res[o + co * plane];
            p.y[o + co * plane] = v;
        }
    }
}

struct WgradP {
    const float* x; const float* g; float* dw; float* dbias;
    int B, Cin, Hin, T, Cout, Hout, KH, KW, sh, dh, dw_, ph, pw;
    long ws_co, ws_ci, ws_kh, ws_kw;
    int n_ci_tiles, rows_per_chunk;
};

constexpr int TCO = 4, TCI = 2, KWMAX = 3;

// grid.x = co_tile * n_ci_tiles * KH + ..., grid.y = chunk of (b, ho) rows.
__global__ __launch_bounds__(256) void k_wgrad_generic(WgradP p) {
    __shared__ float red[4][TCO * TCI * KWMAX + TCO];
    int id = blockIdx.x;
    const int kh = id % p.KH; id /= p.KH;
    const int cit = id % p.n_ci_tiles; const int cot = id / p.n_ci_tiles;
    const int co0 = cot * TCO, ci0 = cit * TCI;
    const bool do_bias = p.dbias && cit == 0 && kh == 0;
    const long xplane = (long)p.Hin * p.T, gplane = (long)p.Hout * p.T;
    float acc[TCO][TCI][KWMAX];
    float bacc[TCO];
#pragma unroll
    for (int a = 0; a < TCO; ++a) {
        bacc[a] = 0.f;
#pragma unroll
        for (int c = 0; c < TCI; ++c)
#pragma unroll
            for (int k = 0; k < KWMAX; ++k) acc[a][c][k] = 0.f;
    }
    const long total_rows = (long)p.B * p.Hout;
    const long r0 = (long)blockIdx.y * p.rows_per_chunk;
    const long r1 = (r0 + p.rows_per_chunk < total_rows) ? r0 + p.rows_per_chunk : total_rows;
    for (long r = r0; r < r1; ++r) {
        const int b = (int)(r / p.Hout), ho = (int)(r - (long)b * p.Hout);
        const int hi = ho * p.sh + kh * p.dh - p.ph;
        const bool hv = hi >= 0 && hi < p.Hin;
        if (!hv && !do_bias) continue;
        const float* gb = p.g + ((long)b * p.Cout + co0) * gplane + (long)ho * p.T;
        const float* xb = p.x + ((long)b * p.Cin + ci0) * xplane + (long)(hv ? hi : 0) * p.T;
        for (int t = threadIdx.x; t < p.T; t += blockDim.x) {
            float gv[TCO];
#pragma unroll
            for (int a = 0; a < TCO; ++a) gv[a] = (co0 + a < p.Cout) ? gb[a * gplane + t] : 0.f;
            if (do_bias) {
#pragma unroll
                for (int a = 0; a < TCO; ++a) bacc[a] += gv[a];
            }
            if (!hv) continue;
#pragma unroll
            for (int k = 0; k < KWMAX; ++k) {
                if (k >= p.KW) break;
                const int ti = t + k * p.dw_ - p.pw;
                if (ti < 0 || ti >= p.T) continue;
#pragma unroll
                for (int c = 0; c < TCI; ++c) {
                    if (ci0 + c >= p.Cin) break;
                    const float xv = xb[c * xplane + ti];
#pragma unroll
                    for (int a = 0; a < TCO; ++a) acc[a][c][k] = fmaf(gv[a], xv, acc[a][c][k]);
                }
            }
        }
    }
    // block reduction: wave shuffle, then across the 4 waves through LDS
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int a = 0; a < TCO; ++a) {
#pragma unroll
        for (int c = 0; c < TCI; ++c)
#pragma unroll
            for (int k = 0; k < KWMAX; ++k) {
                const float s = wave_sum(acc[a][c][k]);
                if (lane == 0) red[wave][(a * TCI + c) * KWMAX + k] = s;
            }
        const float sb = wave_sum(bacc[a]);
        if (lane == 0) red[wave][TCO * TCI * KWMAX + a] = sb;
    }
    __syncthreads();
    const int nw = blockDim.x >> 6;
    if (threadIdx.x < TCO * TCI * KWMAX) {
        const int i = threadIdx.x;
        const int k = i % KWMAX, c = (i / KWMAX) % TCI, a = i / (KWMAX * TCI);
        if (k < p.KW && co0 + a < p.Cout && ci0 + c < p.Cin) {
            float s = 0.f;
            for (int w = 0; w < nw; ++w) s += red[w][i];
            atomicAdd(p.dw + (co0 + a) * p.ws_co + (ci0 + c) * p.ws_ci + kh * p.ws_kh + k * p.ws_kw, s);
        }
    } else if (do_bias && threadIdx.x < TCO * TCI * KWMAX + TCO) {
        const int a = threadIdx.x - TCO * TCI * KWMAX;
        if (co0 + a < p.Cout) {
            float s = 0.f;
            for (int w = 0; w < nw; ++w) s += red[w][TCO * TCI * KWMAX + a];
            atomicAdd(p.dbias + co0 + a, s);
        }
    }
}

__global__ __launch_bounds__(256) void k_elu_bwd(const float* __restrict__ dy, const float* __restrict__ y,
                                                 float* __restrict__ g, long n) {
    const long n4 = n >> 2;
    const float4* dy4 = reinterpret_cast<const float4*>(dy);
    const float4* y4 = reinterpret_cast<const float4*>(y);
    float4* g4 = reinterpret_cast<float4*>(g);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 a = dy4[i], b = y4[i];
        g4[i] = make_float4(a.x * elu_grad_from_out(b.x), a.y * elu_grad_from_out(b.y),
                            a.z * elu_grad_from_out(b.z), a.w * elu_grad_from_out(b.w));
    }
    if (blockIdx.x == 0) {
        const long i = (n4 << 2) + threadIdx.x;
        if (i < n) g[i] = dy[i] * elu_grad_from_out(y[i]);
    }
}

// out[c] += sum_{b, inner} x[b][c][inner] ; grid (C, chunks)
__global__ __launch_bounds__(256) void k_channel_sum(const float* __restrict__ x, float* __restrict__ out, int B, int C,
                                                     long inner) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float acc = 0.f;
    const long per_b = inner;
    const long total = (long)B * per_b;
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long)gridDim.y * 256) {
        const long b = i / per_b, r = i - b * per_b;
        acc += x[(b * C + c) * inner + r];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + c, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_scaled_add(const float* __restrict__ a, const float* __restrict__ b,
                                                    const float* __restrict__ s, int idx, float* __restrict__ y, long n) {
    const float sc = s ? s[idx] : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = fmaf(sc, b[i], a ? a[i] : 0.f);
}

__global__ __launch_bounds__(256) void k_dot(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                             long n) {
    __shared__ float red[4];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc = fmaf(a[i], b[i], acc);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// The same two on 16-bit tensors (the skip joins of the channels-last path: elementwise, so the layout does not matter): E = bf16, or
// fp16 for the _h entry points; eight elements per thread as one 16-byte access, fp32 arithmetic, round-to-nearest-even on the store.
template <class E>
__global__ __launch_bounds__(256) void k_scaled_add16(const E* __restrict__ a, const E* __restrict__ b,
                                                      const float* __restrict__ s, int idx, E* __restrict__ y, long n8, long n) {
    typedef E e8 __attribute__((ext_vector_type(8)));
    const float sc = s ? s[idx] : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const e8 bv = *reinterpret_cast<const e8*>(b + 8 * i);
        e8 av, o;
        if (a) av = *reinterpret_cast<const e8*>(a + 8 * i);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (E)fmaf(sc, (float)bv[j], a ? (float)av[j] : 0.f);
        *reinterpret_cast<e8*>(y + 8 * i) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {              // tail of a length that is not a multiple of 8
        const long i = (n & ~7l) + threadIdx.x;
        y[i] = (E)fmaf(sc, (float)b[i], a ? (float)a[i] : 0.f);
    }
}

template <class E>
__global__ __launch_bounds__(256) void k_dot16(const E* __restrict__ a, const E* __restrict__ b, float* __restrict__ out,
                                               long n8, long n, float scale) {
    typedef E e8 __attribute__((ext_vector_type(8)));
    __shared__ float red[4];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const e8 av = *reinterpret_cast<const e8*>(a + 8 * i), bv = *reinterpret_cast<const e8*>(b + 8 * i);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf((float)av[j], (float)bv[j], acc);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const long i = (n & ~7l) + threadIdx.x;
        acc = fmaf((float)a[i], (float)b[i], acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1] + red[2] + red[3]) * scale);
}

// The weighted skip join of the 16-bit training path as ONE pass each way (round 6; reference modules.py:112 `w_i * e_i` and
// :569-589 `y + skip`): out[r][i] = y[r][i] + s[idx] * e[i] for r < reps -- reps = 2 when the decoder runs the reconstruction and the
// transcription decode of the same latents as one batch of 2 B clips (TimbreTrap.decode_pair): both halves take the SAME encoder embedding,
// which is read once per element here and never duplicated in memory.  Eight elements per thread step as one 16-byte access, fp32
// arithmetic, one rounding per stored element.
template <class E, int REPS>
__global__ __launch_bounds__(256) void k_skip_join_fwd(const E* __restrict__ y, const E* __restrict__ e, const float* __restrict__ s, int idx,
                                                       E* __restrict__ out, long n8) {
    typedef E e8 __attribute__((ext_vector_type(8)));
    const float sc = s ? s[idx] : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const e8 ev = reinterpret_cast<const e8*>(e)[i];
        e8 yv[REPS];
#pragma unroll
        for (int r = 0; r < REPS; ++r) yv[r] = reinterpret_cast<const e8*>(y)[r * n8 + i];
#pragma unroll
        for (int r = 0; r < REPS; ++r) {
            e8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (E)fmaf(sc, (float)ev[j], (float)yv[r][j]);
            reinterpret_cast<e8*>(out)[r * n8 + i] = o;
        }
    }
}

// Its backward: t = sum_r g[r][i] (fp32);  de[i] = s[idx] * t, times ELU'(e[i]) when GATE (e is the output of a strided layer + ELU whose
// backward takes its gradient already gated -- ops.GateLink: the factor every OTHER contribution to that gradient carries);
// ds[idx] += unscale * sum_i t * e[i] (the gradient of the skip weight: leaves the 16-bit region, the fp16 loss scale comes off).
// dy = g is not written: the caller hands g itself on.  ACC: de += instead of = (de already holds the OTHER contribution to the embedding's
// gradient -- the data gradient of the encoder level behind it, gated the same way -- so that autograd has nothing left to add).
template <class E, int REPS, bool GATE, bool ACC>
__global__ __launch_bounds__(256) void k_skip_join_bwd(const E* __restrict__ g, const E* __restrict__ e, const float* __restrict__ s, int idx,
                                                       E* __restrict__ de, float* __restrict__ ds, long n8, float unscale) {
    typedef E e8 __attribute__((ext_vector_type(8)));
    __shared__ float red[4];
    const float sc = s ? s[idx] : 1.f;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const e8 ev = reinterpret_cast<const e8*>(e)[i];
        e8 gv[REPS];
#pragma unroll
        for (int r = 0; r < REPS; ++r) gv[r] = reinterpret_cast<const e8*>(g)[r * n8 + i];
        e8 o;
        if (ACC) o = reinterpret_cast<const e8*>(de)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = (float)gv[0][j];
#pragma unroll
            for (int r = 1; r < REPS; ++r) t += (float)gv[r][j];
            const float x = (float)ev[j];
            acc = fmaf(t, x, acc);
            float d = sc * t;
            if (GATE) d *= __builtin_fminf(x + 1.f, 1.f);             // ELU'(pre-activation) from the ELU's output (bf16_common.h: elu_dout)
            o[j] = ACC ? (E)((float)o[j] + d) : (E)d;
        }
        if (de) reinterpret_cast<e8*>(de)[i] = o;
    }
    if (ds) {
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(ds + idx, (red[0] + red[1] + red[2] + red[3]) * unscale);
    }
}

// ---- LDS-tiled 3x3 boundary convs (2 <-> 4 channels, stride 1, pad 1) ------------------------------------------------
// Same scheme as k_small_lds (conv_small.hip): the CI-channel input tile (16 rows + 2, 64 columns + 8) arrives by
// LDS-DMA, double buffered; a thread computes two output pixels (rows r, r + 8) for all CO channels from LDS taps with
// immediate offsets; generic weight strides cover the forward and the flipped data-gradient form.
__device__ float4 g_zero16_gen;
template <int CI, int CO>
struct BL {
    static constexpr int XR = 18, XCP = 72, PLANE = XR * XCP;
    static constexpr int NQ = CI * PLANE / 4, NP = (NQ + 63) / 64, BUF = NP * 256;
    static constexpr int W_FLOATS = CI * 9 * CO + CO;
    static constexpr int LDS_BYTES = (2 * BUF + W_FLOATS) * 4;
};
template <int CI, int CO>
__global__ __launch_bounds__(512) void k_conv3x3_lds(ConvP p) {
    using L = BL<CI, CO>;
    extern __shared__ __attribute__((aligned(16))) float lds_b[];
    float* xs = lds_b;
    float* wimg = lds_b + 2 * L::BUF;                      // [ci][tap][co], then bias
    for (int i = threadIdx.x; i < CI * 9 * CO; i += 512) {
        const int co = i % CO, tap = (i / CO) % 9, ci = i / (9 * CO);
        wimg[i] = p.w[co * p.ws_co + ci * p.ws_ci + (tap / 3) * p.ws_kh + (tap % 3) * p.ws_kw];
    }
    for (int i = threadIdx.x; i < CO; i += 512) wimg[CI * 9 * CO + i] = p.bias ? p.bias[i] : 0.f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = p.Hin, T = p.T;
    const int tiles_h = (H + 15) / 16, tiles_t = (T + 63) / 64;
    const int ntiles = p.B * tiles_h * tiles_t;
    const long plane = (long)H * T;
    const float* zero = reinterpret_cast<const float*>(&g_zero16_gen);
    auto coords = [&](int v, int& b, int& h0, int& t0) {
        const int per = ntiles >> 3;
        int tt = v < (per << 3) ? (v & 7) * per + (v >> 3) : v;        // XCD-contiguous tile order
        t0 = (tt % tiles_t) * 64; tt /= tiles_t;
        h0 = (tt % tiles_h) * 16; b = tt / tiles_h;
    };
    auto issue = [&](int v, int buf) {
        int b, h0, t0;
        coords(v, b, h0, t0);
        const float* xb = p.x + (long)b * CI * plane;
        float* dst = xs + buf * L::BUF;
#pragma unroll
        for (int jj = 0; jj < (L::NP + 7) / 8; ++jj) {
            const int j = wave + 8 * jj;
            if (j < L::NP) {
                const int q = j * 64 + lane;
                const int ci = q / (L::PLANE / 4);
                const int rem = q - ci * (L::PLANE / 4);
                const int r = rem / 18, c4 = rem - r * 18;
                const int h = h0 - 1 + r, t = t0 - 4 + 4 * c4;
                const bool ok = q < L::NQ && h >= 0 && h < H && t >= 0 && t < T;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ok ? xb + (ci * (int)plane + h * T + t) : zero),
                                                 (__attribute__((address_space(3))) void*)(dst + j * 256), 16, 0, 0);
            }
        }
    };
    int v = blockIdx.x;
    __syncthreads();
    if (v >= ntiles) return;
    int buf = 0;
    issue(v, 0);
    for (; v < ntiles; v += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (v + (int)gridDim.x < ntiles) issue(v + (int)gridDim.x, buf ^ 1);
        int b, h0, t0;
        coords(v, b, h0, t0);
        const float* xt = xs + buf * L::BUF + wave * L::XCP + 3 + lane;      // tap (kh, kw), row r: + ci*PLANE + (8 r + kh)*XCP + kw
        float acc[2][CO];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[r][co] = wimg[CI * 9 * CO + co];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float* wl = wimg + (ci * 9 + tap) * CO;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float xv = xt[ci * L::PLANE + (8 * r + tap / 3) * L::XCP + tap % 3];
#pragma unroll
                    for (int co = 0; co < CO; ++co) acc[r][co] = fmaf(xv, wl[co], acc[r][co]);
                }
            }
        const int t = t0 + lane;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int h = h0 + wave + 8 * r;
            if (h < H && t < T) {
                const long o = (long)b * CO * plane + (long)h * T + t;
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    float val = acc[r][co];
                    if (p.act == TT_ACT_ELU) val = elu1(val);
                    if (p.res) val += p.res[o + co * plane];
                    p.y[o + co * plane] = val;
                }
            }
        }
        buf ^= 1;
    }
}

template <int CI, int CO>
int launch_conv3x3_lds(const ConvP& p, hipStream_t st) {
    using L = BL<CI, CO>;
    static AttrOnce attr;
    if (const int adev_ = attr.pending(); adev_ >= 0) {
        TT_HIP(hipFuncSetAttribute((const void*)k_conv3x3_lds<CI, CO>, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS_BYTES));
        attr.mark(adev_);
    }
    const int ntiles = p.B * ((p.Hin + 15) / 16) * ((p.T + 63) / 64);
    int per_cu = (160 * 1024) / L::LDS_BYTES;
    if (per_cu > 4) per_cu = 4;
    const int grid = ntiles < tt_cus() * per_cu ? ntiles : tt_cus() * per_cu;
    hipLaunchKernelGGL((k_conv3x3_lds<CI, CO>), dim3(grid), dim3(512), L::LDS_BYTES, st, p);
    TT_LAUNCH_CHECK();
    return 0;
}

inline int grid1d(long n, int per_block, int cap) {
    long g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int tt_conv2d(const float* x, const float* w, const float* bias, const float* res, float* y,
                         int B, int Cin, int Hin, int T, int Cout, int Hout,
                         int KH, int KW, int stride_h, int dil_h, int dil_w, int pad_h, int pad_w,
                         int transposed, int64_t ws_co, int64_t ws_ci, int64_t ws_kh, int64_t ws_kw,
                         int act, void* stream) {
    if (!x || !w || !y || B <= 0 || Cin <= 0 || Cout <= 0 || Hin <= 0 || Hout <= 0 || T <= 0 || stride_h <= 0)
        return TT_E_BADARG;
    if (transposed && dil_h != 1) return TT_E_UNSUPPORTED;
    if (Hout > 65535 || B > 65535) return TT_E_UNSUPPORTED;
    ConvP p{x, w, bias, res, y, B, Cin, Hin, T, Cout, Hout, KH, KW, stride_h, dil_h, dil_w, pad_h, pad_w,
            transposed, (long)ws_co, (long)ws_ci, (long)ws_kh, (long)ws_kw, act};
    if (KH == 3 && KW == 3 && stride_h == 1 && dil_h == 1 && dil_w == 1 && pad_h == 1 && pad_w == 1 && !transposed &&
        Hin == Hout && (Cin == 2 || Cin == 4) && (Cout == 2 || Cout == 4)) {
        if (T % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (long)Cin * Hin * T < (1L << 31) && (long)Cout * Hin * T < (1L << 31)) {
            if (Cin == 2 && Cout == 4) return launch_conv3x3_lds<2, 4>(p, tt_stream(stream));
            if (Cin == 4 && Cout == 2) return launch_conv3x3_lds<4, 2>(p, tt_stream(stream));
        }
        dim3 g3((T + 63) / 64, (Hout + 15) / 16, B);
        if (Cin == 2 && Cout == 4) hipLaunchKernelGGL((k_conv3x3_small<2, 4>), g3, dim3(256), 0, tt_stream(stream), p);
        else if (Cin == 4 && Cout == 2) hipLaunchKernelGGL((k_conv3x3_small<4, 2>), g3, dim3(256), 0, tt_stream(stream), p);
        else if (Cin == 2 && Cout == 2) hipLaunchKernelGGL((k_conv3x3_small<2, 2>), g3, dim3(256), 0, tt_stream(stream), p);
        else hipLaunchKernelGGL((k_conv3x3_small<4, 4>), g3, dim3(256), 0, tt_stream(stream), p);
        TT_LAUNCH_CHECK();
        return 0;
    }
    const int tb = T >= 256 ? 256 : (T > 64 ? 128 : 64);
    dim3 grid((T + tb - 1) / tb, Hout, B);
    if (Cout >= 16)
        hipLaunchKernelGGL(k_conv_generic<16>, grid, dim3(tb), 0, tt_stream(stream), p);
    else if (Cout >= 8)
        hipLaunchKernelGGL(k_conv_generic<8>, grid, dim3(tb), 0, tt_stream(stream), p);
    else
        hipLaunchKernelGGL(k_conv_generic<4>, grid, dim3(tb), 0, tt_stream(stream), p);
    TT_LAUNCH_CHECK();
    return 0;
}

// 3x3 / stride 1 / pad 1 weight gradient of the two narrow boundary layers (Encoder.convin 2 -> C0, Decoder.convout
// C0 -> 2; reference modules.py:431,543): every thread walks pixels with all CO*CI*9 (<= 144) accumulators in registers,
// x and g are read ONCE (the generic kernel re-reads them per (co-tile, ci-tile, kh) column), one reduction per workgroup.
template <int CO, int CI>
__global__ __launch_bounds__(256) void k_wgrad3x3_small(WgradP p) {
    __shared__ float red[CO * CI * 9 + CO];
    for (int i = threadIdx.x; i < CO * CI * 9 + CO; i += 256) red[i] = 0.f;
    __syncthreads();
    float acc[CO][CI][9];
    float bacc[CO];
#pragma unroll
    for (int a = 0; a < CO; ++a) {
        bacc[a] = 0.f;
#pragma unroll
        for (int c = 0; c < CI; ++c)
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[a][c][k] = 0.f;
    }
    const int H = p.Hin, T = p.T;
    const long plane = (long)H * T;
    const int tiles_t = (T + 255) / 256;
    const long ntiles = (long)p.B * H * tiles_t;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tt = (int)(tile % tiles_t);
        const long rr = tile / tiles_t;
        const int h = (int)(rr % H), b = (int)(rr / H);
        const int t = tt * 256 + threadIdx.x;
        if (t >= T) continue;
        const float* gb = p.g + (long)b * CO * plane + (long)h * T + t;
        const float* xb = p.x + (long)b * CI * plane;
        float gv[CO];
#pragma unroll
        for (int a = 0; a < CO; ++a) { gv[a] = gb[a * plane]; bacc[a] += gv[a]; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = h + kh - 1;
            if (hh < 0 || hh >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ti = t + kw - 1;
                const bool ok = ti >= 0 && ti < T;
#pragma unroll
                for (int c = 0; c < CI; ++c) {
                    const float xv = ok ? xb[c * plane + (long)hh * T + ti] : 0.f;
#pragma unroll
                    for (int a = 0; a < CO; ++a) acc[a][c][kh * 3 + kw] = fmaf(gv[a], xv, acc[a][c][kh * 3 + kw]);
                }
            }
        }
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int a = 0; a < CO; ++a) {
#pragma unroll
        for (int c = 0; c < CI; ++c)
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float s_ = wave_sum(acc[a][c][k]);
                if (lane == 0) atomicAdd(&red[(a * CI + c) * 9 + k], s_);
            }
        const float sb = wave_sum(bacc[a]);
        if (lane == 0) atomicAdd(&red[CO * CI * 9 + a], sb);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CO * CI * 9; i += 256) {
        const int k = i % 9, c = (i / 9) % CI, a = i / (9 * CI);
        atomicAdd(p.dw + a * p.ws_co + c * p.ws_ci + (k / 3) * p.ws_kh + (k % 3) * p.ws_kw, red[i]);
    }
    if (p.dbias && threadIdx.x < CO) atomicAdd(p.dbias + threadIdx.x, red[CO * CI * 9 + threadIdx.x]);
}

extern "C" int tt_conv2d_wgrad(const float* x, const float* g, float* dw, float* dbias,
                               int B, int Cin, int Hin, int T, int Cout, int Hout,
                               int KH, int KW, int stride_h, int dil_h, int dil_w, int pad_h, int pad_w,
                               int64_t ws_co, int64_t ws_ci, int64_t ws_kh, int64_t ws_kw, void* stream) {
    if (!x || !g || !dw || B <= 0 || Cin <= 0 || Cout <= 0 || T <= 0) return TT_E_BADARG;
    if (KW > KWMAX) return TT_E_UNSUPPORTED;
    WgradP p{x, g, dw, dbias, B, Cin, Hin, T, Cout, Hout, KH, KW, stride_h, dil_h, dil_w, pad_h, pad_w,
             (long)ws_co, (long)ws_ci, (long)ws_kh, (long)ws_kw, 0, 0};
    if (KH == 3 && KW == 3 && stride_h == 1 && dil_h == 1 && dil_w == 1 && pad_h == 1 && pad_w == 1 && Hin == Hout &&
        ((Cin == 2 && Cout == 4) || (Cin == 4 && Cout == 2)) && (long)Cin * Hin * T < (1L << 31)) {
        const long ntiles = (long)B * Hin * ((T + 255) / 256);
        const int grid = (int)(ntiles < 8L * tt_cus() ? ntiles : 8L * tt_cus());
        if (Cin == 2) hipLaunchKernelGGL((k_wgrad3x3_small<4, 2>), dim3(grid), dim3(256), 0, tt_stream(stream), p);
        else hipLaunchKernelGGL((k_wgrad3x3_small<2, 4>), dim3(grid), dim3(256), 0, tt_stream(stream), p);
        TT_LAUNCH_CHECK();
        return 0;
    }
    p.n_ci_tiles = (Cin + TCI - 1) / TCI;
    const int n_co_tiles = (Cout + TCO - 1) / TCO;
    const long total_rows = (long)B * Hout;
    long chunks = 4096 / ((long)n_co_tiles * p.n_ci_tiles * KH);
    if (chunks < 8) chunks = 8;
    if (chunks > total_rows) chunks = total_rows;
    p.rows_per_chunk = (int)((total_rows + chunks - 1) / chunks);
    chunks = (total_rows + p.rows_per_chunk - 1) / p.rows_per_chunk;
    dim3 grid(n_co_tiles * p.n_ci_tiles * KH, (unsigned)chunks);
    const int tb = T >= 256 ? 256 : (T > 64 ? 128 : 64);
    hipLaunchKernelGGL(k_wgrad_generic, grid, dim3(tb), 0, tt_stream(stream), p);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_elu_bwd(const float* dy, const float* y, float* g, int64_t n, void* stream) {
    if (!dy || !y || !g || n < 0) return TT_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_elu_bwd, dim3(grid1d(n / 4 + 1, 256, 4096)), dim3(256), 0, tt_stream(stream), dy, y, g, (long)n);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_channel_sum(const float* x, float* out, int B, int C, int64_t inner, void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || inner <= 0) return TT_E_BADARG;
    int chunks = grid1d((long)B * inner, 256 * 8, 2048 / (C > 2048 ? 2048 : C) + 1);
    hipLaunchKernelGGL(k_channel_sum, dim3(C, chunks), dim3(256), 0, tt_stream(stream), x, out, B, C, (long)inner);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_scaled_add(const float* a, const float* b, const float* s, int idx, float* y, int64_t n,
                             void* stream) {
    if (!b || !y || n < 0) return TT_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_scaled_add, dim3(grid1d(n, 256, 4096)), dim3(256), 0, tt_stream(stream), a, b, s, idx, y, (long)n);
    TT_LAUNCH_CHECK();
    return 0;
}

template <class E>
static int scaled_add16(const void* a, const void* b, const float* s, int idx, void* y, int64_t n, void* stream) {
    if (!b || !y || n < 0) return TT_E_BADARG;
    if ((((uintptr_t)b | (uintptr_t)y | (uintptr_t)a) & 15) != 0) return TT_E_BADARG;        // 16-byte accesses
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_scaled_add16<E>, dim3(grid1d((n + 7) / 8, 256, 4096)), dim3(256), 0, tt_stream(stream), (const E*)a,
                       (const E*)b, s, idx, (E*)y, (long)(n / 8), (long)n);
    TT_LAUNCH_CHECK();
    return 0;
}
template <class E>
static int dot16(const void* a, const void* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n < 0) return TT_E_BADARG;
    if ((((uintptr_t)a | (uintptr_t)b) & 15) != 0) return TT_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_dot16<E>, dim3(grid1d((n + 7) / 8, 256 * 4, 1024)), dim3(256), 0, tt_stream(stream), (const E*)a,
                       (const E*)b, out, (long)(n / 8), (long)n, tt_loss_unscale());   // a = the S-scaled gradient (backward of the skip weights)
    TT_LAUNCH_CHECK();
    return 0;
}
template <class E>
static int skip_join16_fwd(const void* y, const void* e, const float* s, int idx, void* out, int64_t n, int reps, void* stream) {
    if (!y || !e || !out || n < 0 || n % 8 || (reps != 1 && reps != 2) || idx < 0) return TT_E_BADARG;
    if ((((uintptr_t)y | (uintptr_t)e | (uintptr_t)out) & 15) != 0) return TT_E_BADARG;        // 16-byte accesses
    if (n == 0) return 0;
    const long n8 = n / 8;
    const dim3 grid(grid1d(n8, 256, 8 * 256));
    if (reps == 1) hipLaunchKernelGGL((k_skip_join_fwd<E, 1>), grid, dim3(256), 0, tt_stream(stream), (const E*)y, (const E*)e, s, idx, (E*)out, n8);
    else hipLaunchKernelGGL((k_skip_join_fwd<E, 2>), grid, dim3(256), 0, tt_stream(stream), (const E*)y, (const E*)e, s, idx, (E*)out, n8);
    TT_LAUNCH_CHECK();
    return 0;
}
template <class E>
static int skip_join16_bwd(const void* g, const void* e, const float* s, int idx, void* de, float* ds, int64_t n, int reps, int gate,
                           void* stream) {
    const int acc = (gate >> 1) & 1;
    gate &= 1;
    if (!g || !e || n < 0 || n % 8 || (reps != 1 && reps != 2) || idx < 0 || (!de && !ds) || (acc && !de)) return TT_E_BADARG;
    if ((((uintptr_t)g | (uintptr_t)e | (uintptr_t)de) & 15) != 0) return TT_E_BADARG;
    if (n == 0) return 0;
    const long n8 = n / 8;
    const dim3 grid(grid1d(n8, 256, 8 * 256));
    const float un = tt_loss_unscale();
    hipStream_t st = tt_stream(stream);
#define TT_SJB(R, G, A) hipLaunchKernelGGL((k_skip_join_bwd<E, R, G, A>), grid, dim3(256), 0, st, (const E*)g, (const E*)e, s, idx, (E*)de, ds, n8, un)
#define TT_SJB2(R, G) do { if (acc) TT_SJB(R, G, true); else TT_SJB(R, G, false); } while (0)
    if (reps == 1) { if (gate) TT_SJB2(1, true); else TT_SJB2(1, false); }
    else           { if (gate) TT_SJB2(2, true); else TT_SJB2(2, false); }
#undef TT_SJB2
#undef TT_SJB
    TT_LAUNCH_CHECK();
    return 0;
}
extern "C" int tt_skip_join16_fwd(const void* y, const void* e, const float* s, int idx, void* out, int64_t n, int reps, void* stream) {
    return skip_join16_fwd<__bf16>(y, e, s, idx, out, n, reps, stream);
}
extern "C" int tt_skip_join16_fwd_h(const void* y, const void* e, const float* s, int idx, void* out, int64_t n, int reps, void* stream) {
    return skip_join16_fwd<_Float16>(y, e, s, idx, out, n, reps, stream);
}
extern "C" int tt_skip_join16_bwd(const void* g, const void* e, const float* s, int idx, void* de, float* ds, int64_t n, int reps, int gate,
                                  void* stream) {
    return skip_join16_bwd<__bf16>(g, e, s, idx, de, ds, n, reps, gate, stream);
}
extern "C" int tt_skip_join16_bwd_h(const void* g, const void* e, const float* s, int idx, void* de, float* ds, int64_t n, int reps, int gate,
                                    void* stream) {
    return skip_join16_bwd<_Float16>(g, e, s, idx, de, ds, n, reps, gate, stream);
}
extern "C" int tt_scaled_add16(const void* a, const void* b, const float* s, int idx, void* y, int64_t n, void* stream) {
    return scaled_add16<__bf16>(a, b, s, idx, y, n, stream);
}
extern "C" int tt_dot16(const void* a, const void* b, float* out, int64_t n, void* stream) { return dot16<__bf16>(a, b, out, n, stream); }
extern "C" int tt_scaled_add16_h(const void* a, const void* b, const float* s, int idx, void* y, int64_t n, void* stream) {
    return scaled_add16<_Float16>(a, b, s, idx, y, n, stream);
}
extern "C" int tt_dot16_h(const void* a, const void* b, float* out, int64_t n, void* stream) { return dot16<_Float16>(a, b, out, n, stream); }

// Windowed overlap-add of half-overlapping chunks (TimbreTrap.chunked_inference, reference modules.py:259-263):
//   out[b][r][i * M/2 + m] += window[m] * chunks[i - c0][b][r][m]      for the chunks i = c0 .. c1-1 of this call, ascending i
// One thread per four output frames of a row: it visits the (at most two) chunks of the call that cover those frames in
// ascending chunk order, which is the reference's sequential accumulation order -- a frame's two contributions meet in the same
// order whether they arrive in one call or in two consecutive calls, so the sum is bit-identical to the Python loop.
__global__ __launch_bounds__(256) void k_window_ola(const float* __restrict__ chunks, const float* __restrict__ window,
                                                    float* __restrict__ out, long rows, int M, int c0, int c1, long n_frames) {
    const int half = M >> 1;
    const long p0 = (long)c0 * half, span4 = ((long)(c1 - c0 - 1) * half + M) >> 2;      // frames touched by this call / 4
    const long total = rows * span4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / span4, q = i - r * span4;
        const long p = p0 + 4 * q;                               // first of four output frames (M/2 % 4 == 0: same chunks for all four)
        float4 acc = *reinterpret_cast<const float4*>(out + r * n_frames + p);
        const int hi = (int)(p / half);                          // chunks hi - 1 and hi cover frame p
#pragma unroll
        for (int k = 1; k >= 0; --k) {
            const int ci = hi - k;
            if (ci < c0 || ci >= c1) continue;
            const int m = (int)(p - (long)ci * half);
            const float4 w = *reinterpret_cast<const float4*>(window + m);
            const float4 v = *reinterpret_cast<const float4*>(chunks + ((long)(ci - c0) * rows + r) * M + m);
            // product and sum rounded separately: `out += window * chunk` of the reference is two fp32 roundings, so the
            // products are pinned in registers before the adds (the compiler would otherwise contract them into FMAs)
            float px = w.x * v.x, py = w.y * v.y, pz = w.z * v.z, pw = w.w * v.w;
            asm volatile("" : "+v"(px), "+v"(py), "+v"(pz), "+v"(pw));
            acc.x += px; acc.y += py; acc.z += pz; acc.w += pw;
        }
        *reinterpret_cast<float4*>(out + r * n_frames + p) = acc;
    }
}

extern "C" int tt_window_ola(const float* chunks, const float* window, float* out, int64_t rows, int M, int c0, int c1,
                             int64_t n_frames, void* stream) {
    if (!chunks || !window || !out || rows <= 0 || M <= 0 || (M & 7) || c1 <= c0 || c0 < 0) return TT_E_BADARG;
    if ((long)(c1 - 1) * (M / 2) + M > n_frames || (n_frames & 3)) return TT_E_BADARG;
    const long total = rows * ((((long)(c1 - c0 - 1) * (M / 2)) + M) / 4);
    hipLaunchKernelGGL(k_window_ola, dim3(grid1d(total, 256, 8 * tt_cus())), dim3(256), 0, tt_stream(stream), chunks, window, out,
                       (long)rows, M, c0, c1, (long)n_frames);
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" int tt_dot(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n < 0) return TT_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_dot, dim3(grid1d(n, 256 * 16, 1024)), dim3(256), 0, tt_stream(stream), a, b, out, (long)n);
    TT_LAUNCH_CHECK();
    return 0;
}
