// Fused ResidualConv2dBlock for gfx950 on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32,
// bitwise an fmaf chain, at the fp32 vector rate with two LDS reads per 32-cycle instruction).
//
//   y = ELU(W2 . ELU(W1 (*)_dil x + b1) + b2) + x        reference timbre_trap/framework/modules.py:755-777
//
// The 3x3 dilated convolution is an implicit GEMM  D[co][pix] = sum_k W1s[k][co] * X[k][pix],
// k = tap*C + ci, executed as 16(co) x 16(pix) x 4(k) MFMA tiles:
//   A fragment  lane l : W1s[k0 + (l>>4)][mt*16 + (l&15)]        (weights, transposed once into LDS)
//   B fragment  lane l : xs[ci = .. + (l>>4)][row + kh*D][col + (l&15) + kw*D]   (input tile + halo in LDS)
//   D fragment  lane l : rows 4*(l>>4) + r (r = 0..3), column l&15
// Because the D fragment of the 3x3 stage holds, for a fixed register r, channels {4g + r}
// across the four 16-lane groups g, it IS a valid B fragment (k = g) for the following 1x1
// convolution: the hidden activation never leaves registers.
//
// Workgroup = 8 waves = 8 rows x 64 columns of one clip; input staged per 8-channel chunk with a
// D-wide halo; workgroups are persistent over tiles so the weight image is built once.
//
// Backward = three launches (hidden activations recomputed from x, nothing but x saved):
//   k_rb_bwd_a   recompute, pointwise chain -> dA1 (gradient at the 3x3 pre-activation) to scratch,
//                db1, db2, dW2 accumulated in registers across tiles then atomically added
//   k_rb_conv    (same kernel as forward, MODE 1) dx = dy + W1^T (*) dA1   (flipped weights)
//   k_rb_wgrad   dW1[co][tap,ci] = sum_pix dA1[co][pix] * x[ci][pix + tap]  as MFMA with K = pixels
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int plane_pad(int n) {          // smallest p >= n with p % 32 == 17 (conflict-light for both access patterns)
    int p = n;
    while (p % 32 != 17) ++p;
    return p;
}

template <int C, int D>
struct Cfg {
    static constexpr int MT = (C + 15) / 16;
    static constexpr int CPAD = MT * 16;
    static constexpr int CP = CPAD + ((MT % 2 == 0) ? 16 : 0);     // weight row pitch (floats)
    static constexpr int TH = 8, TW = 64, NT = 4;
    static constexpr int XR = TH + 2 * D, XC = TW + 2 * D;
    static constexpr int PLANE = plane_pad(XR * XC);
    static constexpr int CC = C < 8 ? C : 8;
    static constexpr int K1 = 9 * C;
    // LDS carve (floats)
    static constexpr int W1_OFF = 0;
    static constexpr int W2_OFF = W1_OFF + K1 * CP;
    static constexpr int W2T_OFF = W2_OFF + CPAD * CP;
    static constexpr int B1_OFF = W2T_OFF + CPAD * CP;
    static constexpr int B2_OFF = B1_OFF + CPAD;
    static constexpr int XS_OFF = B2_OFF + CPAD;
    static constexpr int XS_SIZE_CONV = CC * PLANE;
    static constexpr int TP = 17;                                   // pitch of the per-wave transpose tiles
    static constexpr int XS_SIZE_TR = 8 * 2 * CPAD * TP;            // 8 waves x {dA2, h1} x CPAD x 16 pixels
    static constexpr int XS_SIZE = XS_SIZE_CONV > XS_SIZE_TR ? XS_SIZE_CONV : XS_SIZE_TR;
    static constexpr int LDS_FLOATS = XS_OFF + XS_SIZE;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

struct Tile { int b, h0, t0; };
__device__ __forceinline__ Tile decode_tile(int tile, int tiles_h, int tiles_t, int TH, int TW) {
    Tile r;
    const int tt = tile % tiles_t; tile /= tiles_t;
    const int th = tile % tiles_h;
    r.b = tile / tiles_h; r.h0 = th * TH; r.t0 = tt * TW;
    return r;
}

// Build the LDS weight images.  flip = data-gradient form (in/out channels swapped, taps reversed).
template <int C, int D>
__device__ __forceinline__ void load_weights(float* lds, const float* __restrict__ w1, const float* __restrict__ b1,
                                             const float* __restrict__ w2, const float* __restrict__ b2, bool flip,
                                             int tid, int nthreads) {
    using K = Cfg<C, D>;
    for (int i = tid; i < K::K1 * K::CP; i += nthreads) {
        const int k = i / K::CP, co = i - k * K::CP;
        const int tap = k / C, ci = k - tap * C;
        float v = 0.f;
        if (co < C) v = flip ? w1[(ci * C + co) * 9 + (8 - tap)] : w1[(co * C + ci) * 9 + tap];
        lds[K::W1_OFF + i] = v;
    }
    for (int i = tid; i < K::CPAD * K::CP; i += nthreads) {
        const int r = i / K::CP, c = i - r * K::CP;
        // W2s[c_in = r][co2 = c] ; W2t[co2 = r][c_in = c]
        lds[K::W2_OFF + i] = (w2 && r < C && c < C) ? w2[c * C + r] : 0.f;
        lds[K::W2T_OFF + i] = (w2 && r < C && c < C) ? w2[r * C + c] : 0.f;
    }
    for (int i = tid; i < K::CPAD; i += nthreads) {
        lds[K::B1_OFF + i] = (b1 && i < C) ? b1[i] : 0.f;
        lds[K::B2_OFF + i] = (b2 && i < C) ? b2[i] : 0.f;
    }
}

// 3x3 dilated implicit-GEMM of one tile into acc[MT][NT]; x staged per channel chunk.
template <int C, int D>
__device__ __forceinline__ void conv3x3_tile(const float* __restrict__ x, float* lds, const Tile& tl, int H, int T,
                                             f32x4 (&acc)[Cfg<C, D>::MT][4], int tid, int wave, int g, int l15) {
    using K = Cfg<C, D>;
    float* xs = lds + K::XS_OFF;
    const float* W1s = lds + K::W1_OFF;
#pragma unroll
    for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long plane = (long)H * T;
    for (int c0 = 0; c0 < C; c0 += K::CC) {
        __syncthreads();
        const float* xb = x + ((long)tl.b * C + c0) * plane;
        for (int i = tid; i < K::CC * K::XR * K::XC; i += 512) {
            const int ci = i / (K::XR * K::XC);
            const int rem = i - ci * (K::XR * K::XC);
            const int r = rem / K::XC, c = rem - r * K::XC;
            const int h = tl.h0 - D + r, t = tl.t0 - D + c;
            float v = 0.f;
            if (h >= 0 && h < H && t >= 0 && t < T) v = xb[ci * plane + (long)h * T + t];
            xs[ci * K::PLANE + r * K::XC + c] = v;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
#pragma unroll
            for (int cc = 0; cc < K::CC; cc += 4) {
                float a[K::MT];
#pragma unroll
                for (int mt = 0; mt < K::MT; ++mt) a[mt] = W1s[(tap * C + c0 + cc + g) * K::CP + mt * 16 + l15];
                const float* bp = xs + (cc + g) * K::PLANE + (wave + kh * D) * K::XC + l15 + kw * D;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float bv = bp[nt * 16];
#pragma unroll
                    for (int mt = 0; mt < K::MT; ++mt) acc[mt][nt] = mfma16(a[mt], bv, acc[mt][nt]);
                }
            }
        }
    }
}

// MODE 0: full residual block.   MODE 1: y = conv3x3(x; W1) + res   (no bias, no activation)
template <int C, int D, int MODE>
__global__ __launch_bounds__(512) void k_rb_conv(const float* __restrict__ x, const float* __restrict__ w1,
                                                 const float* __restrict__ b1, const float* __restrict__ w2,
                                                 const float* __restrict__ b2, const float* __restrict__ res,
                                                 float* __restrict__ y, int B, int H, int T, int flip) {
    using K = Cfg<C, D>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    load_weights<C, D>(lds, w1, b1, MODE == 0 ? w2 : nullptr, b2, flip != 0, tid, 512);
    const int tiles_h = (H + K::TH - 1) / K::TH, tiles_t = (T + K::TW - 1) / K::TW;
    const int ntiles = B * tiles_h * tiles_t;
    const long plane = (long)H * T;
    const float* W2s = lds + K::W2_OFF;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const Tile tl = decode_tile(tile, tiles_h, tiles_t, K::TH, K::TW);
        f32x4 acc[K::MT][4];
        conv3x3_tile<C, D>(x, lds, tl, H, T, acc, tid, wave, g, l15);
        const int h = tl.h0 + wave;
        if (MODE == 0) {
            // hidden activation in registers, then the 1x1 convolution with the D fragments as B operands
            f32x4 acc2[K::MT][4];
#pragma unroll
            for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mt][nt][r] = elu1(acc[mt][nt][r] + lds[K::B1_OFF + mt * 16 + 4 * g + r]);
                }
#pragma unroll
            for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a2[K::MT];
#pragma unroll
                    for (int m2 = 0; m2 < K::MT; ++m2) a2[m2] = W2s[(mt * 16 + 4 * g + r) * K::CP + m2 * 16 + l15];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int m2 = 0; m2 < K::MT; ++m2) acc2[m2][nt] = mfma16(a2[m2], acc[mt][nt][r], acc2[m2][nt]);
                }
            if (h < H) {
#pragma unroll
                for (int m2 = 0; m2 < K::MT; ++m2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = m2 * 16 + 4 * g + r;
                        if (co >= C) continue;
                        const float bias = lds[K::B2_OFF + co];
                        const long base = ((long)tl.b * C + co) * plane + (long)h * T;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const int t = tl.t0 + nt * 16 + l15;
                            if (t < T) y[base + t] = elu1(acc2[m2][nt][r] + bias) + x[base + t];
                        }
                    }
            }
        } else {
            if (h < H) {
#pragma unroll
                for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = mt * 16 + 4 * g + r;
                        if (co >= C) continue;
                        const long base = ((long)tl.b * C + co) * plane + (long)h * T;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const int t = tl.t0 + nt * 16 + l15;
                            if (t < T) y[base + t] = acc[mt][nt][r] + (res ? res[base + t] : 0.f);
                        }
                    }
            }
        }
    }
}

// sum the 16 lanes that share (g) and add to out[index(g, r)]
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}

template <int C, int D>
__global__ __launch_bounds__(512) void k_rb_bwd_a(const float* __restrict__ x, const float* __restrict__ dy,
                                                  const float* __restrict__ w1, const float* __restrict__ b1,
                                                  const float* __restrict__ w2, const float* __restrict__ b2,
                                                  float* __restrict__ da1, float* __restrict__ db1,
                                                  float* __restrict__ dw2, float* __restrict__ db2, int B, int H, int T) {
    using K = Cfg<C, D>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    load_weights<C, D>(lds, w1, b1, w2, b2, false, tid, 512);
    const int tiles_h = (H + K::TH - 1) / K::TH, tiles_t = (T + K::TW - 1) / K::TW;
    const int ntiles = B * tiles_h * tiles_t;
    const long plane = (long)H * T;
    const float* W2s = lds + K::W2_OFF;
    const float* W2t = lds + K::W2T_OFF;
    float* trA = lds + K::XS_OFF + wave * 2 * K::CPAD * K::TP;     // dA2 [CPAD][TP]
    float* trB = trA + K::CPAD * K::TP;                              // h1  [CPAD][TP]

    f32x4 accw2[K::MT][K::MT];            // dW2[co2 tile][c tile]
    float db1acc[K::MT][4], db2acc[K::MT][4];
#pragma unroll
    for (int a = 0; a < K::MT; ++a) {
#pragma unroll
        for (int c = 0; c < K::MT; ++c) accw2[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) { db1acc[a][r] = 0.f; db2acc[a][r] = 0.f; }
    }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const Tile tl = decode_tile(tile, tiles_h, tiles_t, K::TH, K::TW);
        f32x4 h1[K::MT][4];
        conv3x3_tile<C, D>(x, lds, tl, H, T, h1, tid, wave, g, l15);
        const int h = tl.h0 + wave;
        f32x4 a2[K::MT][4];
#pragma unroll
        for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                a2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[mt][nt][r] = elu1(h1[mt][nt][r] + lds[K::B1_OFF + mt * 16 + 4 * g + r]);
            }
#pragma unroll
        for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float av[K::MT];
#pragma unroll
                for (int m2 = 0; m2 < K::MT; ++m2) av[m2] = W2s[(mt * 16 + 4 * g + r) * K::CP + m2 * 16 + l15];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int m2 = 0; m2 < K::MT; ++m2) a2[m2][nt] = mfma16(av[m2], h1[mt][nt][r], a2[m2][nt]);
            }
        // dA2 = dy * ELU'(a2 + b2), in place in a2
#pragma unroll
        for (int m2 = 0; m2 < K::MT; ++m2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m2 * 16 + 4 * g + r;
                const float bias = lds[K::B2_OFF + co];
                const long base = ((long)tl.b * C + (co < C ? co : 0)) * plane + (long)(h < H ? h : 0) * T;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int t = tl.t0 + nt * 16 + l15;
                    float d = 0.f;
                    if (co < C && h < H && t < T) d = dy[base + t];
                    const float h2 = elu1(a2[m2][nt][r] + bias);
                    const float gd = d * elu_grad_from_out(h2);
                    a2[m2][nt][r] = gd;
                    db2acc[m2][r] += gd;
                }
            }
        // dH1 = W2^T . dA2   (dA2 fragments as B operands), then dA1 = dH1 * ELU'(h1)
        f32x4 d1[K::MT][4];
#pragma unroll
        for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) d1[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m2 = 0; m2 < K::MT; ++m2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float av[K::MT];
#pragma unroll
                for (int mt = 0; mt < K::MT; ++mt) av[mt] = W2t[(m2 * 16 + 4 * g + r) * K::CP + mt * 16 + l15];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < K::MT; ++mt) d1[mt][nt] = mfma16(av[mt], a2[m2][nt][r], d1[mt][nt]);
            }
#pragma unroll
        for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = mt * 16 + 4 * g + r;
                const long base = ((long)tl.b * C + (co < C ? co : 0)) * plane + (long)(h < H ? h : 0) * T;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int t = tl.t0 + nt * 16 + l15;
                    const float gd = d1[mt][nt][r] * elu_grad_from_out(h1[mt][nt][r]);
                    d1[mt][nt][r] = gd;
                    db1acc[mt][r] += gd;
                    if (co < C && h < H && t < T) da1[base + t] = gd;
                }
            }
        // dW2[co2][c] += sum_pix dA2[co2][pix] * h1[c][pix] : transpose 16 pixels at a time through LDS
        __syncthreads();                       // every wave is done with the conv staging area
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    trA[(mt * 16 + 4 * g + r) * K::TP + l15] = a2[mt][nt][r];
                    trB[(mt * 16 + 4 * g + r) * K::TP + l15] = h1[mt][nt][r];
                }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float av[K::MT], bv[K::MT];
#pragma unroll
                for (int mt = 0; mt < K::MT; ++mt) {
                    av[mt] = trA[(mt * 16 + l15) * K::TP + ks * 4 + g];
                    bv[mt] = trB[(mt * 16 + l15) * K::TP + ks * 4 + g];
                }
#pragma unroll
                for (int m2 = 0; m2 < K::MT; ++m2)
#pragma unroll
                    for (int mt = 0; mt < K::MT; ++mt) accw2[m2][mt] = mfma16(av[m2], bv[mt], accw2[m2][mt]);
            }
            __syncthreads();
        }
    }
    // flush the per-wave accumulators
#pragma unroll
    for (int m2 = 0; m2 < K::MT; ++m2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = m2 * 16 + 4 * g + r;
            const float s1 = group16_sum(db1acc[m2][r]), s2 = group16_sum(db2acc[m2][r]);
            if (l15 == 0 && co < C) { atomicAdd(db1 + co, s1); atomicAdd(db2 + co, s2); }
#pragma unroll
            for (int mt = 0; mt < K::MT; ++mt) {
                const int c = mt * 16 + l15;
                if (co < C && c < C) atomicAdd(dw2 + co * C + c, accw2[m2][mt][r]);
            }
        }
}

// ---- weight gradient of the 3x3 convolution ----------------------------------------------------
template <int C, int D>
struct WCfg {
    static constexpr int MT = (C + 15) / 16;
    static constexpr int CPAD = MT * 16;
    static constexpr int NN = 9 * C;
    static constexpr int NTN = (NN + 15) / 16;
    static constexpr int TH = 4, TW = 64;
    static constexpr int XR = TH + 2 * D, XC = TW + 2 * D;
    static constexpr int PLANE = plane_pad(XR * XC);
    static constexpr int AP = 65;                                   // pitch of the dA1 rows
    static constexpr int XS_OFF = 0;
    static constexpr int AS_OFF = XS_OFF + C * PLANE;
    static constexpr int LDS_FLOATS = AS_OFF + 4 * CPAD * AP;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

template <int C, int D>
__global__ __launch_bounds__(256) void k_rb_wgrad(const float* __restrict__ x, const float* __restrict__ da1,
                                                  float* __restrict__ dw1, int B, int H, int T) {
    using K = WCfg<C, D>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    float* xs = lds + K::XS_OFF;
    float* as = lds + K::AS_OFF + wave * K::CPAD * K::AP;
    const int tiles_h = (H + K::TH - 1) / K::TH, tiles_t = (T + K::TW - 1) / K::TW;
    const int ntiles = B * tiles_h * tiles_t;
    const long plane = (long)H * T;

    int noff[K::NTN];                      // per-lane LDS offset of column n = nt*16 + l15 -> (tap, ci)
#pragma unroll
    for (int nt = 0; nt < K::NTN; ++nt) {
        int n = nt * 16 + l15;
        if (n >= K::NN) n = K::NN - 1;
        const int tap = n / C, ci = n - tap * C;
        noff[nt] = ci * K::PLANE + (tap / 3) * D * K::XC + (tap % 3) * D;
    }
    f32x4 acc[K::MT][K::NTN];
#pragma unroll
    for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < K::NTN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const Tile tl = decode_tile(tile, tiles_h, tiles_t, K::TH, K::TW);
        __syncthreads();
        const float* xb = x + (long)tl.b * C * plane;
        for (int i = tid; i < C * K::XR * K::XC; i += 256) {
            const int ci = i / (K::XR * K::XC);
            const int rem = i - ci * (K::XR * K::XC);
            const int r = rem / K::XC, c = rem - r * K::XC;
            const int h = tl.h0 - D + r, t = tl.t0 - D + c;
            float v = 0.f;
            if (h >= 0 && h < H && t >= 0 && t < T) v = xb[ci * plane + (long)h * T + t];
            xs[ci * K::PLANE + r * K::XC + c] = v;
        }
        {   // each wave stages its own row of dA1: as[co][t]
            const int h = tl.h0 + wave, t = tl.t0 + lane;
            const bool ok = h < H && t < T;
            const float* ab = da1 + (long)tl.b * C * plane + (long)(h < H ? h : 0) * T + (t < T ? t : 0);
#pragma unroll 4
            for (int co = 0; co < K::CPAD; ++co) as[co * K::AP + lane] = (ok && co < C) ? ab[co * plane] : 0.f;
        }
        __syncthreads();
        const float* xrow = xs + wave * K::XC;
#pragma unroll 2
        for (int ks = 0; ks < 16; ++ks) {
            float av[K::MT];
#pragma unroll
            for (int mt = 0; mt < K::MT; ++mt) av[mt] = as[(mt * 16 + l15) * K::AP + ks * 4 + g];
            const float* xp = xrow + ks * 4 + g;
#pragma unroll
            for (int nt = 0; nt < K::NTN; ++nt) {
                const float bv = xp[noff[nt]];
#pragma unroll
                for (int mt = 0; mt < K::MT; ++mt) acc[mt][nt] = mfma16(av[mt], bv, acc[mt][nt]);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < K::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < K::NTN; ++nt) {
            const int n = nt * 16 + l15;
            if (n >= K::NN) continue;
            const int tap = n / C, ci = n - tap * C;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = mt * 16 + 4 * g + r;
                if (co < C) atomicAdd(dw1 + (co * C + ci) * 9 + tap, acc[mt][nt][r]);
            }
        }
}

inline int persistent_grid(int ntiles, int per_cu) {
    const int cap = 256 * per_cu;
    return ntiles < cap ? ntiles : cap;
}

template <int C, int D>
int launch_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y, int B,
               int H, int T, hipStream_t st) {
    using K = Cfg<C, D>;
    static bool attr = false;
    if (!attr) {
        TT_HIP(hipFuncSetAttribute((const void*)k_rb_conv<C, D, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
        attr = true;
    }
    const int ntiles = B * ((H + K::TH - 1) / K::TH) * ((T + K::TW - 1) / K::TW);
    const int per_cu = (160 * 1024) / K::LDS_BYTES > 4 ? 4 : ((160 * 1024) / K::LDS_BYTES < 1 ? 1 : (160 * 1024) / K::LDS_BYTES);
    hipLaunchKernelGGL((k_rb_conv<C, D, 0>), dim3(persistent_grid(ntiles, per_cu)), dim3(512), K::LDS_BYTES, st, x, w1, b1,
                       w2, b2, (const float*)nullptr, y, B, H, T, 0);
    TT_LAUNCH_CHECK();
    return 0;
}

template <int C, int D>
int launch_bwd(const float* x, const float* dy, const float* w1, const float* b1, const float* w2, const float* b2,
               float* dx, float* dw1, float* db1, float* dw2, float* db2, float* ws, int B, int H, int T, hipStream_t st) {
    using K = Cfg<C, D>;
    using W = WCfg<C, D>;
    static bool attr = false;
    if (!attr) {
        TT_HIP(hipFuncSetAttribute((const void*)k_rb_bwd_a<C, D>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
        TT_HIP(hipFuncSetAttribute((const void*)k_rb_conv<C, D, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
        TT_HIP(hipFuncSetAttribute((const void*)k_rb_wgrad<C, D>, hipFuncAttributeMaxDynamicSharedMemorySize, W::LDS_BYTES));
        attr = true;
    }
    const int ntiles = B * ((H + K::TH - 1) / K::TH) * ((T + K::TW - 1) / K::TW);
    int per_cu = (160 * 1024) / K::LDS_BYTES;
    per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
    hipLaunchKernelGGL((k_rb_bwd_a<C, D>), dim3(persistent_grid(ntiles, per_cu)), dim3(512), K::LDS_BYTES, st, x, dy, w1, b1,
                       w2, b2, ws, db1, dw2, db2, B, H, T);
    TT_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_rb_conv<C, D, 1>), dim3(persistent_grid(ntiles, per_cu)), dim3(512), K::LDS_BYTES, st,
                       (const float*)ws, w1, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, dy, dx,
                       B, H, T, 1);
    TT_LAUNCH_CHECK();
    const int wtiles = B * ((H + W::TH - 1) / W::TH) * ((T + W::TW - 1) / W::TW);
    int wper = (160 * 1024) / W::LDS_BYTES;
    wper = wper > 4 ? 4 : (wper < 1 ? 1 : wper);
    hipLaunchKernelGGL((k_rb_wgrad<C, D>), dim3(persistent_grid(wtiles, wper)), dim3(256), W::LDS_BYTES, st, x,
                       (const float*)ws, dw1, B, H, T);
    TT_LAUNCH_CHECK();
    return 0;
}

#define TT_DISPATCH_CD(FN, ...)                                                       \
    switch (C * 10 + dilation) {                                                      \
        case 41: return FN<4, 1>(__VA_ARGS__);   case 42: return FN<4, 2>(__VA_ARGS__);   case 43: return FN<4, 3>(__VA_ARGS__);   \
        case 81: return FN<8, 1>(__VA_ARGS__);   case 82: return FN<8, 2>(__VA_ARGS__);   case 83: return FN<8, 3>(__VA_ARGS__);   \
        case 161: return FN<16, 1>(__VA_ARGS__); case 162: return FN<16, 2>(__VA_ARGS__); case 163: return FN<16, 3>(__VA_ARGS__); \
        case 321: return FN<32, 1>(__VA_ARGS__); case 322: return FN<32, 2>(__VA_ARGS__); case 323: return FN<32, 3>(__VA_ARGS__); \
        default: return TT_E_UNSUPPORTED;                                             \
    }

}  // namespace

extern "C" int tt_resblock_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                               float* y, int B, int C, int H, int T, int dilation, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || B <= 0 || H <= 0 || T <= 0) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    TT_DISPATCH_CD(launch_fwd, x, w1, b1, w2, b2, y, B, H, T, st)
}

extern "C" int tt_resblock_bwd(const float* x, const float* dy, const float* w1, const float* b1, const float* w2,
                               const float* b2, float* dx, float* dw1, float* db1, float* dw2, float* db2, float* ws,
                               int B, int C, int H, int T, int dilation, void* stream) {
    if (!x || !dy || !w1 || !b1 || !w2 || !b2 || !dx || !dw1 || !db1 || !dw2 || !db2 || !ws || B <= 0 || H <= 0 || T <= 0)
        return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    TT_DISPATCH_CD(launch_bwd, x, dy, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st)
}
