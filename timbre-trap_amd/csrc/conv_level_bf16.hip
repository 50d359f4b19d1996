// Fused kernels of the wide residual levels (C = 16, 32; reference modules.py:621-624, 690-693, 721-777) on bf16
// channel-innermost tensors [B][H][T][C] for gfx950.  conv_wide_bf16.hip runs one kernel per STAGE of a block (forward;
// pointwise chain, data gradient, weight gradient) and every stage goes through HBM.  Here:
//
//   k_wrb_bwd_fused<C,D,TH,TW>  the WHOLE backward of a ResidualConv2dBlock in one pass from x and dy only:
//       phase 0  x tile with a 2D halo and dy tile with a D halo arrive by LDS-DMA (out-of-image pieces from a zero page);
//       phase 1  on every pixel of the tile + D halo the hidden activation is RECOMPUTED (h1 = ELU(W1 (*)_D x + b1), the
//                forward's own product order -> bit-identical to what the forward would have stored), then the pointwise
//                chain a2 = W2 h1 + b2, dA2 = dy ELU'(a2), dh1 = W2^T dA2, dA1 = dh1 ELU'(a1); dA1 (bf16) OVERWRITES dy in
//                LDS; db2 / dW2 over the tile's own pixels (K = pixels via the per-wave transposition buffers);
//       phase 2  dx = dy + W1^T (*)_D dA1 (dy of the tile's own pixels re-read from L2), db1 from the centre tap, and
//                dW1[co][ci][tap] = sum_pix dA1[co][pix] x[ci][pix + tap] by LDS transpose reads of the two images.
//     Neither h1 nor dA1 ever exists in HBM: a block's backward reads x and dy and writes dx (3 tensors instead of 8), and the
//     forward no longer stores h1 (2 tensors instead of 3).  The matrix pipe was under 25 % busy in every per-stage kernel, so
//     the 18 extra products per 16 pixels (times the halo overhead of phase 1) are paid from idle issue slots.
//
// One LDS layout serves both access patterns: the 16-byte channel group cg of the pixel in image column `col` sits at position
// cg ^ fswz(col).  C = 32: fswz = bit-reversed (col >> 2) & 3, so that (i) the sixteen consecutive columns a k-group reads as
// B operand (ds_read_b128) cover all sixteen bank quads and (ii) the eight consecutive pixels one half-wave addresses in a
// transpose read (ds_read_b64_tr_b16, 32 bytes of each) use alternating 32-byte halves, for ANY column alignment (the taps
// shift the columns by multiples of D).  C = 16: (col >> 3) & 1.
#include "wide_common.h"

namespace {

// blockIdx.y selects one of up to four (w1, w2, image) triples: tt_wide_level_bwd prepares the images of all blocks of a level in one launch
struct WPrep { const float* w1[4]; const float* w2[4]; e16x8* img[4]; };
template <int C>
__global__ __launch_bounds__(64) void k_lvl_wprep(WPrep wp_) {
    using K = WK<C>;
    const float* __restrict__ w1 = wp_.w1[blockIdx.y];
    const float* __restrict__ w2 = wp_.w2[blockIdx.y];
    e16x8* __restrict__ img = wp_.img[blockIdx.y];
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4, blk = blockIdx.x;
    if (blk < K::NK * K::NCT) {
        const int k = blk / K::NCT, ct = blk - k * K::NCT;
        e16x8 f, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int tap, kc;                                         // tap and contraction channel of this lane's k = 8 g + j
            if (C == 32) { tap = k; kc = 8 * g + j; }
            else { tap = 2 * k + (g >> 1); kc = 8 * (g & 1) + j; }
            const int mo = chan_of<C>(ct, n);
            f[j] = (e16)(tap < 9 ? w1[(mo * C + kc) * 9 + tap] : 0.f);
            b[j] = (e16)(tap < 9 ? w1[(kc * C + mo) * 9 + (8 - tap)] : 0.f);
        }
        img[blk * 64 + lane] = f;
        img[K::W3 + blk * 64 + lane] = b;
    } else {
#pragma unroll
        for (int ct = 0; ct < K::NCT; ++ct) {
            e16x8 a, at;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float va = 0.f, vt = 0.f;
                if (C == 32) { va = w2[chan_of<C>(ct, n) * C + 8 * g + j]; vt = w2[(8 * g + j) * C + chan_of<C>(ct, n)]; }
                else if (j < 4) { va = w2[n * C + 4 * g + j]; vt = w2[(4 * g + j) * C + n]; }
                a[j] = (e16)va; at[j] = (e16)vt;
            }
            img[2 * K::W3 + ct * 64 + lane] = a;
            img[2 * K::W3 + K::NCT * 64 + ct * 64 + lane] = at;
        }
    }
}

template <int C, int D, int TH, int TW, int NW> struct FB {
    static constexpr int NTH = NW * 64;
    static constexpr int CG = C / 8, PB = C * 2;
    static constexpr int XR = TH + 4 * D, XW = TW + 4 * D, GR = TH + 2 * D, GW = TW + 2 * D;
    static constexpr int XP = XR * XW * CG, GP = GR * GW * CG;   // 16-byte pieces
    static constexpr int XPR = (XP + NTH - 1) / NTH * NTH, GPR = (GP + NTH - 1) / NTH * NTH;
    static constexpr int X_BYTES = XPR * 16, G_BYTES = GPR * 16;
    static constexpr int PS = C * 2 + 8;                         // bytes per pixel of the transposition buffers (bank spread)
    static constexpr int T_BYTES = NW * 2 * 16 * PS;
    static constexpr int ADUMP = C * C + 2 * C;
    static constexpr int WDUMP = 9 * (C / 16) * 256;
    static constexpr int LDS_BYTES = X_BYTES + G_BYTES + T_BYTES;
    static constexpr int NG1 = (GR * GW + 15) / 16;              // 16-pixel groups of phase 1 (linear over the dy / dA1 image)
    static_assert(NW * ADUMP * 4 <= X_BYTES, "final dump reduction reuses the x image");
    static_assert(TW % 32 == 0, "the K = 32 pixels of a weight-gradient product are 32 consecutive columns");
};

template <int C, int D, int TH, int TW, int NW>
__global__ __launch_bounds__(NW * 64, C == 32 ? 2 : 3) void k_wrb_bwd_fused(const e16* __restrict__ x, const e16* __restrict__ dy,
                                                              const e16x8* __restrict__ wimg, const float* __restrict__ b1,
                                                              const float* __restrict__ b2, e16* __restrict__ dx,
                                                              float* __restrict__ part_a, float* __restrict__ part_w, int B, int H,
                                                              int T, int tiles_h, int tiles_t, int ntiles) {
    using G = FB<C, D, TH, TW, NW>;
    using K = WK<C>;
    constexpr int NCT = K::NCT, NK = K::NK, NCH = K::NCH, PB = G::PB;
    typedef typename std::conditional<C == 32, e16x8, e16x4>::type vec_t;     // a lane's channels of one pixel
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* xs = smem;
    unsigned char* gs = smem + G::X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4, trj = n >> 2, trq = n & 3;
    unsigned char* tg = smem + G::X_BYTES + G::G_BYTES + wave * (2 * 16 * G::PS);   // this wave's dA2 tile, then its h1 tile
    unsigned char* thh = tg + 16 * G::PS;
    // the lane's own channels NCH g .. NCH g + NCH - 1 of a pixel: 16-byte piece and byte inside it
    const int opiece = C == 32 ? g : (g >> 1), obyte = C == 32 ? 0 : 8 * (g & 1);

    // weight-gradient roles.  C = 32: wave = (ci-tile wave & 1, co-tile wave >> 1), every wave walks all rows and 32-column
    // chunks (ONE co-tile of accumulators per wave: 36 VGPRs instead of 72 -- the 3x3 weights of a phase are 72 more);
    // C = 16: the waves split the 32-column chunks, then the rows.
    static_assert(C == 16 || NW == 4, "C = 32 roles need exactly four waves");
    constexpr int NCHK = TW / 32;
    static_assert(C == 32 || (NW % NCHK == 0), "wave roles");
    const int cit = C == 32 ? (wave & 1) : 0, aw = C == 32 ? (wave >> 1) : 0;
    const int ch0 = C == 32 ? 0 : wave % NCHK, rpar = C == 32 ? 0 : wave / NCHK;
    constexpr int CHSTEP = C == 32 ? 1 : NCHK, RSTEP = C == 32 ? 1 : NW / NCHK;

    f32x4 wacc[9], dw2[NCT][NCT];
#pragma unroll
    for (int k = 0; k < 9; ++k) wacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < NCT; ++a)
#pragma unroll
        for (int c = 0; c < NCT; ++c) dw2[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db1a[NCH], db2a[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) { db1a[j] = 0.f; db2a[j] = 0.f; }
    vec_t zero_v;
#pragma unroll
    for (int j = 0; j < NCH; ++j) zero_v[j] = (e16)0.f;

    const e16* zero = reinterpret_cast<const e16*>(&g_wzero16);
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        int tile = xcd_order(v, ntiles);
        const int tt = tile % tiles_t; tile /= tiles_t;
        const int th = tile % tiles_h;
        const int b = tile / tiles_h, h0 = th * TH, t0 = tt * TW;
        const e16* xb = x + (long)b * H * T * C;
        const e16* gb = dy + (long)b * H * T * C;

        __syncthreads();                                         // the previous tile has been consumed
        for (int i = wave * 64; i < G::XPR; i += G::NTH) {
            const int p = i + lane, q = p / G::CG, s = p - q * G::CG;
            const int row = q / G::XW, px = q - row * G::XW;
            const int h = h0 - 2 * D + row, t = t0 - 2 * D + px;
            const bool ok = p < G::XP && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
            glds16(ok ? xb + ((long)h * T + t) * C + (s ^ fswz<C>(px)) * 8 : zero, xs + (long)i * 16);
        }
        for (int i = wave * 64; i < G::GPR; i += G::NTH) {
            const int p = i + lane, q = p / G::CG, s = p - q * G::CG;
            const int row = q / G::GW, px = q - row * G::GW;
            const int h = h0 - D + row, t = t0 - D + px;
            const bool ok = p < G::GP && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
            glds16(ok ? gb + ((long)h * T + t) * C + (s ^ fswz<C>(px)) * 8 : zero, gs + (long)i * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // The weights of each phase are (re)loaded per tile through laundered pointers so that their registers are free in
        // the other phase (72 VGPRs per orientation at C = 32 beside 72 of weight-gradient accumulators).
        const e16x8* wp = wimg;
        const float* b1p = b1;
        const float* b2p = b2;
        asm volatile("" : "+s"(wp), "+s"(b1p), "+s"(b2p));

        // ---- phase 1: h1 recomputed, pointwise chain, dA1 over dy in LDS; db2 / dW2 over the tile's own pixels ----
        {
            e16x8 A[NK][NCT];
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) A[k][ct] = wp[(k * NCT + ct) * 64 + lane];
            e16x8 A2[NCT], A2T[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                A2[ct] = wp[2 * K::W3 + ct * 64 + lane];
                A2T[ct] = wp[2 * K::W3 + NCT * 64 + ct * 64 + lane];
            }
            const s16x4 A2s = __builtin_bit_cast(s16x4, __builtin_shufflevector(A2[0], A2[0], 0, 1, 2, 3));      // C = 16: K = 16
            const s16x4 A2Ts = __builtin_bit_cast(s16x4, __builtin_shufflevector(A2T[0], A2T[0], 0, 1, 2, 3));
            float b1r[NCH], b2r[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) { b1r[j] = b1p[NCH * g + j]; b2r[j] = b2p[NCH * g + j]; }

            for (int grp = wave; grp < G::NG1; grp += NW) {
                const int q = grp * 16 + n;
                const bool inq = q < G::GR * G::GW;
                const int qq = inq ? q : G::GR * G::GW - 1;
                const int row = qq / G::GW, col = qq - row * G::GW;
                const bool core = inq && row >= D && row < D + TH && col >= D && col < D + TW;
                f32x4 acc[NCT];                                  // biases enter as the accumulators' initial values
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{b1r[4 * ct], b1r[4 * ct + 1], b1r[4 * ct + 2], b1r[4 * ct + 3]};
                e16x8 unused;
                conv_taps<C, D, G::XW>(xs, row, col, g, A, acc, unused);
                unsigned char* gp = gs + (row * G::GW + col) * PB + 16 * (opiece ^ fswz<C>(col)) + obyte;
                const vec_t dq = *reinterpret_cast<const vec_t*>(gp);
                vec_t hq;
#pragma unroll
                for (int j = 0; j < NCH; ++j) hq[j] = (e16)elu_f(acc[j >> 2][j & 3]);
                float hv[NCH], gv[NCH];
                vec_t gq, aq;
                f32x4 z[NCT], u[NCT];
                if constexpr (C == 32) {
                    z[0] = mma32(A2[0], hq, f32x4{b2r[0], b2r[1], b2r[2], b2r[3]});
                    z[1] = mma32(A2[1], hq, f32x4{b2r[4], b2r[5], b2r[6], b2r[7]});
                } else {
                    z[0] = mma16(A2s, __builtin_bit_cast(s16x4, hq), f32x4{b2r[0], b2r[1], b2r[2], b2r[3]});
                }
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const float a2 = z[j >> 2][j & 3];
                    gv[j] = (float)dq[j] * elu_dpre(a2);
                    gq[j] = (e16)gv[j];
                    hv[j] = (float)hq[j];
                }
                if constexpr (C == 32) {
                    u[0] = mma32(A2T[0], gq, f32x4{0.f, 0.f, 0.f, 0.f});
                    u[1] = mma32(A2T[1], gq, f32x4{0.f, 0.f, 0.f, 0.f});
                } else {
                    u[0] = mma16(A2Ts, __builtin_bit_cast(s16x4, gq), f32x4{0.f, 0.f, 0.f, 0.f});
                }
#pragma unroll
                for (int j = 0; j < NCH; ++j) aq[j] = (e16)(u[j >> 2][j & 3] * elu_dout(hv[j]));
                if (inq) *reinterpret_cast<vec_t*>(gp) = aq;
                // sums over the tile's own pixels only (halo pixels belong to the neighbours; out-of-image ones are zero)
                const vec_t gm = core ? gq : zero_v;
#pragma unroll
                for (int j = 0; j < NCH; ++j) db2a[j] += core ? gv[j] : 0.f;
                if constexpr (C == 32) {
                    const uint2* s1 = reinterpret_cast<const uint2*>(&gm);
                    const uint2* s2 = reinterpret_cast<const uint2*>(&hq);
                    uint2* d1 = reinterpret_cast<uint2*>(tg + n * G::PS + 16 * g);
                    uint2* d2 = reinterpret_cast<uint2*>(thh + n * G::PS + 16 * g);
                    d1[0] = s1[0]; d1[1] = s1[1]; d2[0] = s2[0]; d2[1] = s2[1];
                } else {
                    *reinterpret_cast<uint2*>(tg + n * G::PS + 8 * g) = __builtin_bit_cast(uint2, gm);
                    *reinterpret_cast<uint2*>(thh + n * G::PS + 8 * g) = __builtin_bit_cast(uint2, hq);
                }
                // dW2[co][ci] += sum over the 16 pixels dA2[co][p] h1[ci][p]: the tiles read back transposed (the same wave
                // wrote them: LDS operations of one wave complete in order; the fences only stop the compiler)
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
                s16x4 ga[NCT], hb[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    ga[ct] = lds_tr16(tg + (4 * g + trj) * G::PS + (16 * ct + 4 * trq) * 2);
                    hb[ct] = lds_tr16(thh + (4 * g + trj) * G::PS + (16 * ct + 4 * trq) * 2);
                }
#pragma unroll
                for (int a = 0; a < NCT; ++a)
#pragma unroll
                    for (int c = 0; c < NCT; ++c) dw2[a][c] = mma16(ga[a], hb[c], dw2[a][c]);
                asm volatile("" ::: "memory");
            }
        }
        __syncthreads();

        // ---- phase 2a: dx = dy + W1^T (*) dA1 over the tile's own pixels; db1 from the centre tap ----
        {
            e16x8 A[NK][NCT];
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) A[k][ct] = wp[K::W3 + (k * NCT + ct) * 64 + lane];
            constexpr int GPRW = TW / 16;
            for (int grp = wave; grp < TH * GPRW; grp += NW) {
                const int r = grp / GPRW, c = (grp - r * GPRW) * 16 + n;
                const int h = h0 + r;
                if (h >= H) break;
                const int t = t0 + c;
                const bool valid = t < T;
                const long pix = ((long)b * H + h) * T + t;
                // unconditional (clamped) so that no branch pins a wait in front of the products
                const vec_t rq = *reinterpret_cast<const vec_t*>(dy + (valid ? pix : pix - (t - (T - 1))) * C + NCH * g);
                f32x4 acc[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                e16x8 centre;
                conv_taps<C, D, G::GW>(gs, r, c, g, A, acc, centre);
                vec_t cen;
                if constexpr (C == 32) cen = centre;
                else cen = *reinterpret_cast<const vec_t*>(gs + ((r + D) * G::GW + c + D) * PB + 16 * (opiece ^ fswz<C>(c + D)) + obyte);
                vec_t o;
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    db1a[j] += (float)cen[j];                    // out-of-image pixels hold zeros
                    o[j] = (e16)(acc[j >> 2][j & 3] + (float)rq[j]);
                }
                if (valid) *reinterpret_cast<vec_t*>(dx + pix * C + NCH * g) = o;
            }
        }

        // ---- phase 2b: dW1, K = pixels: 32 consecutive columns of a row per product, both operands by transpose reads ----
        for (int r = rpar; r < TH; r += RSTEP) {
            if (h0 + r >= H) break;
#pragma unroll
            for (int ch = ch0; ch < NCHK; ch += CHSTEP) {
                s16x4 lo, hi;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int cc = D + ch * 32 + 4 * g + trj + 16 * u;
                    const s16x4 t4 = lds_tr16(gs + ((r + D) * G::GW + cc) * PB +
                                              16 * (((C == 32 ? 2 * aw : 0) + (trq >> 1)) ^ fswz<C>(cc)) + 8 * (trq & 1));
                    if (u == 0) lo = t4; else hi = t4;
                }
                const e16x8 ga = __builtin_bit_cast(e16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int kh = k / 3, kw = k - 3 * kh;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int xc = D + kw * D + ch * 32 + 4 * g + trj + 16 * u;
                        const s16x4 t4 = lds_tr16(xs + ((r + D + kh * D) * G::XW + xc) * PB +
                                                  16 * (((C == 32 ? 2 * cit : 0) + (trq >> 1)) ^ fswz<C>(xc)) + 8 * (trq & 1));
                        if (u == 0) lo = t4; else hi = t4;
                    }
                    const e16x8 xq = __builtin_bit_cast(e16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    wacc[k] = mma32(ga, xq, wacc[k]);
                }
            }
        }
    }

    // ---- dumps: the weight-gradient accumulators per wave, everything else summed over the waves through LDS ----
    float* pw = part_w + ((long)blockIdx.x * NW + wave) * G::WDUMP;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[((k * NCT + aw) * 4 + r) * 64 + lane] = wacc[k][r];   // C = 32: the other co-tile's slots stay unwritten (and unread)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem) + wave * G::ADUMP;
#pragma unroll
    for (int a = 0; a < NCT; ++a)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((a * NCT + c) * 4 + r) * 64 + lane] = dw2[a][c][r];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        float s1 = db1a[j], s2 = db2a[j];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (n == 0) { red[C * C + NCH * g + j] = s1; red[C * C + C + NCH * g + j] = s2; }
    }
    __syncthreads();
    const float* all = reinterpret_cast<const float*>(smem);
    float* pa = part_a + (long)blockIdx.x * G::ADUMP;
    for (int i = tid; i < G::ADUMP; i += G::NTH) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += all[w * G::ADUMP + i];
        pa[i] = s;
    }
}

// ---- launchers -------------------------------------------------------------------------------------------------------
template <int C> constexpr long fused_scratch_bytes() {
    return (long)WK<C>::IMG_BYTES + ((long)MAX_W_WG * (C * C + 2 * C) + (long)MAX_W_WG * 4 * 9 * (C / 16) * 256) * 4;
}

template <int C, int D, int TH, int TW>
int launch_fused(const e16* x, const e16* dy, const float* w1, const float* b1, const float* w2, const float* b2, e16* dx,
                 float* dw1, float* db1, float* dw2, float* db2, unsigned char* ws, int B, int H, int T, hipStream_t st) {
    constexpr int NW = 4;
    using G = FB<C, D, TH, TW, NW>;
    using K = WK<C>;
    e16x8* wimg = reinterpret_cast<e16x8*>(ws);
    float* part_a = reinterpret_cast<float*>(ws + K::IMG_BYTES);
    float* part_w = part_a + (long)MAX_W_WG * G::ADUMP;
    if (!ttx_wprep_done) {                                       // inside tt_wide_level_bwd the images of all blocks were prepared in one launch
        WPrep wp1{{w1}, {w2}, {wimg}};
        hipLaunchKernelGGL(k_lvl_wprep<C>, dim3(K::NK * K::NCT + 1, 1), dim3(64), 0, st, wp1);
        TT_LAUNCH_CHECK();
    }
    static AttrOnce once;
    auto kern = k_wrb_bwd_fused<C, D, TH, TW, NW>;
    if (int rc = raise_lds(kern, G::LDS_BYTES, once)) return rc;
    const int tiles_h = (H + TH - 1) / TH, tiles_t = (T + TW - 1) / TW, ntiles = B * tiles_h * tiles_t;
    static const int per_cu = tt_tune("TTRAP_FBWD_PER_CU", C == 32 ? 2 : 3);        // registers: 2 waves per SIMD at C = 32, 3 at C = 16
    int grid = grid_for(ntiles, G::LDS_BYTES, per_cu);
    if (grid > MAX_W_WG) grid = MAX_W_WG;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), G::LDS_BYTES, st, x, dy, wimg, b1, b2, dx, part_a, part_w, B, H, T, tiles_h,
                       tiles_t, ntiles);
    TT_LAUNCH_CHECK();
    RedArgs ra{part_w, grid, part_a, grid, dw1, db1, dw2, db2, C == 32 ? 1 : 0};
    constexpr int total = 9 * C * C + C * C + 2 * C;
    return reduce_or_defer(k_wrb_reduce<C>, total, ra, st);
}

// Tile shapes (TTRAP_FBWD_TILE = 0 / 1 selects the alternative set, for tuning):
//   LDS per workgroup = x image (TH + 4D)(TW + 4D) + dy/dA1 image (TH + 2D)(TW + 2D) pixels of 2C bytes + transposition buffers.
template <int C>
int fused_c(const e16* x, const e16* dy, const float* w1, const float* b1, const float* w2, const float* b2, e16* dx,
            float* dw1, float* db1, float* dw2, float* db2, unsigned char* ws, int B, int H, int T, int D, hipStream_t st) {
    static const int alt = tt_tune("TTRAP_FBWD_TILE", 0);
#define TT_FB(DD, TH_, TW_) return launch_fused<C, DD, TH_, TW_>(x, dy, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st)
    if constexpr (C == 32) {
        switch (D) {
            case 1: if (alt) TT_FB(1, 8, 64); TT_FB(1, 8, 32);
            case 2: if (alt) TT_FB(2, 8, 64); TT_FB(2, 8, 32);
            case 3: if (alt) TT_FB(3, 4, 32); TT_FB(3, 8, 32);
        }
    } else {
        switch (D) {
            case 1: if (alt) TT_FB(1, 8, 32); TT_FB(1, 8, 64);
            case 2: if (alt) TT_FB(2, 8, 32); TT_FB(2, 8, 64);
            case 3: if (alt) TT_FB(3, 4, 64); TT_FB(3, 8, 32);
        }
    }
#undef TT_FB
    return TT_E_UNSUPPORTED;
}

// ==== one-pass backward, h1 saved, COLUMN STRIPS with a rolling ring of dA1 rows (round 4) ===================================
//   k_wrb_bwds<C,D,TH,TW>   the whole backward of a wide ResidualConv2dBlock from x, the SAVED h1 and dy in ONE pass, nothing
//   recomputed: dA1 = dL/d(conv1 pre-activation) exists in LDS only, and the 3x3 weight gradient is indexed by the pixel of x,
//   dW1[tap] = sum_q x[q] (x) dA1[q - tap D] (every image pixel q is visited once and dA1 is zero outside the image), so that x
//   needs no halo.  A first form with 8 x 32 TILES (k_wrb_bwd1, round 4, removed) redid the pointwise chain on a D halo all round
//   -- 1.33x / 1.69x / 2.08x the pixels at dilation 1 / 2 / 3 -- and lost to the per-stage kernels (C = 32: 0.514 / 0.510 /
//   0.554 ms against 0.469 / 0.442 / 0.477; C = 16: 0.473 / 0.473 / 0.508 against 0.470 / 0.435 / 0.443).  Here a workgroup owns a STRIP of TW columns over the
//   whole height and walks it downwards TH rows at a time; dA1 lives in a ring of TH + 2D image rows in LDS, so the rows a step
//   needs above it are the ones the previous steps computed -- every image row goes through the pointwise chain ONCE per strip and
//   only the column halo is redone (1.06x / 1.13x / 1.19x).  Step j:
//       a. dy rows [j TH + D - TH, j TH + D) -> the ring slots of the rows that just fell out of reach, h1 of the same rows -> a
//          staging image, both by LDS-DMA in the fswz layout (out-of-image pieces from the zero page: dy = 0 gives dA1 = 0, the
//          zero padding of the data gradient, above the image, below it and beside it);
//       b. pointwise chain on those rows, dA1 written over dy in the ring; db1 / db2 / dW2 over the strip's own columns;
//       c. the x rows [(j - 1) TH, j TH) start towards the staging image (h1 is dead), while
//       d. dx of those rows = dy + W1^T (*)_D dA1 from the ring, then
//       e. dW1[tap] += x[q] (x) dA1[q - tap D] over those rows (x needs no halo).
//   HBM traffic per block: h1, dy, x in, dx out, plus the column halo of h1 / dy (2D / TW).
#ifndef TT_BWDS_XE
#define TT_BWDS_XE 1
#endif
// The strip kernel's RING (dy, then dA1) has a placement of its own (round 6).  ds_read_b128 is serviced in four NON-contiguous 16-lane
// groups -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- not in four runs of sixteen lanes.  At C = 16
// a group of the data gradient's B-operand read therefore holds pixels n = 0-3, 12-15 of piece 0 and n = 4-11 of piece 1 (lane = 16 g + n,
// g & 1 = the 16-byte piece of the 32-byte pixel): with fswz's flip at column bit 3, pixels n and n + 8 of DIFFERENT pieces meet on one bank
// quad -- every such read took 8 LDS cycles instead of 4 (phase ablation + PMC, profiles/r06_bwds_ablation.txt: step d. owns 64 % of the kernel's
// bank-conflict cycles, 44 % of its own LDS cycles).  A flip at column bit 2 is conflict-free for that read at every column shift; the 8-byte
// reads of step b. (two groups of 32 lanes) then cost 4 cycles instead of 2 on the ring -- 2 cycles per 16 pixels against 20 saved.  The h1 / x
// staging images, which only 8-byte and transpose reads touch, keep fswz.  C = 32: fswz (not re-derived).  Measured: conflicts 17.1 M -> 7.2 M per
// launch, LDS-active cycles 56.7 M -> 46.7 M, time unchanged -- the LDS pipe (18 % busy) was never this kernel's bound.
#ifndef TT_BWDS_RSWZ
#define TT_BWDS_RSWZ 1
#endif
template <int C> __device__ __forceinline__ int rswz(int col) { return (C == 16 && TT_BWDS_RSWZ) ? ((col >> 2) & 1) : fswz<C>(col); }

#ifndef TT_BWDS_ABLATE
#define TT_BWDS_ABLATE 0        // MEASUREMENT ONLY (wrong results): bit 0 skips step b. (pointwise chain), bit 1 step d. (dx), bit 2 step e. (dW1) --
#endif                          // which step owns the strip kernel's LDS bank conflicts and its time (tools/r06_bwds_ablation.sh, profiles/r06_bwds_ablation.txt)
template <int C, int D, int TH, int TW> struct OS {
    static constexpr int CG = C / 8, PB = C * 2;
    static constexpr int GW = TW + 2 * D, RING = TH + 2 * D;
    static constexpr int ROWB = GW * PB;                         // bytes per image row
    static constexpr int RPP = GW * CG;                          // 16-byte pieces per image row
    static constexpr int IPR = (RPP + 63) / 64;                  // DMA wave-instructions per row
    static constexpr int RING_BYTES = (RING * ROWB + IPR * 1024 - RPP * 16 + 255) / 256 * 256;   // + the overrun of a row's last instruction
    static constexpr int XP = TH * TW * CG;                      // 16-byte pieces of the x rows
    static constexpr int HS_BYTES_ = TH * ROWB + IPR * 1024 - RPP * 16;
    static constexpr int HS_BYTES = ((HS_BYTES_ > XP * 16 ? HS_BYTES_ : XP * 16) + 255) / 256 * 256;
    static constexpr int PS = C * 2 + 8;
    static constexpr int T_BYTES = 4 * 2 * 16 * PS;
    static constexpr int ADUMP = C * C + 2 * C;
    static constexpr int WDUMP = 9 * (C / 16) * 256;
    // TT_BWDS_XE (C = 16): the x rows in an image of their own, staged in step a. with dy and h1 (they used to wait for h1 to die, step c.):
    // steps d. and e. then run back to back without the wait and the barrier between them, and the gated form reads its x from LDS
    static constexpr bool XE = TT_BWDS_XE != 0 && C == 16 && D <= 2;             // (dilation 3: 41 KB with it = three workgroups per CU instead of four)
    static constexpr int XS_BYTES = XE ? XP * 16 : 0;
    static constexpr int LDS_BYTES = RING_BYTES + HS_BYTES + XS_BYTES + T_BYTES;
    static constexpr int NG1 = (TH * GW + 15) / 16;
    static_assert(XP % NT == 0, "whole DMA instructions for the x rows");
    static_assert(4 * ADUMP * 4 <= RING_BYTES + HS_BYTES, "final dump reduction reuses the images");
    static_assert(TW % 32 == 0, "the K = 32 pixels of a weight-gradient product are 32 consecutive columns");
    static_assert(TH >= D, "a step must reach the rows the next one needs");
};

// 3x3 dilated product on one 16-pixel group out of the ring: `ro[kh]` = byte offset of the image row of tap row kh
template <int C, int D, int IW>
__device__ __forceinline__ void conv_taps_ring(const unsigned char* img, const int (&ro)[3], int col, int g,
                                               const e16x8 (&A)[WK<C>::NK][WK<C>::NCT], f32x4 (&acc)[WK<C>::NCT]) {
    constexpr int NK = WK<C>::NK, NCT = WK<C>::NCT, PB = C * 2;
    const int gsel = C == 32 ? g : (g & 1);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        int tap = C == 32 ? k : 2 * k + (g >> 1);
        if (tap > 8) tap = 8;                                    // the weights of the missing tenth tap are zero
        const int kh = tap / 3, kw = tap - 3 * kh;
        const int xc = col + kw * D;
        const int rb = C == 32 ? ro[k / 3] : (kh == 0 ? ro[0] : (kh == 1 ? ro[1] : ro[2]));
        const e16x8 bq = *reinterpret_cast<const e16x8*>(img + rb + xc * PB + 16 * (gsel ^ rswz<C>(xc)));
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = mma32(A[k][ct], bq, acc[ct]);
    }
}

// GOUT: dx leaves as dx * ELU'(x) for the layer in front of the level (k_nrb_bwd_fused, conv_wide_bf16.hip); the x rows of the step are
// still on their way into LDS at that point, so the lane's channels of x come from memory like its dy (the same lines the DMA is fetching).
// (Round 5, measured and dropped: h1 straight from memory into the lanes of step b. -- no staging image for it, the x rows into the freed
// image already in step a., three barriers per step instead of four: 53.18 / 52.62 ms per step against 51.96 / 52.06, profiles/r05_bwds_hd_ab.txt.)
// SJ (C = 16, with GOUT): the backward of a skip join on this block's input rides on the gated epilogue of step d. (SkipJ, wide_common.h)
// (The kernel's body is a device function so that the riding form is a kernel of its OWN signature, k_wrb_bwds_sj: one more -- unused --
// kernel argument on k_wrb_bwds itself moved its register allocation from 127 registers / no spill to 128 / one spilled.)
template <int C, int D, int TH, int TW, int MINW, bool GOUT, bool SJ>
__device__ __forceinline__ void wrb_bwds_body(const e16* __restrict__ x, const e16* __restrict__ h1, const e16* __restrict__ dy,
                                              const e16x8* __restrict__ wimg, const float* __restrict__ b2, e16* __restrict__ dx,
                                              float* __restrict__ part_a, float* __restrict__ part_w, int B, int H, int T,
                                              int tiles_t, int nstrips, const SkipJ& sj) {
    static_assert(!SJ || (GOUT && C == 16), "a skip join rides on the gated epilogue of the C = 16 strips only");
    float sjw = 0.f, sjdot = 0.f;
    if constexpr (SJ) sjw = sj.w ? sj.w[0] : 1.f;
    using G = OS<C, D, TH, TW>;
    using K = WK<C>;
    constexpr int NCT = K::NCT, NK = K::NK, NCH = K::NCH, PB = G::PB, GW = G::GW, RING = G::RING;
    typedef typename std::conditional<C == 32, e16x8, e16x4>::type vec_t;     // a lane's channels of one pixel
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* ring = smem;                                  // dy, then dA1: RING image rows of GW pixels
    unsigned char* hst = smem + G::RING_BYTES;                   // h1 of the step's new rows, then the step's x rows
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4, trj = n >> 2, trq = n & 3;
    constexpr bool XE = G::XE;
    unsigned char* xst = XE ? smem + G::RING_BYTES + G::HS_BYTES : hst;          // the step's x rows (XE: an image of their own)
    unsigned char* tg = smem + G::RING_BYTES + G::HS_BYTES + G::XS_BYTES + wave * (2 * 16 * G::PS);   // this wave's dA2 tile, then its h1 tile
    unsigned char* thh = tg + 16 * G::PS;
    const int opiece = C == 32 ? g : (g >> 1), obyte = C == 32 ? 0 : 8 * (g & 1);

    constexpr int NCHK = TW / 32;
    static_assert(C == 32 || (4 % NCHK == 0), "wave roles");
    const int cit = C == 32 ? (wave & 1) : 0, aw = C == 32 ? (wave >> 1) : 0;
    const int ch0 = C == 32 ? 0 : wave % NCHK, rpar = C == 32 ? 0 : wave / NCHK;
    constexpr int CHSTEP = C == 32 ? 1 : NCHK, RSTEP = C == 32 ? 1 : 4 / NCHK;

    f32x4 wacc[9], dw2[NCT][NCT];
#pragma unroll
    for (int k = 0; k < 9; ++k) wacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < NCT; ++a)
#pragma unroll
        for (int c = 0; c < NCT; ++c) dw2[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db1a[NCH], db2a[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) { db1a[j] = 0.f; db2a[j] = 0.f; }
    vec_t zero_v;
#pragma unroll
    for (int j = 0; j < NCH; ++j) zero_v[j] = (e16)0.f;

    const e16* zero = reinterpret_cast<const e16*>(&g_wzero16);
    static_assert((TW * G::CG) % 64 == 0, "a DMA wave-instruction of the x rows stays inside one row");
    int coff[G::IPR], roff[G::IPR], xoff[G::XP / NT];           // per-lane element offsets of the staged pieces inside their row (h1 | ring | x)
#pragma unroll
    for (int part = 0; part < G::IPR; ++part) {
        const int p = part * 64 + lane, px = p / G::CG, sgrp = p - px * G::CG;
        coff[part] = px * C + (sgrp ^ fswz<C>(px)) * 8;
        roff[part] = px * C + (sgrp ^ rswz<C>(px)) * 8;
    }
#pragma unroll
    for (int it = 0; it < G::XP / NT; ++it) {
        const int p = it * NT + wave * 64 + lane, q = p / G::CG, sgrp = p - q * G::CG, px = q % TW;
        xoff[it] = px * C + (sgrp ^ fswz<C>(px)) * 8;
    }
    const int nsteps = (H + TH - 1) / TH + 1;
    // ring slot of image row R (R >= -RING): (R + RING) mod RING, on the scalar unit where R is wave-uniform
    auto slot = [&](int R) -> int { return (R + RING) % RING; };

    for (int v = blockIdx.x; v < nstrips; v += gridDim.x) {
        const int strip = xcd_order(v, nstrips);
        const int b = strip / tiles_t, t0 = (strip - b * tiles_t) * TW;
        const long ib = (long)b * H * T * C;
        const bool edge = t0 - D < 0 || t0 + TW + D > T;         // the strip's halo crosses the left / right image border

        for (int j = 0; j < nsteps; ++j) {
            const int N0 = j * TH + D - TH;                      // first new dA1 row of this step
            const int X0 = (j - 1) * TH;                         // first row whose dx / dW1 this step produces
            __syncthreads();                                     // the previous step has been consumed
            // ---- a. dy -> ring slots of the new rows, h1 -> staging; one image row per IPR wave-instructions ----
            // A wave stages whole rows (rl = wave, wave + 4, ..); the row is wave-uniform, so its in-image test and its base address
            // are scalar, and away from the left / right image border the per-lane part of the address is the constant coff[]
            // (one 32-bit add per piece instead of ~20 vector instructions of index arithmetic and bounds tests).
            for (int rl = wave; rl < TH; rl += 4) {
                const int h = N0 + rl;
                unsigned char* rdst = ring + slot(h) * G::ROWB;
                unsigned char* hdst = hst + rl * G::ROWB;
                if ((unsigned)h >= (unsigned)H) {                // a row above / below the image: zero page (dy = 0 -> dA1 = 0)
#pragma unroll
                    for (int part = 0; part < G::IPR; ++part)
                        if (part * 64 + lane < G::RPP) { glds16(zero, rdst + part * 1024); glds16(zero, hdst + part * 1024); }
                } else if (!edge) {
                    const long rowoff = ib + ((long)h * T + (t0 - D)) * C;
                    const e16* gr_ = dy + rowoff;
                    const e16* hr_ = h1 + rowoff;
#pragma unroll
                    for (int part = 0; part < G::IPR; ++part)
                        if (part * 64 + lane < G::RPP) { glds16(gr_ + roff[part], rdst + part * 1024); glds16_h1(hr_ + coff[part], hdst + part * 1024); }
                } else {
#pragma unroll
                    for (int part = 0; part < G::IPR; ++part) {
                        const int p = part * 64 + lane;
                        const int px = p / G::CG;
                        const int t = t0 - D + px;
                        const bool ok = (unsigned)t < (unsigned)T;
                        const long off = ib + ((long)h * T + (t0 - D)) * C;
                        if (p < G::RPP) { glds16(ok ? dy + off + roff[part] : zero, rdst + part * 1024); glds16_h1(ok ? h1 + off + coff[part] : zero, hdst + part * 1024); }
                    }
                }
            }
            // the x rows of this step's dx / dW1 (a wave instruction = 64 pieces of ONE row: TW * CG is a multiple of 64)
            auto stage_x = [&]() {
#pragma unroll
                for (int it = 0; it < G::XP / NT; ++it) {
                    const int i = it * NT + wave * 64;
                    const int row = i / (TW * G::CG);
                    const int h = X0 + row;
                    const int px = ((i + lane) / G::CG) % TW;
                    const bool ok = h < H && (!edge || t0 + px < T);
                    glds16_x(ok ? x + ib + ((long)h * T + t0) * C + xoff[it] : zero, xst + (long)i * 16);
                }
            };
            if constexpr (XE) { if (j > 0) stage_x(); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            const e16x8* wp = wimg;
            const float* b2p = b2;
            asm volatile("" : "+s"(wp), "+s"(b2p));

            // ---- b. pointwise chain on the new rows, dA1 over dy in the ring; db1 / db2 / dW2 over the strip's own columns ----
            if (!(TT_BWDS_ABLATE & 1) && N0 < H) {               // rows below the image: dy = 0 staged, dA1 = 0 already
                e16x8 A2[NCT], A2T[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    A2[ct] = wp[2 * K::W3 + ct * 64 + lane];
                    A2T[ct] = wp[2 * K::W3 + NCT * 64 + ct * 64 + lane];
                }
                const s16x4 A2s = __builtin_bit_cast(s16x4, __builtin_shufflevector(A2[0], A2[0], 0, 1, 2, 3));      // C = 16: K = 16
                const s16x4 A2Ts = __builtin_bit_cast(s16x4, __builtin_shufflevector(A2T[0], A2T[0], 0, 1, 2, 3));
                float b2r[NCH];
#pragma unroll
                for (int jj = 0; jj < NCH; ++jj) b2r[jj] = b2p[NCH * g + jj];
                const int s0 = slot(N0);
                for (int grp = wave; grp < G::NG1; grp += 4) {
                    const int q = grp * 16 + n;
                    const bool inq = q < TH * GW;
                    const int qq = inq ? q : TH * GW - 1;
                    const int rl = qq / GW, col = qq - rl * GW;
                    int sl = s0 + rl;
                    sl = sl >= RING ? sl - RING : sl;
                    const bool core = inq && col >= D && col < D + TW;       // halo columns belong to the neighbouring strips
                    const int po = col * PB + 16 * (opiece ^ fswz<C>(col)) + obyte;            // in the h1 staging image
                    const int pr = col * PB + 16 * (opiece ^ rswz<C>(col)) + obyte;            // in the ring
                    unsigned char* gp = ring + sl * G::ROWB + pr;
                    const vec_t hq = *reinterpret_cast<const vec_t*>(hst + rl * G::ROWB + po);
                    const vec_t dq = *reinterpret_cast<const vec_t*>(gp);
                    float hv[NCH], gv[NCH], a1g[NCH];
                    vec_t gq, aq;
                    f32x4 z[NCT], u[NCT];
                    if constexpr (C == 32) {
                        z[0] = mma32(A2[0], hq, f32x4{b2r[0], b2r[1], b2r[2], b2r[3]});
                        z[1] = mma32(A2[1], hq, f32x4{b2r[4], b2r[5], b2r[6], b2r[7]});
                    } else {
                        z[0] = mma16(A2s, __builtin_bit_cast(s16x4, hq), f32x4{b2r[0], b2r[1], b2r[2], b2r[3]});
                    }
#pragma unroll
                    for (int jj = 0; jj < NCH; ++jj) {
                        const float a2 = z[jj >> 2][jj & 3];
                        gv[jj] = (float)dq[jj] * elu_dpre(a2);
                        gq[jj] = (e16)gv[jj];
                        hv[jj] = (float)hq[jj];
                    }
                    if constexpr (C == 32) {
                        u[0] = mma32(A2T[0], gq, f32x4{0.f, 0.f, 0.f, 0.f});
                        u[1] = mma32(A2T[1], gq, f32x4{0.f, 0.f, 0.f, 0.f});
                    } else {
                        u[0] = mma16(A2Ts, __builtin_bit_cast(s16x4, gq), f32x4{0.f, 0.f, 0.f, 0.f});
                    }
                    const float cm = core ? 1.f : 0.f;
#pragma unroll
                    for (int jj = 0; jj < NCH; ++jj) {
                        a1g[jj] = u[jj >> 2][jj & 3] * elu_dout(hv[jj]);
                        aq[jj] = (e16)a1g[jj];
                        db2a[jj] += cm * gv[jj]; db1a[jj] += cm * a1g[jj];
                    }
                    if (inq) *reinterpret_cast<vec_t*>(gp) = aq;
                    const vec_t gm = core ? gq : zero_v;
                    if constexpr (C == 32) {
                        const uint2* s1 = reinterpret_cast<const uint2*>(&gm);
                        const uint2* s2 = reinterpret_cast<const uint2*>(&hq);
                        uint2* d1 = reinterpret_cast<uint2*>(tg + n * G::PS + 16 * g);
                        uint2* d2 = reinterpret_cast<uint2*>(thh + n * G::PS + 16 * g);
                        d1[0] = s1[0]; d1[1] = s1[1]; d2[0] = s2[0]; d2[1] = s2[1];
                    } else {
                        *reinterpret_cast<uint2*>(tg + n * G::PS + 8 * g) = __builtin_bit_cast(uint2, gm);
                        *reinterpret_cast<uint2*>(thh + n * G::PS + 8 * g) = __builtin_bit_cast(uint2, hq);
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
                    s16x4 ga[NCT], hb[NCT];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        ga[ct] = lds_tr16(tg + (4 * g + trj) * G::PS + (16 * ct + 4 * trq) * 2);
                        hb[ct] = lds_tr16(thh + (4 * g + trj) * G::PS + (16 * ct + 4 * trq) * 2);
                    }
#pragma unroll
                    for (int a = 0; a < NCT; ++a)
#pragma unroll
                        for (int c = 0; c < NCT; ++c) dw2[a][c] = mma16(ga[a], hb[c], dw2[a][c]);
                    asm volatile("" ::: "memory");
                }
            }
            if (j == 0) continue;                                // nothing above the image to produce
            __syncthreads();                                     // dA1 rows complete, h1 staging dead

            // ---- c. (!XE) the x rows of this step towards the staging image (consumed in e.) ----
            if constexpr (!XE) stage_x();

            // ---- d. dx = dy + W1^T (*) dA1 over the step's rows ----
            // (Round 6, measured and dropped: TWO pixel groups per iteration at C = 16 -- two independent chains of five matrix products instead of
            //  one per wave: 13-27 registers spilled at the 128-register cap, 0.573 / 0.526 / 0.514 ms per call against 0.418 / 0.395 / 0.393; at three
            //  waves per SIMD without spills 0.533 / 0.491 / 0.486; train step 53.4 / 52.5-52.8 against 50.4-50.6 ms -- profiles/r06_bwds_ablation.txt.)
            if constexpr ((TT_BWDS_ABLATE & 2) != 0) {
            } else if constexpr (C == 32) {
                // C = 32: a wave computes ONE co-tile (16 of the 32 output channels: for a lane the four consecutive channels
                // 8 g + 4 ct ..) of twice as many pixel groups -- 36 registers of weights instead of 72, which is what brings the
                // kernel under 168 registers (a third workgroup per CU); the B operands are read twice from LDS instead.
                const int ctd = wave & 1;
                e16x8 A1[NK];
#pragma unroll
                for (int k = 0; k < NK; ++k) A1[k] = wp[K::W3 + (k * NCT + ctd) * 64 + lane];
                constexpr int GPRW = TW / 16;
                for (int grp = wave >> 1; grp < TH * GPRW; grp += 2) {
                    const int r = grp / GPRW, c = (grp - r * GPRW) * 16 + n;
                    const int h = X0 + r;
                    if (h >= H) break;
                    const int t = t0 + c;
                    const bool valid = t < T;
                    const long pix = ((long)b * H + h) * T + t;
                    const e16x4 rq = *reinterpret_cast<const e16x4*>(dy + (valid ? pix : pix - (t - (T - 1))) * C + 8 * g + 4 * ctd);
                    e16x4 xg;
                    if constexpr (GOUT && XE) xg = *reinterpret_cast<const e16x4*>(xst + (r * TW + c) * PB + 16 * (g ^ fswz<C>(c)) + 8 * ctd);
                    else if constexpr (GOUT) xg = *reinterpret_cast<const e16x4*>(x + (valid ? pix : pix - (t - (T - 1))) * C + 8 * g + 4 * ctd);
                    const int ro[3] = {slot(h - D) * G::ROWB, slot(h) * G::ROWB, slot(h + D) * G::ROWB};
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        const int kh = k / 3, kw = k - 3 * kh;
                        const int xc = c + kw * D;
                        const e16x8 bq = *reinterpret_cast<const e16x8*>(ring + ro[kh] + xc * PB + 16 * (g ^ rswz<C>(xc)));
                        acc = mma32(A1[k], bq, acc);
                    }
                    e16x4 o;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) o[jj] = GOUT ? (e16)((acc[jj] + (float)rq[jj]) * elu_dout((float)xg[jj])) : (e16)(acc[jj] + (float)rq[jj]);
                    if (valid) *reinterpret_cast<e16x4*>(dx + pix * C + 8 * g + 4 * ctd) = o;
                }
            } else {
                e16x8 A[NK][NCT];
#pragma unroll
                for (int k = 0; k < NK; ++k)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) A[k][ct] = wp[K::W3 + (k * NCT + ct) * 64 + lane];
                constexpr int GPRW = TW / 16;
                for (int grp = wave; grp < TH * GPRW; grp += 4) {
                    const int r = grp / GPRW, c = (grp - r * GPRW) * 16 + n;
                    const int h = X0 + r;
                    if (h >= H) break;
                    const int t = t0 + c;
                    const bool valid = t < T;
                    const long pix = ((long)b * H + h) * T + t;
                    // unconditional (clamped) so that no branch pins a wait in front of the products
                    const vec_t rq = *reinterpret_cast<const vec_t*>(dy + (valid ? pix : pix - (t - (T - 1))) * C + NCH * g);
                    vec_t xg, sg0, sg1;
                    if constexpr (GOUT && XE) xg = *reinterpret_cast<const vec_t*>(xst + (r * TW + c) * PB + 16 * (opiece ^ fswz<C>(c)) + obyte);
                    else if constexpr (GOUT) xg = *reinterpret_cast<const vec_t*>(x + (valid ? pix : pix - (t - (T - 1))) * C + NCH * g);
                    if constexpr (SJ) {                          // the join's gradient at this pixel, both batches: requested with dy
                        const e16* gp = sj.g + (valid ? pix : pix - (t - (T - 1))) * C + NCH * g;
                        sg0 = *reinterpret_cast<const vec_t*>(gp);
                        sg1 = *reinterpret_cast<const vec_t*>(gp + sj.half);
                    }
                    const int ro[3] = {slot(h - D) * G::ROWB, slot(h) * G::ROWB, slot(h + D) * G::ROWB};
                    f32x4 acc[NCT];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                    conv_taps_ring<C, D, GW>(ring, ro, c, g, A, acc);
                    vec_t o;
                    if constexpr (SJ) {
                        float add[NCH], dot = 0.f;
                        skipj_terms<vec_t, NCH>(sg0, sg1, sj.half != 0, xg, sjw, add, dot);
                        if (valid) sjdot += dot;
#pragma unroll
                        for (int jj = 0; jj < NCH; ++jj) o[jj] = (e16)((acc[jj >> 2][jj & 3] + (float)rq[jj] + add[jj]) * elu_dout((float)xg[jj]));
                    } else {
#pragma unroll
                        for (int jj = 0; jj < NCH; ++jj)
                            o[jj] = GOUT ? (e16)((acc[jj >> 2][jj & 3] + (float)rq[jj]) * elu_dout((float)xg[jj])) : (e16)(acc[jj >> 2][jj & 3] + (float)rq[jj]);
                    }
                    if (valid) *reinterpret_cast<vec_t*>(dx + pix * C + NCH * g) = o;
                }
            }
            if constexpr (!XE) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the x rows have landed
                __syncthreads();
            }

            // ---- e. dW1[tap] += x[q] (x) dA1[q - tap D] over the step's rows, K = 32 consecutive columns per product ----
            for (int r = rpar; r < ((TT_BWDS_ABLATE & 4) ? 0 : TH); r += RSTEP) {
                const int h = X0 + r;
                if (h >= H) break;
                const int ro[3] = {slot(h + D) * G::ROWB, slot(h) * G::ROWB, slot(h - D) * G::ROWB};   // tap row kh reads dA1 row h + (1 - kh) D
#pragma unroll
                for (int ch = ch0; ch < NCHK; ch += CHSTEP) {
                    s16x4 lo, hi;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int xc = ch * 32 + 4 * g + trj + 16 * u;
                        const s16x4 t4 = lds_tr16(xst + (r * TW + xc) * PB + 16 * (((C == 32 ? 2 * cit : 0) + (trq >> 1)) ^ fswz<C>(xc)) + 8 * (trq & 1));
                        if (u == 0) lo = t4; else hi = t4;
                    }
                    const e16x8 xq = __builtin_bit_cast(e16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const int kh = k / 3, kw = k - 3 * kh;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int cc = (2 - kw) * D + ch * 32 + 4 * g + trj + 16 * u;
                            const s16x4 t4 = lds_tr16(ring + ro[kh] + cc * PB +
                                                      16 * (((C == 32 ? 2 * aw : 0) + (trq >> 1)) ^ rswz<C>(cc)) + 8 * (trq & 1));
                            if (u == 0) lo = t4; else hi = t4;
                        }
                        const e16x8 ga = __builtin_bit_cast(e16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                        wacc[k] = mma32(ga, xq, wacc[k]);
                    }
                }
            }
        }
    }

    if constexpr (SJ) skipj_finish(sjdot, sj);
    // ---- dumps: the weight-gradient accumulators (C = 32: per wave, each holds its own (ci-tile, co-tile); C = 16: the four waves hold
    //      the same elements and are summed through LDS -- one dump per workgroup, the reduce reads a quarter of the bytes:
    //      RedArgs::one_dump), everything else summed over the waves through LDS ----
    if constexpr (C == 32) {
        float* pw = part_w + ((long)blockIdx.x * 4 + wave) * G::WDUMP;
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) pw[((k * NCT + aw) * 4 + r) * 64 + lane] = wacc[k][r];   // only this wave's co-tile (RedArgs::split_a)
        __syncthreads();
    } else {
        static_assert(C == 32 || 2 * G::WDUMP * 4 <= G::LDS_BYTES, "two waves' accumulators fit the images");
        __syncthreads();
        float* wr = reinterpret_cast<float*>(smem);              // two slots: (wave 2 + wave 0), (wave 3 + wave 1), then their sum
        if (wave >= 2) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) wr[(wave - 2) * G::WDUMP + (k * 4 + r) * 64 + lane] = wacc[k][r];
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) wr[wave * G::WDUMP + (k * 4 + r) * 64 + lane] += wacc[k][r];
        }
        __syncthreads();
        float* pw = part_w + (long)blockIdx.x * 4 * G::WDUMP;
        for (int i = tid; i < G::WDUMP; i += NT) pw[i] = wr[i] + wr[G::WDUMP + i];
        __syncthreads();
    }
    float* red = reinterpret_cast<float*>(smem) + wave * G::ADUMP;
#pragma unroll
    for (int a = 0; a < NCT; ++a)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((a * NCT + c) * 4 + r) * 64 + lane] = dw2[a][c][r];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        float s1 = db1a[j], s2 = db2a[j];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (n == 0) { red[C * C + NCH * g + j] = s1; red[C * C + C + NCH * g + j] = s2; }
    }
    __syncthreads();
    const float* all = reinterpret_cast<const float*>(smem);
    float* pa = part_a + (long)blockIdx.x * G::ADUMP;
    for (int i = tid; i < G::ADUMP; i += NT) pa[i] = (all[i] + all[G::ADUMP + i]) + (all[2 * G::ADUMP + i] + all[3 * G::ADUMP + i]);
}

template <int C, int D, int TH, int TW, int MINW, bool GOUT = false>
__global__ __launch_bounds__(NT, MINW) void k_wrb_bwds(const e16* __restrict__ x, const e16* __restrict__ h1, const e16* __restrict__ dy,
                                                     const e16x8* __restrict__ wimg, const float* __restrict__ b2, e16* __restrict__ dx,
                                                     float* __restrict__ part_a, float* __restrict__ part_w, int B, int H, int T,
                                                     int tiles_t, int nstrips) {
    wrb_bwds_body<C, D, TH, TW, MINW, GOUT, false>(x, h1, dy, wimg, b2, dx, part_a, part_w, B, H, T, tiles_t, nstrips, SkipJ());
}
template <int C, int D, int TH, int TW, int MINW>
__global__ __launch_bounds__(NT, MINW) void k_wrb_bwds_sj(const e16* __restrict__ x, const e16* __restrict__ h1, const e16* __restrict__ dy,
                                                            const e16x8* __restrict__ wimg, const float* __restrict__ b2, e16* __restrict__ dx,
                                                            float* __restrict__ part_a, float* __restrict__ part_w, int B, int H, int T,
                                                            int tiles_t, int nstrips, SkipJ sj) {
    wrb_bwds_body<C, D, TH, TW, MINW, true, true>(x, h1, dy, wimg, b2, dx, part_a, part_w, B, H, T, tiles_t, nstrips, sj);
}

template <int C, int D, int TH, int TW>
int launch_bwds(const e16* x, const e16* h1, const e16* dy, const float* w1, const float* w2, const float* b2, e16* dx,
                float* dw1, float* db1, float* dw2, float* db2, unsigned char* ws, int B, int H, int T, hipStream_t st) {
    using G = OS<C, D, TH, TW>;
    using K = WK<C>;
    e16x8* wimg = reinterpret_cast<e16x8*>(ws);
    float* part_a = reinterpret_cast<float*>(ws + K::IMG_BYTES);
    float* part_w = part_a + (long)MAX_W_WG * G::ADUMP;
    if (!ttx_wprep_done) {                                       // inside tt_wide_level_bwd the images of all blocks were prepared in one launch
        WPrep wp1{{w1}, {w2}, {wimg}};
        hipLaunchKernelGGL(k_lvl_wprep<C>, dim3(K::NK * K::NCT + 1, 1), dim3(64), 0, st, wp1);
        TT_LAUNCH_CHECK();
    }
    constexpr int MINW = C == 32 ? 3 : 4;
    const int tiles_t = (T + TW - 1) / TW, nstrips = B * tiles_t;
    static const int per_cu = tt_tune("TTRAP_BWDS_PER_CU", MINW);
    int grid = grid_for(nstrips, G::LDS_BYTES, per_cu);
    if (grid > MAX_W_WG) grid = MAX_W_WG;
    bool gated = false;
    if constexpr (D == 1) gated = ttx_gate_dx == 1;              // a level's first block: dx leaves gated
    if (gated) {
        if constexpr (D == 1) {
            bool took = false;
            if constexpr (C == 16) {
                if (ttx_skip.g) {
                    static AttrOnce once_j;
                    auto kj = k_wrb_bwds_sj<C, D, TH, TW, MINW>;
                    if (int rc = raise_lds(kj, G::LDS_BYTES, once_j)) return rc;
                    hipLaunchKernelGGL(kj, dim3(grid), dim3(NT), G::LDS_BYTES, st, x, h1, dy, wimg, b2, dx, part_a, part_w, B, H, T, tiles_t, nstrips, ttx_skip);
                    ttx_skip.g = nullptr;
                    took = true;
                }
            }
            if (!took) {
                static AttrOnce once_g;
                auto kg = k_wrb_bwds<C, D, TH, TW, MINW, true>;
                if (int rc = raise_lds(kg, G::LDS_BYTES, once_g)) return rc;
                hipLaunchKernelGGL(kg, dim3(grid), dim3(NT), G::LDS_BYTES, st, x, h1, dy, wimg, b2, dx, part_a, part_w, B, H, T, tiles_t, nstrips);
            }
            ttx_gate_dx = 2;
        }
    } else {
        static AttrOnce once;
        auto kern = k_wrb_bwds<C, D, TH, TW, MINW, false>;
        if (int rc = raise_lds(kern, G::LDS_BYTES, once)) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), G::LDS_BYTES, st, x, h1, dy, wimg, b2, dx, part_a, part_w, B, H, T, tiles_t, nstrips);
    }
    TT_LAUNCH_CHECK();
    RedArgs ra{part_w, grid, part_a, grid, dw1, db1, dw2, db2, C == 32 ? 1 : 0, C == 32 ? 0 : 1};
    constexpr int total = 9 * C * C + C * C + 2 * C;
    return reduce_or_defer(k_wrb_reduce<C>, total, ra, st);
}

template <int C>
int bwds_c(const e16* x, const e16* h1, const e16* dy, const float* w1, const float* w2, const float* b2, e16* dx,
           float* dw1, float* db1, float* dw2, float* db2, unsigned char* ws, int B, int H, int T, int D, hipStream_t st) {
#define TT_BS(DD, TH_, TW_) return launch_bwds<C, DD, TH_, TW_>(x, h1, dy, w1, w2, b2, dx, dw1, db1, dw2, db2, ws, B, H, T, st)
    switch (D) {
        // 8-row steps everywhere.  C = 32 (round 4): 6-row steps at dilation 2, 3 fit a third workgroup per CU (46.9 / 53.3 KB of LDS) and
        // changed nothing at dilation 2 (0.458 ms either way) and lost at dilation 3 (0.637 vs 0.453 ms: the third workgroup does not
        // become resident at 53.3 KB); 16- and 4-row steps lose at both widths (profiles/r04_bwds_ab.txt).
        case 1: TT_BS(1, 8, 32);
        case 2: TT_BS(2, 8, 32);
        case 3: TT_BS(3, 8, 32);
    }
#undef TT_BS
    return TT_E_UNSUPPORTED;
}

inline bool fshape_ok(int B, int C, int H, int T) {
    return B > 0 && H > 0 && T > 0 && (C == 16 || C == 32) && (long)H * T * C < (1l << 31);
}

}  // namespace

// called by tt_wide_level_bwd (conv_wide_bf16.hip): the bf16 weight images of n blocks (image i at ws_i, where the one-pass kernels expect
// it) in ONE launch; not part of the C ABI
int ttx_wide_wprep_batch(int C, int n, const float* const* w1, const float* const* w2, void* const* ws, hipStream_t st) {
    if (n < 1 || n > 4) return TT_E_BADARG;
    WPrep wp{};
    for (int i = 0; i < n; ++i) { wp.w1[i] = w1[i]; wp.w2[i] = w2[i]; wp.img[i] = reinterpret_cast<e16x8*>(ws[i]); }
    if (C == 16) hipLaunchKernelGGL(k_lvl_wprep<16>, dim3(WK<16>::NK * WK<16>::NCT + 1, n), dim3(64), 0, st, wp);
    else if (C == 32) hipLaunchKernelGGL(k_lvl_wprep<32>, dim3(WK<32>::NK * WK<32>::NCT + 1, n), dim3(64), 0, st, wp);
    else return TT_E_UNSUPPORTED;
    TT_LAUNCH_CHECK();
    return 0;
}

extern "C" {

int64_t tt_wide_fused_scratch_bytes(int C) {
    return C == 16 ? fused_scratch_bytes<16>() : C == 32 ? fused_scratch_bytes<32>() : -1;
}

int tt_wide_rb_bwd_fused(const void* x, const void* dy, const float* w1, const float* b1, const float* w2, const float* b2, void* dx,
                         float* dw1, float* db1, float* dw2, float* db2, void* ws, int B, int C, int H, int T, int dilation,
                         void* stream) {
    if (!x || !dy || !w1 || !b1 || !w2 || !b2 || !dx || !dw1 || !db1 || !dw2 || !db2 || !ws) return TT_E_BADARG;
    if (C != 16 && C != 32) return TT_E_UNSUPPORTED;
    if (!fshape_ok(B, C, H, T)) return TT_E_BADARG;
    const e16 *xi = (const e16*)x, *gi = (const e16*)dy;
    hipStream_t st = tt_stream(stream);
    if (C == 16) return fused_c<16>(xi, gi, w1, b1, w2, b2, (e16*)dx, dw1, db1, dw2, db2, (unsigned char*)ws, B, H, T, dilation, st);
    return fused_c<32>(xi, gi, w1, b1, w2, b2, (e16*)dx, dw1, db1, dw2, db2, (unsigned char*)ws, B, H, T, dilation, st);
}

int64_t tt_wide_onepass_scratch_bytes(int C) { return tt_wide_fused_scratch_bytes(C); }

int tt_wide_rb_bwd_onepass(const void* x, const void* h1, const void* dy, const float* w1, const float* w2, const float* b2, void* dx,
                           float* dw1, float* db1, float* dw2, float* db2, void* ws, int B, int C, int H, int T, int dilation,
                           void* stream) {
    if (!x || !h1 || !dy || !w1 || !w2 || !b2 || !dx || !dw1 || !db1 || !dw2 || !db2 || !ws) return TT_E_BADARG;
    if (C != 16 && C != 32) return TT_E_UNSUPPORTED;
    if (!fshape_ok(B, C, H, T)) return TT_E_BADARG;
    const e16 *xi = (const e16*)x, *hi = (const e16*)h1, *gi = (const e16*)dy;
    hipStream_t st = tt_stream(stream);
    if (C == 16) return bwds_c<16>(xi, hi, gi, w1, w2, b2, (e16*)dx, dw1, db1, dw2, db2, (unsigned char*)ws, B, H, T, dilation, st);
    return bwds_c<32>(xi, hi, gi, w1, w2, b2, (e16*)dx, dw1, db1, dw2, db2, (unsigned char*)ws, B, H, T, dilation, st);
}

}  // extern "C"
