// bf16 channel-innermost (4,1) strided / transposed layers between the levels (reference modules.py:626-630, 685-689).
//
//   EncoderBlock.sconv   y[ho,t,co] = ELU(b[co] + sum_{kh<4,ci} W[co][ci][kh] x[2ho+kh,t,ci])            C -> 2C, H -> H/2 - 1
//   DecoderBlock.tconv   y[h,t,c]   = ELU(b[c]  + sum_{ci<2C, kh = h (mod 2)} W[ci][c][kh] x[(h-kh)/2,t,ci])   2C -> C, H -> 2H+2+p
// Both weights are (2C, C, 4, 1) tensors; with index(wide, narrow, kh) = (wide * C + narrow) * 4 + kh they are the SAME
// array layout, and the four products of the two layers are two kernels:
//   k_s4  "gather four rows"   out[ho] <- in rows 2ho..2ho+3, C -> 2C channels:  sconv forward, tconv data gradient
//   k_p2  "two rows by parity" out[h]  <- in rows (h - kh)/2,  2C -> C channels:  tconv forward, sconv data gradient
// Everything is pointwise along T, so the B operand of v_mfma_f32_16x16x32_bf16 (one lane = one pixel, eight channels)
// comes straight from HBM as 16-byte loads -- no LDS, no halo.  In the data-gradient use the operand is gated on the fly,
// g = dy * ELU'(y) from the saved OUTPUT y, so no separate gating pass and no scratch tensor.
//   k_w4  weight gradient  dW[a][b][kh] = sum small[r,t,a] * big[2r+kh,t,b]  (+ the bias gradient = sum of the gated
//         operand): K = pixels; both operands from channel-innermost LDS images by transpose reads (ds_read_b64_tr_b16),
//         the gated one staged through registers, the other by LDS-DMA.  Per-wave register dumps, summed by k_w4_reduce.
// fp32 master weights rounded to bf16 into registers; fp32 accumulation, bias, ELU; bf16 activations and gradients.
#include "bf16_common.h"

namespace {

#ifndef W4_WAVES16
#define W4_WAVES16 3       // waves per SIMD the strided weight-gradient kernels at C <= 16 must allow (three workgroups per CU)
#endif

__device__ __forceinline__ float gate_f(float dy, float y) { return dy * elu_dout(y); }

// number of 16-row output tiles and of channels a lane ends up with, for COUT output channels
template <int COUT> struct OutT {
    static constexpr int NCT = COUT >= 16 ? COUT / 16 : 1;
    static constexpr int NCH = 4 * NCT;                          // channels per lane (lanes with rows beyond COUT hold nothing)
};
// output channel of row m of tile ct: a lane's rows 4g..4g+3 of all tiles are NCH consecutive channels
template <int COUT> __device__ __forceinline__ int och(int ct, int m) {
    return OutT<COUT>::NCH * (m >> 2) + 4 * ct + (m & 3);
}

// ---- k_s4: C -> 2C, rows 2ho + kh ---------------------------------------------------------------------------------------
// K = 4 C.  Steps of 32 (16 at C = 4): lane group g of step j covers
//   C = 32: kh = j, channels 8g..      C = 16: kh = 2j + (g >> 1), channels 8 (g & 1)..      C = 8: kh = g, channels 0..7
//   C = 4 (K = 16, v_mfma_f32_16x16x16_bf16): kh = g, channels 0..3
template <int C> struct S4 {
    static constexpr int NS = C == 32 ? 4 : (C == 16 ? 2 : 1);
    static constexpr int CE = C == 4 ? 4 : 8;                    // channels per lane per step
    __device__ static int kh(int j, int g) { return C == 32 ? j : (C == 16 ? 2 * j + (g >> 1) : g); }
    __device__ static int c0(int g) { return C == 32 ? 8 * g : (C == 16 ? 8 * (g & 1) : 0); }
};

template <int CE> struct VecE { typedef typename std::conditional<CE == 8, e16x8, e16x4>::type type; };

template <int CE>
__device__ __forceinline__ typename VecE<CE>::type load_gated(const e16* in, const e16* gy, long off, bool ok, bool gate) {
    typedef typename VecE<CE>::type vec_t;
    vec_t v;
#pragma unroll
    for (int j = 0; j < CE; ++j) v[j] = (e16)0.f;
    if (ok) {
        v = *reinterpret_cast<const vec_t*>(in + off);
        if (gate) {
            const vec_t yv = *reinterpret_cast<const vec_t*>(gy + off);
#pragma unroll
            for (int j = 0; j < CE; ++j) v[j] = (e16)gate_f((float)v[j], (float)yv[j]);
        }
    }
    return v;
}

template <int CE> __device__ __forceinline__ f32x4 mma_e(typename VecE<CE>::type a, typename VecE<CE>::type b, f32x4 c) {
    if constexpr (CE == 8) return mma32(a, b, c);
    else return mma16(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c);
}

template <int COUT, int NCH>
__device__ __forceinline__ void store_lane(e16* out, long pix, int g, const float (&v)[NCH], bool ok) {
    if (!ok || NCH * g >= COUT) return;
    e16* d = out + pix * COUT + NCH * g;
    if constexpr (NCH == 4) {
        e16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (e16)v[j];
        *reinterpret_cast<e16x4*>(d) = o;
    } else {
#pragma unroll
        for (int q = 0; q < NCH / 8; ++q) {
            e16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (e16)v[8 * q + j];
            *reinterpret_cast<e16x8*>(d + 8 * q) = o;
        }
    }
}

// The next group's rows are requested before this group's products (PF) -- except at C = 32, where the 64 VGPRs of weights plus
// two sets of six 16-byte pieces (and their gates) put the kernel at 240-276 registers = one or two waves per SIMD; without the second
// set it is 160-208 and the waves cover each other instead (round 3, library A/B: tconv C = 32 forward 0.145 -> 0.120 ms, backward
// 0.520 -> 0.487 ms, sconv backward 0.456 -> 0.426 ms).
template <int C, bool GATE, bool ACT>
__global__ __launch_bounds__(NT) void k_s4(const e16* __restrict__ in, const e16* __restrict__ gy,
                                            const float* __restrict__ w, const float* __restrict__ bias,
                                            e16* __restrict__ out, int B, int Hin, int Hout, int T, int tb, long ngroups) {
    using S = S4<C>;
    constexpr int COUT = 2 * C, NCT = OutT<COUT>::NCT, NCH = OutT<COUT>::NCH;
    typedef typename VecE<S::CE>::type vec_t;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the group index math below stays on the SALU
    const int n = lane & 15, g = lane >> 4;
    vec_t A[S::NS][NCT];
#pragma unroll
    for (int j = 0; j < S::NS; ++j)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int co = och<COUT>(ct, n), kh = S::kh(j, g), c0 = S::c0(g);
#pragma unroll
            for (int e = 0; e < S::CE; ++e) A[j][ct][e] = (e16)(co < COUT ? w[(co * C + c0 + e) * 4 + kh] : 0.f);
        }
    float br[NCH];
#pragma unroll
    for (int e = 0; e < NCH; ++e) br[e] = (ACT && NCH * g + e < COUT) ? bias[NCH * g + e] : 0.f;

    // C >= 16: one group = 16 frames of the output-row PAIR 2m, 2m + 1, which share two of their four input rows: six rows are
    // fetched (and, in the data-gradient use, gated) for two outputs instead of eight.  Pieces: one row each at C = 32, the two rows
    // of a tap pair at C = 16; output o uses pieces o * OFS + j.  C <= 8 (all four rows in one piece): one output row per group.
    constexpr bool PAIR = C >= 16;
    constexpr int NO = PAIR ? 2 : 1, OFS = PAIR ? S::NS / 2 : 0, NPC = S::NS + OFS;
    const int Hg = PAIR ? (Hout + 1) >> 1 : Hout;
    auto fetch = [&](int grp, vec_t (&q)[NPC]) {
        const int tblk = grp % tb;
        const int bh = grp / tb;
        const int m = bh % Hg, b = bh / Hg;
        const int t = tblk * 16 + n;
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            // piece j of output 0 is step j; beyond NS it is step (j - OFS) of output 1
            const int hi = 2 * (NO * m) + (j < S::NS ? S::kh(j, g) : 2 + S::kh(j - OFS, g));
            q[j] = load_gated<S::CE>(in, gy, (((long)b * Hin + hi) * T + t) * C + S::c0(g), grp < (int)ngroups && t < T && hi < Hin, GATE);
        }
    };
    const int gstride = gridDim.x * 4;                            // ngroups < 2^31 (checked by the launcher): 32-bit scalar arithmetic
    int grp = blockIdx.x * 4 + wave;
    vec_t bq[NPC], bn[NPC];
    constexpr bool PF = C != 32;      // C = 32: no double buffer -- 64 VGPRs of weights leave no room for it (see the note at k_s4)
    if (PF) fetch(grp, bq);
    for (; grp < (int)ngroups; grp += gstride) {
        if (PF) fetch(grp + gstride, bn); else fetch(grp, bq);   // next group's rows are in flight during this one's products
        const int tblk = grp % tb;
        const int bh = grp / tb;
        const int m = bh % Hg, b = bh / Hg;
        const int t = tblk * 16 + n;
        const bool ok = t < T;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const int ho = NO * m + o;
            f32x4 acc[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < S::NS; ++j)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = mma_e<S::CE>(A[j][ct], bq[o * OFS + j], acc[ct]);
            float v[NCH];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = acc[ct][r] + br[4 * ct + r];
                    v[4 * ct + r] = ACT ? elu_f(a) : a;
                }
            store_lane<COUT, NCH>(out, ((long)b * Hout + ho) * T + t, g, v, ok && ho < Hout);
        }
        if (PF) {
#pragma unroll
        for (int j = 0; j < NPC; ++j) bq[j] = bn[j];
        }
    }
}

// ---- k_p2: 2C -> C, rows (h - kh) / 2 for kh = h (mod 2) ------------------------------------------------------------------
// K = 4 C over (row select rs: kh = parity + 2 rs, input row (h - parity)/2 - rs; 2C channels).  Lane group g of step j:
//   C = 32: rs = j >> 1, channels 32 (j & 1) + 8g..     C = 16: rs = j, channels 8g..     C = 8: rs = g >> 1, channels 8 (g & 1)..
//   C = 4 (K = 16): rs = g >> 1, channels 4 (g & 1)..
template <int C> struct P2 {
    static constexpr int NS = C == 32 ? 4 : (C == 16 ? 2 : 1);
    static constexpr int CE = C == 4 ? 4 : 8;
    __device__ static int rs(int j, int g) { return C == 32 ? (j >> 1) : (C == 16 ? j : (g >> 1)); }
    __device__ static int c0(int j, int g) { return C == 32 ? 32 * (j & 1) + 8 * g : (C == 16 ? 8 * g : (C == 8 ? 8 * (g & 1) : 4 * (g & 1))); }
};

template <int C, bool GATE, bool ACT>
__global__ __launch_bounds__(NT) void k_p2(const e16* __restrict__ in, const e16* __restrict__ gy,
                                            const float* __restrict__ w, const float* __restrict__ bias,
                                            e16* __restrict__ out, int B, int Hin, int Hout, int T, int tb, long ngroups) {
    using S = P2<C>;
    constexpr int CIN = 2 * C, NCT = OutT<C>::NCT, NCH = OutT<C>::NCH;
    typedef typename VecE<S::CE>::type vec_t;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the group index math below stays on the SALU
    const int n = lane & 15, g = lane >> 4;
    vec_t A[2][S::NS][NCT];                                      // [output-row parity]
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int j = 0; j < S::NS; ++j)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int co = och<C>(ct, n), kh = par + 2 * S::rs(j, g), c0 = S::c0(j, g);
#pragma unroll
                for (int e = 0; e < S::CE; ++e) A[par][j][ct][e] = (e16)(co < C ? w[((c0 + e) * C + co) * 4 + kh] : 0.f);
            }
    float br[NCH];
#pragma unroll
    for (int e = 0; e < NCH; ++e) br[e] = (ACT && NCH * g + e < C) ? bias[NCH * g + e] : 0.f;

    // one group = 16 frames of the output row PAIR 2m, 2m+1: both parities read the same two input rows m and m - 1
    const int Hp = (Hout + 1) >> 1;
    auto fetch = [&](int grp, vec_t (&q)[S::NS]) {
        const int tblk = grp % tb;
        const int bm = grp / tb;
        const int m = bm % Hp, b = bm / Hp;
        const int t = tblk * 16 + n;
#pragma unroll
        for (int j = 0; j < S::NS; ++j) {
            const int hi = m - S::rs(j, g);
            q[j] = load_gated<S::CE>(in, gy, (((long)b * Hin + hi) * T + t) * CIN + S::c0(j, g),
                                     grp < (int)ngroups && t < T && hi >= 0 && hi < Hin, GATE);
        }
    };
    const int gstride = gridDim.x * 4;                            // ngroups < 2^31 (checked by the launcher): 32-bit scalar arithmetic
    int grp = blockIdx.x * 4 + wave;
    vec_t bq[S::NS], bn[S::NS];
    constexpr bool PF = C != 32;      // C = 32: no double buffer -- 64 VGPRs of weights leave no room for it (see the note at k_s4)
    if (PF) fetch(grp, bq);
    for (; grp < (int)ngroups; grp += gstride) {
        if (PF) fetch(grp + gstride, bn); else fetch(grp, bq);   // next group's rows are in flight during this one's products
        const int tblk = grp % tb;
        const int bm = grp / tb;
        const int m = bm % Hp, b = bm / Hp;
        const int t = tblk * 16 + n;
        const bool ok = t < T;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int h = 2 * m + par;
            f32x4 acc[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < S::NS; ++j)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = mma_e<S::CE>(A[par][j][ct], bq[j], acc[ct]);
            float v[NCH];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = acc[ct][r] + br[4 * ct + r];
                    v[4 * ct + r] = ACT ? elu_f(a) : a;
                }
            store_lane<C, NCH>(out, ((long)b * Hout + h) * T + t, g, v, ok && h < Hout);
        }
        if (PF) {
#pragma unroll
        for (int j = 0; j < S::NS; ++j) bq[j] = bn[j];
        }
    }
}

// ---- C = 4, 8: a lane is a pixel (v_mfma_f32_4x4x4_16b_bf16, see conv_wide_bf16.hip) -------------------------------------
// With 4 / 8 channels a 16-row tile is mostly padding and three quarters of the lanes would idle in the epilogue; here every
// lane loads its own pixel's rows (8 / 16 bytes), owns all output channels of its pixel and stores 16 / 8 bytes.
__device__ __forceinline__ s16x4 lo4(e16x8 v) { return __builtin_bit_cast(s16x4, __builtin_shufflevector(v, v, 0, 1, 2, 3)); }
__device__ __forceinline__ s16x4 hi4(e16x8 v) { return __builtin_bit_cast(s16x4, __builtin_shufflevector(v, v, 4, 5, 6, 7)); }

// all N channels of one pixel as N / 4 four-channel operands of the 4x4x4 product, optionally gated by the saved output
template <int N, bool GATE>
__device__ __forceinline__ void load_px(const e16* in, const e16* gy, long off, bool ok, s16x4 (&c)[N / 4]) {
    if constexpr (N == 4) {
        c[0] = __builtin_bit_cast(s16x4, load_gated<4>(in, gy, off, ok, GATE));
    } else {
#pragma unroll
        for (int q = 0; q < N / 8; ++q) {
            const e16x8 v = load_gated<8>(in, gy, off + 8 * q, ok, GATE);
            c[2 * q] = lo4(v); c[2 * q + 1] = hi4(v);
        }
    }
}
template <int N, bool ACT>
__device__ __forceinline__ void store_px(e16* out, long off, const f32x4 (&acc)[N / 4], const float (&br)[N], bool ok) {
    if (!ok) return;
    if constexpr (N == 4) {
        e16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float a = acc[0][e] + br[e]; o[e] = (e16)(ACT ? elu_f(a) : a); }
        *reinterpret_cast<e16x4*>(out + off) = o;
    } else {
#pragma unroll
        for (int q = 0; q < N / 8; ++q) {
            e16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float a = acc[2 * q + (e >> 2)][e & 3] + br[8 * q + e]; o[e] = (e16)(ACT ? elu_f(a) : a); }
            *reinterpret_cast<e16x8*>(out + off + 8 * q) = o;
        }
    }
}

// "gather four rows", C -> 2C (C = 4, 8); one group = 64 frames of the output-row pair 2m, 2m + 1: six input rows for two outputs
template <int C, bool GATE, bool ACT>
__global__ __launch_bounds__(NT) void k_s4n(const e16* __restrict__ in, const e16* __restrict__ gy,
                                             const float* __restrict__ w, const float* __restrict__ bias,
                                             e16* __restrict__ out, int B, int Hin, int Hout, int T, int tb, long ngroups) {
    constexpr int NBI = C / 4, NBO = C / 2, CO = 2 * C;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i4 = lane & 3;
    s16x4 A[4][NBO][NBI];                                        // [kh][output block][input block]
#pragma unroll
    for (int kh = 0; kh < 4; ++kh)
#pragma unroll
        for (int ob = 0; ob < NBO; ++ob)
#pragma unroll
            for (int kb = 0; kb < NBI; ++kb) {
                e16x4 t;
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = (e16)w[((4 * ob + i4) * C + 4 * kb + k) * 4 + kh];
                A[kh][ob][kb] = __builtin_bit_cast(s16x4, t);
            }
    float br[CO];
#pragma unroll
    for (int e = 0; e < CO; ++e) br[e] = ACT ? bias[e] : 0.f;
    const int Hg = (Hout + 1) >> 1;
    auto fetch = [&](int grp, s16x4 (&q)[6][NBI]) {
        const int tblk = grp % tb;
        const int bh = grp / tb;
        const int m = bh % Hg, b = bh / Hg;
        const int t = tblk * 64 + lane;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int hi = 4 * m + r;
            load_px<C, GATE>(in, gy, (((long)b * Hin + hi) * T + t) * C, grp < (int)ngroups && t < T && hi < Hin, q[r]);
        }
    };
    // Neighbouring row pairs share two of their six input rows.  Workgroups go to the XCDs round-robin, each XCD has its own L2, and in
    // plain group order the sharers sit four workgroups apart -- on different XCDs: every shared row came from HBM twice (PMC: 1.25x the
    // algorithmic bytes here, 1.50x in k_p2n).  xcd_order hands each XCD one contiguous eighth of the group raster.
    const int nvb = ((int)ngroups + 3) >> 2;                      // ngroups < 2^30 (checked by the launcher): 32-bit scalar arithmetic
    int v = blockIdx.x;
    int grp = xcd_order(v, nvb) * 4 + wave;
    s16x4 bq[6][NBI], bn[6][NBI];
    fetch(grp, bq);
    for (; v < nvb; v += gridDim.x) {
        grp = xcd_order(v, nvb) * 4 + wave;
        fetch(v + (int)gridDim.x < nvb ? xcd_order(v + gridDim.x, nvb) * 4 + wave : (int)ngroups, bn);
        const int tblk = grp % tb;
        const int bh = grp / tb;
        const int m = bh % Hg, b = bh / Hg;
        const int t = tblk * 64 + lane;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int ho = 2 * m + o;
            f32x4 acc[NBO];
#pragma unroll
            for (int ob = 0; ob < NBO; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                for (int ob = 0; ob < NBO; ++ob)
#pragma unroll
                    for (int kb = 0; kb < NBI; ++kb) acc[ob] = mma4(A[kh][ob][kb], bq[2 * o + kh][kb], acc[ob]);
            store_px<CO, ACT>(out, (((long)b * Hout + ho) * T + t) * CO, acc, br, grp < (int)ngroups && t < T && ho < Hout);
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int kb = 0; kb < NBI; ++kb) bq[r][kb] = bn[r][kb];
    }
}

// "two rows by parity", 2C -> C (C = 4, 8); one group = 64 frames of the output-row pair 2m, 2m + 1 from input rows m, m - 1
template <int C, bool GATE, bool ACT>
__global__ __launch_bounds__(NT) void k_p2n(const e16* __restrict__ in, const e16* __restrict__ gy,
                                             const float* __restrict__ w, const float* __restrict__ bias,
                                             e16* __restrict__ out, int B, int Hin, int Hout, int T, int tb, long ngroups) {
    constexpr int CI = 2 * C, NBI = CI / 4, NBO = C / 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i4 = lane & 3;
    s16x4 A[2][2][NBO][NBI];                                     // [parity][row select][output block][input block]
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int rs = 0; rs < 2; ++rs)
#pragma unroll
            for (int ob = 0; ob < NBO; ++ob)
#pragma unroll
                for (int kb = 0; kb < NBI; ++kb) {
                    e16x4 t;
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] = (e16)w[((4 * kb + k) * C + 4 * ob + i4) * 4 + par + 2 * rs];
                    A[par][rs][ob][kb] = __builtin_bit_cast(s16x4, t);
                }
    float br[C];
#pragma unroll
    for (int e = 0; e < C; ++e) br[e] = ACT ? bias[e] : 0.f;
    const int Hp = (Hout + 1) >> 1;
    auto fetch = [&](int grp, s16x4 (&q)[2][NBI]) {
        const int tblk = grp % tb;
        const int bm = grp / tb;
        const int m = bm % Hp, b = bm / Hp;
        const int t = tblk * 64 + lane;
#pragma unroll
        for (int rs = 0; rs < 2; ++rs) {
            const int hi = m - rs;
            load_px<CI, GATE>(in, gy, (((long)b * Hin + hi) * T + t) * CI, grp < (int)ngroups && t < T && hi >= 0 && hi < Hin, q[rs]);
        }
    };
    const int nvb = ((int)ngroups + 3) >> 2;                      // XCD-contiguous group order: see k_s4n (each input row serves two groups)
    int v = blockIdx.x;
    int grp = xcd_order(v, nvb) * 4 + wave;
    s16x4 bq[2][NBI], bn[2][NBI];
    fetch(grp, bq);
    for (; v < nvb; v += gridDim.x) {
        grp = xcd_order(v, nvb) * 4 + wave;
        fetch(v + (int)gridDim.x < nvb ? xcd_order(v + gridDim.x, nvb) * 4 + wave : (int)ngroups, bn);
        const int tblk = grp % tb;
        const int bm = grp / tb;
        const int m = bm % Hp, b = bm / Hp;
        const int t = tblk * 64 + lane;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            f32x4 acc[NBO];
#pragma unroll
            for (int ob = 0; ob < NBO; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rs = 0; rs < 2; ++rs)
#pragma unroll
                for (int ob = 0; ob < NBO; ++ob)
#pragma unroll
                    for (int kb = 0; kb < NBI; ++kb) acc[ob] = mma4(A[par][rs][ob][kb], bq[rs][kb], acc[ob]);
            const int h = 2 * m + par;
            store_px<C, ACT>(out, (((long)b * Hout + h) * T + t) * C, acc, br, grp < (int)ngroups && t < T && h < Hout);
        }
#pragma unroll
        for (int rs = 0; rs < 2; ++rs)
#pragma unroll
            for (int kb = 0; kb < NBI; ++kb) bq[rs][kb] = bn[rs][kb];
    }
}

// ---- k_w4: weight and bias gradient ----------------------------------------------------------------------------------------
//   dW[a][b][kh] += sum_{r,t} small[r,t,a] big[2r+kh,t,b]       a < 2C, b < C;   db += sum of the gated operand
// GS = true: small is gated (sconv: small = dy with y, big = x); false: big is gated (tconv: small = x, big = dy with y).
// One K slot = one pixel; an operand tile = the pixel's 16 channels [16 tile .. 16 tile + 15] (the lanes of channels that do
// not exist re-read existing ones; their rows / columns of D are never used).  Pixels of 64 / 128 bytes have their 32-byte
// blocks XOR-swizzled with pixel bits so that the eight pixels of a half-wave read fall on different banks.
template <int PB> __device__ __forceinline__ int blk_swz(int p) { return PB == 128 ? ((p >> 1) & 3) : (PB == 64 ? ((p >> 2) & 1) : 0); }

// Tile height TR (rows of the small side): a tile's bytes scale with C, and at C = 4 four rows are 10 KB per barrier pair -- the launch
// is then all latency (round 3: sconv / tconv backward at C = 4 0.478 / 0.531 ms with TR = 4, 0.420 / 0.434 ms with TR = 16; C = 8:
// 0.359 / 0.382 -> 0.362 / 0.357 ms with TR = 8; TR = 16 at C = 8 gives nothing more).
template <int C> struct W4 {
    static constexpr int TR = C == 4 ? 16 : (C == 8 ? 8 : 4), TW = 64;
    static constexpr int SB = 4 * C, BB = 2 * C;                 // bytes per pixel: small (2C channels), big (C channels)
    static constexpr int BROWS = 2 * TR + 3;                     // 2 TR + 2 rows feed the products; one more so that the last tile
                                                                 // reaches the output_padding row of a transposed layer (bias gradient)
    // images rounded up to whole rounds of DMA instructions (256 pieces of 16 bytes): the tail lanes write zeros there
    static constexpr int S_BYTES = (TR * TW * SB + 4095) / 4096 * 4096, B_BYTES = (BROWS * TW * BB + 4095) / 4096 * 4096;
    static constexpr int SX_BYTES = ((TR + 1) * TW * SB + 4095) / 4096 * 4096;      // with the halo row of the merged sconv backward
    static constexpr int NA = 2 * C >= 16 ? 2 * C / 16 : 1, NBT = C >= 16 ? C / 16 : 1;
    static constexpr int DUMP = 4 * NA * NBT * 256;
    static constexpr int LDS_BYTES = S_BYTES + B_BYTES;
};

// byte offset inside a pixel of the 8-byte run this lane supplies for operand tile `tile`
template <int PB> __device__ __forceinline__ int tr_off(int p, int tile, int trq) {
    if (PB >= 32) return ((tile ^ blk_swz<PB>(p)) << 5) + 8 * trq;
    return PB == 16 ? 8 * (trq & 1) : 0;
}

template <int PB, int ROWS, bool GATED, int BATCH = 64>
__device__ __forceinline__ void stage_tile(unsigned char* lds, const e16* src, const e16* ysrc, int row_h0, int Hs,
                                           int t0, int T, int tid, float (&dbacc)[8], int db_rows, int db_row0 = 0) {
    // image [ROWS][TW][PB bytes]; piece = 16 bytes; GATED: through registers with dy * ELU'(y), else LDS-DMA
    constexpr int TWp = 64, PPP = PB >= 16 ? 1 : 16 / PB, CGn = PB >= 16 ? PB / 16 : 1;
    constexpr int NPC = ROWS * TWp * PB / 16, NIT = (NPC + NT - 1) / NT;
    const e16* zero = reinterpret_cast<const e16*>(&g_wzero16);
    const int lane = tid & 63, wave = tid >> 6;
    if constexpr (!GATED) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * NT + wave * 64, p = i + lane;
            const int q = (p / CGn) * PPP, cgp = p % CGn;            // first pixel of the piece, physical 16-byte position in it
            const int row = q / TWp, px = q - row * TWp;
            // physical position -> logical channel group (32-byte blocks swizzled)
            const int cg = PB >= 32 ? ((((cgp >> 1) ^ blk_swz<PB>(q)) << 1) | (cgp & 1)) : cgp;
            const int h = row_h0 + row, t = t0 + px;
            const bool ok = p < NPC && (unsigned)h < (unsigned)Hs && t < T;
            const long off = ((long)h * T + t) * (PB / 2) + cg * 8;
            glds16(ok ? src + off : zero, lds + (long)i * 16);
        }
    } else {
        // the requests of a batch go out first (fully unrolled), the gating and the LDS writes follow; one batch = the whole tile
        // except where the registers are needed elsewhere (C = 32: 11 iterations x 8 VGPRs)
        constexpr int NB_ = BATCH < NIT ? BATCH : NIT;
#pragma unroll
        for (int it0 = 0; it0 < NIT; it0 += NB_) {
            e16x8 v[NB_], yv[NB_];
#pragma unroll
            for (int u = 0; u < NB_; ++u) {
                const int it = it0 + u;
                const int p = it * NT + tid;
                const int q = (p / CGn) * PPP, cgp = p % CGn;
                const int row = q / TWp, px = q - row * TWp;
                const int cg = PB >= 32 ? ((((cgp >> 1) ^ blk_swz<PB>(q)) << 1) | (cgp & 1)) : cgp;
                const int h = row_h0 + row, t = t0 + px;
                const bool ok = it < NIT && p < NPC && (unsigned)h < (unsigned)Hs && t < T;
                // clamped unconditional loads (no branch in front of the later requests); masked when used
                const long o2 = ok ? ((long)h * T + t) * (PB / 2) + cg * 8 : 0;
                v[u] = *reinterpret_cast<const e16x8*>(src + o2);
                yv[u] = *reinterpret_cast<const e16x8*>(ysrc + o2);
            }
#pragma unroll
            for (int u = 0; u < NB_; ++u) {
                const int it = it0 + u;
                const int p = it * NT + tid;
                const int q = (p / CGn) * PPP;
                const int row = q / TWp, px = q - row * TWp;
                const bool ok = it < NIT && p < NPC && (unsigned)(row_h0 + row) < (unsigned)Hs && t0 + px < T;
                e16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gq = ok ? gate_f((float)v[u][j], (float)yv[u][j]) : 0.f;
                    if (row >= db_row0 && row < db_rows) dbacc[j] += gq;    // rows shared with a neighbouring tile are counted once
                    o[j] = (e16)gq;
                }
                if (it < NIT && p < NPC) *reinterpret_cast<e16x8*>(lds + (long)p * 16) = o;
            }
        }
    }
}

// DX = true: the data gradient of the layer is computed in the same pass from the gated tile already in LDS -- the separate
// k_s4 / k_p2 launch read dy and y a second time.  GS = false (tconv): dx_small[r] = sum_kh W^T g_big[2r + kh] (the k_s4 product, B operand
// from the big image); GS = true (sconv): dx_big[2m + par] = sum_rs W g_small[m - rs] (the k_p2 product), which needs one small row
// ABOVE the tile: the small image is staged with a halo row (it takes no part in the weight / bias gradient).
// C = 32 (SPLIT): the 4 x 4 x 2 accumulator tiles (128 VGPRs) are split over the waves by small-side tile a = wave, every wave then
// walks all rows and both column halves; with the gated operand staged four iterations at a time the kernel fits three waves per
// SIMD with the 64 VGPRs of data-gradient weights aboard (468 VGPRs = one wave per SIMD before).
template <int PB> __device__ __forceinline__ const unsigned char* px_piece(const unsigned char* img, int q, int ch0) {
    // address of the channels ch0.. (8 of them, 4 at 8-byte pixels) of pixel q of a stage_tile image
    const int cg = ch0 >> 3;
    if (PB >= 64) return img + (long)q * PB + ((((cg >> 1) ^ blk_swz<PB>(q)) << 1) | (cg & 1)) * 16;
    return img + (long)q * PB + ch0 * 2;
}

// PRE (round 5): the gated operand arrives ALREADY gated -- the backward of the residual level behind this layer left dx * ELU'(x), and its
// x is this layer's output (tt_wide_level_bwd_gated) -- so the saved output is not read at all (one tensor pass less per layer backward),
// both operands come in by LDS-DMA (nothing is staged through registers: no dependent load -> gate -> write batches), and the bias
// gradient, which the register staging summed on the side, is one more matrix product per visit of a gated pixel group: ones (x) operand.
// GDX (transposed layer, DX): dx leaves as dx * ELU'(x) -- x, the layer's input, is the ELU output of Decoder.convin in the first
// DecoderBlock, whose backward then takes its gradient gated (tt_latent16_*_pregated); x is the small image already in LDS.
template <int C, bool GS, bool DX, bool PRE = false, bool GDX = false>
__global__ __launch_bounds__(NT, C == 32 ? ((!DX && GS) ? 3 : 2) : W4_WAVES16) void k_w4(const e16* __restrict__ small, const e16* __restrict__ big,
                                            const e16* __restrict__ ygate, float* __restrict__ part, float* __restrict__ dbpart,
                                            const float* __restrict__ w, e16* __restrict__ dx,
                                            int B, int Hs, int Hb, int T, int tiles_h, int tiles_t, int ntiles) {
    using G = W4<C>;
    constexpr int SROW0 = (DX && GS) ? 1 : 0;                    // image row of the tile's first small row
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* ss = smem;
    unsigned char* bs = smem + ((DX && GS) ? G::SX_BYTES : G::S_BYTES);
    // PRE: the wave index on the scalar unit -- the bias-gradient products are taken by one wave each, and with a vector `wave` the compiler
    // predicates them through EXEC (which matrix instructions ignore) and merges the accumulators with masked moves: wrong sums, measured
    const int tid = threadIdx.x, lane = tid & 63, wave = PRE ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    const int n = lane & 15, g = lane >> 4, trj = n >> 2, trq = n & 3;
    constexpr bool SPLIT = C == 32;                              // accumulators split over the waves by small-side tile (NA = 4 = waves)
    constexpr int NAW = SPLIT ? 1 : G::NA;
    constexpr int SBATCH = SPLIT ? 4 : 64;                       // iterations of the gated staging in flight at a time
    f32x4 acc[4][NAW][G::NBT];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int a = 0; a < NAW; ++a)
#pragma unroll
            for (int c = 0; c < G::NBT; ++c) acc[k][a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // PRE: bias-gradient accumulators, one per 16-channel tile of the gated operand this wave visits
    // (GS = false: ONE per wave -- C <= 16 has one big-side tile; C = 32: wave w owns tile w & 1 and the rows of parity w >> 1)
    static_assert(!(PRE && !GS) || SPLIT || G::NBT == 1, "one big-side tile per wave");
    constexpr int NDB = (PRE && GS) ? NAW : 1;
    f32x4 dbm[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) dbm[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    e16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (e16)1.f;
    // data-gradient weights in registers (the A operands of k_s4 / k_p2)
    using S = S4<C>;
    using P = P2<C>;
    typedef typename VecE<S::CE>::type dvec_t;
    constexpr int NCTS = OutT<2 * C>::NCT, NCHS = OutT<2 * C>::NCH, NCTP = OutT<C>::NCT, NCHP = OutT<C>::NCH;
    dvec_t AS[DX && !GS ? S::NS : 1][DX && !GS ? NCTS : 1], AP[DX && GS ? 2 : 1][DX && GS ? P::NS : 1][DX && GS ? NCTP : 1];
    if constexpr (DX && !GS) {
#pragma unroll
        for (int j = 0; j < S::NS; ++j)
#pragma unroll
            for (int ct = 0; ct < NCTS; ++ct) {
                const int co = och<2 * C>(ct, n), kh = S::kh(j, g), c0 = S::c0(g);
#pragma unroll
                for (int e = 0; e < S::CE; ++e) AS[j][ct][e] = (e16)(co < 2 * C ? w[(co * C + c0 + e) * 4 + kh] : 0.f);
            }
    }
    if constexpr (DX && GS) {
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int j = 0; j < P::NS; ++j)
#pragma unroll
                for (int ct = 0; ct < NCTP; ++ct) {
                    const int co = och<C>(ct, n), kh = par + 2 * P::rs(j, g), c0 = P::c0(j, g);
#pragma unroll
                    for (int e = 0; e < P::CE; ++e) AP[par][j][ct][e] = (e16)(co < C ? w[((c0 + e) * C + co) * 4 + kh] : 0.f);
                }
    }

    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        int tile = xcd_order(v, ntiles);
        const int tt = tile % tiles_t; tile /= tiles_t;
        const int th = tile % tiles_h;
        const int b = tile / tiles_h, r0 = th * G::TR, t0 = tt * G::TW;
        const e16* sb = small + (long)b * Hs * T * (2 * C);
        const e16* bb = big + (long)b * Hb * T * C;
        __syncthreads();
        if constexpr (PRE) {
            stage_tile<G::BB, G::BROWS, false>(bs, bb, nullptr, 2 * r0, Hb, t0, T, tid, dbacc, 0);
            stage_tile<G::SB, G::TR + SROW0, false>(ss, sb, nullptr, r0 - SROW0, Hs, t0, T, tid, dbacc, 0);
        } else if constexpr (GS) {
            stage_tile<G::BB, G::BROWS, false>(bs, bb, nullptr, 2 * r0, Hb, t0, T, tid, dbacc, 0);
            stage_tile<G::SB, G::TR + SROW0, true, SBATCH>(ss, sb, ygate + (long)b * Hs * T * (2 * C), r0 - SROW0, Hs, t0, T, tid, dbacc,
                                                           G::TR + SROW0, SROW0);
        } else {
            stage_tile<G::SB, G::TR, false>(ss, sb, nullptr, r0, Hs, t0, T, tid, dbacc, 0);
            stage_tile<G::BB, G::BROWS, true, SBATCH>(bs, bb, ygate + (long)b * Hb * T * C, 2 * r0, Hb, t0, T, tid, dbacc,
                                                      th == tiles_h - 1 ? G::BROWS : 2 * G::TR);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        if constexpr (DX && !GS) {
            // dx_small rows r0 .. r0 + TR - 1: gather four gated big rows 2r + kh (image rows 2 (r - r0) + kh)
            for (int grp = wave; grp < G::TR * 4; grp += 4) {
                const int r = grp >> 2, col = (grp & 3) * 16 + n;
                if (r0 + r >= Hs) break;
                const int t = t0 + col;
                f32x4 acc[NCTS];
#pragma unroll
                for (int ct = 0; ct < NCTS; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < S::NS; ++j) {
                    const int q = (2 * r + S::kh(j, g)) * G::TW + col;
                    const dvec_t bq = *reinterpret_cast<const dvec_t*>(px_piece<G::BB>(bs, q, S::c0(g)));
#pragma unroll
                    for (int ct = 0; ct < NCTS; ++ct) acc[ct] = mma_e<S::CE>(AS[j][ct], bq, acc[ct]);
                }
                float vv[NCHS];
#pragma unroll
                for (int ct = 0; ct < NCTS; ++ct)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) vv[4 * ct + rr] = acc[ct][rr];
                if constexpr (GDX) {
                    static_assert(!GDX || NCHS % 8 == 0, "whole 16-byte pieces of the small image per lane");
                    if (NCHS * g < 2 * C) {
#pragma unroll
                        for (int q8 = 0; q8 < NCHS / 8; ++q8) {
                            const e16x8 xv = *reinterpret_cast<const e16x8*>(px_piece<G::SB>(ss, r * G::TW + col, NCHS * g + 8 * q8));
#pragma unroll
                            for (int j = 0; j < 8; ++j) vv[8 * q8 + j] *= elu_dout((float)xv[j]);
                        }
                    }
                }
                store_lane<2 * C, NCHS>(dx, ((long)b * Hs + r0 + r) * T + t, g, vv, t < T);
            }
        }
        if constexpr (DX && GS) {
            // dx_big row pairs (2m, 2m + 1), m = r0 .. r0 + TR - 1 (the last tile also the rows past 2 Hs): gated small rows m, m - 1
            const int Hp = (Hb + 1) >> 1;
            const int mend = (th == tiles_h - 1) ? Hp : r0 + G::TR;
            for (int grp = wave; grp < (mend - r0) * 4; grp += 4) {
                const int m = r0 + (grp >> 2), col = (grp & 3) * 16 + n;
                const int t = t0 + col;
                dvec_t bq[P::NS];
#pragma unroll
                for (int j = 0; j < P::NS; ++j) {
                    const int hi = m - P::rs(j, g);                 // small row; rows outside [0, Hs) contribute nothing
                    const int q = (hi - r0 + SROW0) * G::TW + col;
                    dvec_t z;
#pragma unroll
                    for (int e = 0; e < P::CE; ++e) z[e] = (e16)0.f;
                    bq[j] = (hi >= 0 && hi < Hs) ? *reinterpret_cast<const dvec_t*>(px_piece<G::SB>(ss, q, P::c0(j, g))) : z;
                }
#pragma unroll
                for (int par = 0; par < 2; ++par) {
                    const int h = 2 * m + par;
                    f32x4 acc[NCTP];
#pragma unroll
                    for (int ct = 0; ct < NCTP; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < P::NS; ++j)
#pragma unroll
                        for (int ct = 0; ct < NCTP; ++ct) acc[ct] = mma_e<P::CE>(AP[par][j][ct], bq[j], acc[ct]);
                    float vv[NCHP];
#pragma unroll
                    for (int ct = 0; ct < NCTP; ++ct)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) vv[4 * ct + rr] = acc[ct][rr];
                    store_lane<C, NCHP>(dx, ((long)b * Hb + h) * T + t, g, vv, t < T && h < Hb);
                }
            }
        }

        // weight gradient.  SPLIT: wave = small-side tile, all rows and both column halves; else: (column half, row parity) per wave
        const int rfirst = SPLIT ? 0 : (wave >> 1), rstep = SPLIT ? 1 : 2;
        for (int r = rfirst; r < G::TR; r += rstep) {
            if (r0 + r >= Hs) break;
#pragma unroll
            for (int chh = 0; chh < (SPLIT ? 2 : 1); ++chh) {
                const int colh = SPLIT ? chh : (wave & 1);
                e16x8 sa[NAW];
#pragma unroll
                for (int a = 0; a < NAW; ++a) {
                    s16x4 h2[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int p = (r + SROW0) * G::TW + colh * 32 + 16 * u + 4 * g + trj;
                        h2[u] = lds_tr16(ss + (long)p * G::SB + tr_off<G::SB>(p, SPLIT ? wave : a, trq));
                    }
                    sa[a] = __builtin_bit_cast(e16x8, __builtin_shufflevector(h2[0], h2[1], 0, 1, 2, 3, 4, 5, 6, 7));
                    if constexpr (PRE && GS) dbm[a] = mma32(sa[a], ones, dbm[a]);      // every small pixel is visited once, by one wave per tile a
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int c = 0; c < G::NBT; ++c) {
                        s16x4 h2[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int p = (2 * r + k) * G::TW + colh * 32 + 16 * u + 4 * g + trj;
                            h2[u] = lds_tr16(bs + (long)p * G::BB + tr_off<G::BB>(p, c, trq));
                        }
                        const e16x8 bq = __builtin_bit_cast(e16x8, __builtin_shufflevector(h2[0], h2[1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                        for (int a = 0; a < NAW; ++a) acc[k][a][c] = mma32(sa[a], bq, acc[k][a][c]);
                        if constexpr (PRE && !GS) {
                            // big rows 2 r + k, k = 0, 1 meet every big row of the tile once; SPLIT: all four waves read the same pixels,
                            // wave (k, c) takes the product
                            if (k < 2 && (!SPLIT || (k == (wave >> 1) && c == (wave & 1)))) dbm[0] = mma32(ones, bq, dbm[0]);
                        }
                    }
            }
        }
        if constexpr (PRE && !GS) {
            // the big rows behind the last row pair the loop above visited: the two rows only taps 2, 3 of the last small row reach and the
            // output_padding row -- in the last tile of the column only (elsewhere they are the next tile's first rows)
            if (th == tiles_h - 1) {
                const int nvalid = Hs - r0 < G::TR ? Hs - r0 : G::TR;
                // C = 32: tile c = wave & 1, the (row, column half) pairs alternate between the two waves of a tile; else the four waves
                // share the pairs of the one tile
                const int c = SPLIT ? (wave & 1) : 0;
                const int nx = (G::BROWS - 2 * nvalid) * 2;
                for (int i = SPLIT ? (wave >> 1) : wave; i < nx; i += SPLIT ? 2 : 4) {
                    const int colh = i & 1, row = 2 * nvalid + (i >> 1);
                    s16x4 h2[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int p = row * G::TW + colh * 32 + 16 * u + 4 * g + trj;
                        h2[u] = lds_tr16(bs + (long)p * G::BB + tr_off<G::BB>(p, c, trq));
                    }
                    const e16x8 bq = __builtin_bit_cast(e16x8, __builtin_shufflevector(h2[0], h2[1], 0, 1, 2, 3, 4, 5, 6, 7));
                    dbm[0] = mma32(ones, bq, dbm[0]);
                }
            }
        }
    }
    if constexpr (SPLIT) {
        float* pw = part + ((long)blockIdx.x * 4 + wave) * G::DUMP;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < G::NBT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r)      // only this wave's small-side tile (the reduce reads exactly those slots)
                    pw[(((k * G::NA + wave) * G::NBT + c) * 4 + r) * 64 + lane] = acc[k][0][c][r];
    } else {
        // the four waves hold the same elements (they split columns and rows): summed through LDS in a fixed order -- (wave 2 + wave 0) +
        // (wave 3 + wave 1) -- and ONE dump per workgroup (wave slot 0): k_w4_reduce reads a quarter of the bytes (round 4)
        static_assert(2 * G::DUMP * 4 <= G::LDS_BYTES, "two waves' accumulators fit the images");
        __syncthreads();
        float* wr = reinterpret_cast<float*>(smem);
        auto each = [&](auto&& f) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int a = 0; a < NAW; ++a)
#pragma unroll
                    for (int c = 0; c < G::NBT; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) f((((k * G::NA + a) * G::NBT + c) * 4 + r) * 64 + lane, acc[k][a][c][r]);
        };
        if (wave >= 2) each([&](int i, float v) { wr[(wave - 2) * G::DUMP + i] = v; });
        __syncthreads();
        if (wave < 2) each([&](int i, float v) { wr[wave * G::DUMP + i] += v; });
        __syncthreads();
        float* pw = part + (long)blockIdx.x * 4 * G::DUMP;
        for (int i = tid; i < G::DUMP; i += NT) pw[i] = wr[i] + wr[G::DUMP + i];
    }
    // bias gradient: a thread always stages the same physical 16-byte position of the gated operand's pixels (256 pieces per
    // round is a multiple of 8 pixels, which leaves the swizzle bits unchanged), hence one fixed logical channel group
    constexpr int GB = GS ? G::SB : G::BB, GC = GB / 2;
    constexpr int CGn = GB >= 16 ? GB / 16 : 1, PPP = GB >= 16 ? 1 : 16 / GB;
    __syncthreads();
    float* dl = reinterpret_cast<float*>(smem);
    if constexpr (PRE) {
        // a wave's share of channel ch sits in tile ch / 16: GS = false (A = ones): every row of D is the column sum -- row 0, lanes g == 0;
        // GS = true (B = ones): every column is the row sum -- column 0, lanes n == 0
        constexpr int NTL = GS ? G::NA : G::NBT;
        for (int i = tid; i < 4 * NTL * 16; i += NT) dl[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NDB; ++i) {
            const int tl = SPLIT ? (GS ? wave : (wave & 1)) : i;
            if (GS) { if (n == 0) for (int r = 0; r < 4; ++r) dl[(wave * NTL + tl) * 16 + 4 * g + r] = dbm[i][r]; }
            else { if (g == 0) dl[(wave * NTL + tl) * 16 + n] = dbm[i][0]; }
        }
        __syncthreads();
        if (tid < GC) dbpart[(long)blockIdx.x * 64 + tid] = (dl[tid] + dl[NTL * 16 + tid]) + (dl[2 * NTL * 16 + tid] + dl[3 * NTL * 16 + tid]);
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) dl[tid * 8 + j] = dbacc[j];
    __syncthreads();
    if (tid < GC) {
        float sum = 0.f;
        for (int t2 = 0; t2 < NT; ++t2) {
            const int q = (t2 / CGn) * PPP, cgp = t2 % CGn;
            const int cg = GB >= 32 ? ((((cgp >> 1) ^ blk_swz<GB>(q)) << 1) | (cgp & 1)) : cgp;
            if (GC >= 8) { if (cg == (tid >> 3)) sum += dl[t2 * 8 + (tid & 7)]; }
            else sum += dl[t2 * 8 + tid] + dl[t2 * 8 + 4 + tid];
        }
        dbpart[(long)blockIdx.x * 64 + tid] = sum;
    }
}

struct W4Red { const float* part; const float* dbpart; int gw; float* dw; float* db; float scale = tt_loss_unscale(); };

// 1024 threads = REL consecutive dump elements x RSL slices of the contributing waves (dumps) / workgroups (bias partials)
template <int C, bool GS>
__global__ __launch_bounds__(1024) void k_w4_reduce(W4Red ar) {
    using G = W4<C>;
    constexpr int GB = GS ? G::SB : G::BB;                       // bytes per pixel of the gated operand
    constexpr int GC = GB / 2;                                   // its channel count = length of db
    __shared__ float red[RSL][REL];
    const int el = threadIdx.x % REL, sl = threadIdx.x / REL;
    const int e = blockIdx.x * REL + el;
    float p8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e < G::DUMP) {
        if (C == 32) {                                          // accumulators split by small-side tile: one wave per workgroup holds e
            const int wv = ((e >> 8) / G::NBT) % G::NA;
            for (int j0 = sl; j0 < ar.gw; j0 += 8 * RSL)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (j0 + RSL * u < ar.gw) p8[u] += ar.part[((long)(j0 + RSL * u) * 4 + wv) * G::DUMP + e];
        } else {                                                // one dump per workgroup, in wave slot 0
            for (int j0 = sl; j0 < ar.gw; j0 += 8 * RSL)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (j0 + RSL * u < ar.gw) p8[u] += ar.part[(long)(j0 + RSL * u) * 4 * G::DUMP + e];
        }
    } else if (e < G::DUMP + GC) {
        for (int j = sl; j < ar.gw; j += RSL) p8[0] += ar.dbpart[(long)j * 64 + (e - G::DUMP)];
    }
    red[sl][el] = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
    __syncthreads();
    if (sl != 0) return;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < RSL; ++i) sum += red[i][el];
    if (e < G::DUMP) {
        const int lane = e & 63, r = (e >> 6) & 3, t3 = e >> 8;
        const int c = t3 % G::NBT, a = (t3 / G::NBT) % G::NA, k = t3 / (G::NBT * G::NA);
        const int ach = 16 * a + 4 * (lane >> 4) + r, bch = 16 * c + (lane & 15);
        if (ach < 2 * C && bch < C) ar.dw[(ach * C + bch) * 4 + k] += sum * ar.scale;
    } else if (e < G::DUMP + GC) {
        ar.db[e - G::DUMP] += sum * ar.scale;
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------------------
#ifndef W4_MAX_WG
#define W4_MAX_WG 768
#endif
constexpr int MAX_WG = W4_MAX_WG;             // three workgroups per CU where LDS and registers allow (C <= 16)

template <int C> constexpr long w4_scratch_floats() { return (long)MAX_WG * 4 * W4<C>::DUMP + (long)MAX_WG * 64; }

inline int flat_grid(long ngroups) {
    const long want = (ngroups + 3) / 4;
    const long cap = (long)8 * tt_cus();
    return (int)(want < cap ? want : cap);
}

template <int C, bool GATE, bool ACT>
int launch_s4(const e16* in, const e16* gy, const float* w, const float* bias, e16* out, int B, int Hin, int Hout, int T,
              hipStream_t st) {
    if constexpr (C == 4 || (C == 8 && !GATE)) {              // gated C = 8 (tconv data gradient): the 16-row tile form is faster (0.21 vs 0.26 ms)
        const int tb = (T + 63) / 64;
        const long ngroups = (long)B * ((Hout + 1) / 2) * tb;
        if (ngroups >= (1l << 30)) return TT_E_UNSUPPORTED;       // 32-bit group indices in the kernels
        hipLaunchKernelGGL((k_s4n<C, GATE, ACT>), dim3(flat_grid(ngroups)), dim3(NT), 0, st, in, gy, w, bias, out, B, Hin, Hout, T, tb, ngroups);
        TT_LAUNCH_CHECK();
        return 0;
    }
    const int tb = (T + 15) / 16;
    const long ngroups = (long)B * (C >= 16 ? (Hout + 1) / 2 : Hout) * tb;       // C >= 16: pairs of output rows
    if (ngroups >= (1l << 30)) return TT_E_UNSUPPORTED;
    hipLaunchKernelGGL((k_s4<C, GATE, ACT>), dim3(flat_grid(ngroups)), dim3(NT), 0, st, in, gy, w, bias, out, B, Hin, Hout, T, tb, ngroups);
    TT_LAUNCH_CHECK();
    return 0;
}
template <int C, bool GATE, bool ACT>
int launch_p2(const e16* in, const e16* gy, const float* w, const float* bias, e16* out, int B, int Hin, int Hout, int T,
              hipStream_t st) {
    if constexpr (C <= 8) {
        const int tb = (T + 63) / 64;
        const long ngroups = (long)B * ((Hout + 1) / 2) * tb;
        if (ngroups >= (1l << 30)) return TT_E_UNSUPPORTED;       // 32-bit group indices in the kernels
        hipLaunchKernelGGL((k_p2n<C, GATE, ACT>), dim3(flat_grid(ngroups)), dim3(NT), 0, st, in, gy, w, bias, out, B, Hin, Hout, T, tb, ngroups);
        TT_LAUNCH_CHECK();
        return 0;
    }
    const int tb = (T + 15) / 16;
    const long ngroups = (long)B * ((Hout + 1) / 2) * tb;        // one group = a pair of output rows x 16 frames
    if (ngroups >= (1l << 30)) return TT_E_UNSUPPORTED;
    hipLaunchKernelGGL((k_p2<C, GATE, ACT>), dim3(flat_grid(ngroups)), dim3(NT), 0, st, in, gy, w, bias, out, B, Hin, Hout, T, tb, ngroups);
    TT_LAUNCH_CHECK();
    return 0;
}
// the data gradient rides along in the weight-gradient pass where the registers allow it (TTRAP_W4X=0: always two kernels)
template <int C> inline bool w4x_enabled() {
    static const int on = tt_switch("TTRAP_W4X", 1);
    return on && C <= 32;
}

template <int C, bool GS, bool DX, bool PRE = false, bool GDX = false>
int launch_w4(const e16* small, const e16* big, const e16* ygate, float* dw, float* db, float* ws, const float* w, e16* dx,
              int B, int Hs, int Hb, int T, hipStream_t st) {
    using G = W4<C>;
    constexpr int LDS = ((DX && GS) ? G::SX_BYTES : G::S_BYTES) + G::B_BYTES;
    static AttrOnce once;
    auto kern = k_w4<C, GS, DX, PRE, GDX>;
    if (int rc = raise_lds(kern, LDS, once)) return rc;
    const int tiles_h = (Hs + G::TR - 1) / G::TR, tiles_t = (T + G::TW - 1) / G::TW, ntiles = B * tiles_h * tiles_t;
    int gw = grid_for(ntiles, LDS, 3);
    if (gw > MAX_WG) gw = MAX_WG;
    float* part = ws;
    float* dbpart = ws + (long)MAX_WG * 4 * G::DUMP;
    hipLaunchKernelGGL(kern, dim3(gw), dim3(NT), LDS, st, small, big, ygate, part, dbpart, w, dx, B, Hs, Hb, T, tiles_h, tiles_t, ntiles);
    TT_LAUNCH_CHECK();
    W4Red ra{part, dbpart, gw, dw, db};
    constexpr int total = G::DUMP + (GS ? 2 * C : C);
    hipLaunchKernelGGL((k_w4_reduce<C, GS>), dim3((total + REL - 1) / REL), dim3(1024), 0, st, ra);
    TT_LAUNCH_CHECK();
    return 0;
}

inline bool ok_shape(int B, int C, int H, int T) {
    return B > 0 && H >= 4 && T > 0 && (C == 4 || C == 8 || C == 16 || C == 32) && (C != 4 || T % 2 == 0) &&
           (long)H * T * 2 * C < (1l << 31);
}

}  // namespace

#define TT_BY_C(C, CALL)                   \
    switch (C) {                           \
        case 4: { constexpr int CC = 4; return CALL; }   \
        case 8: { constexpr int CC = 8; return CALL; }   \
        case 16: { constexpr int CC = 16; return CALL; } \
        case 32: { constexpr int CC = 32; return CALL; } \
    }                                      \
    return TT_E_UNSUPPORTED

extern "C" {

int64_t tt_stride16_scratch_bytes(int C) {
    switch (C) {
        case 4: return w4_scratch_floats<4>() * 4;
        case 8: return w4_scratch_floats<8>() * 4;
        case 16: return w4_scratch_floats<16>() * 4;
        case 32: return w4_scratch_floats<32>() * 4;
    }
    return -1;
}

int tt_sconv16_fwd(const void* x, const float* w, const float* b, void* y, int B, int C, int H, int T, void* stream) {
    if (!x || !w || !b || !y || !ok_shape(B, C, H, T)) return TT_E_BADARG;
    const int Ho = (H - 4) / 2 + 1;
    hipStream_t st = tt_stream(stream);
    TT_BY_C(C, (launch_s4<CC, false, true>((const e16*)x, nullptr, w, b, (e16*)y, B, H, Ho, T, st)));
}

int tt_sconv16_bwd(const void* x, const void* y, const void* dy, const float* w, void* dx, float* dw, float* db, void* ws, int B,
                   int C, int H, int T, void* stream) {
    if (!x || !y || !dy || !w || !dw || !db || !ws || !ok_shape(B, C, H, T)) return TT_E_BADARG;
    const int Ho = (H - 4) / 2 + 1;
    hipStream_t st = tt_stream(stream);
    if (dx && C <= 16 && w4x_enabled<32>()) {                     // one pass: weight, bias and data gradient
        switch (C) {      // C = 32: the gated-small form spills with the data-gradient weights aboard (0.512 vs 0.400 ms in two kernels) -- not merged
            case 4: return launch_w4<4, true, true>((const e16*)dy, (const e16*)x, (const e16*)y, dw, db, (float*)ws, w, (e16*)dx, B, Ho, H, T, st);
            case 8: return launch_w4<8, true, true>((const e16*)dy, (const e16*)x, (const e16*)y, dw, db, (float*)ws, w, (e16*)dx, B, Ho, H, T, st);
            case 16: return launch_w4<16, true, true>((const e16*)dy, (const e16*)x, (const e16*)y, dw, db, (float*)ws, w, (e16*)dx, B, Ho, H, T, st);
        }
    }
    if (dx) {
        int rc = TT_E_UNSUPPORTED;
        switch (C) {
            case 4: rc = launch_p2<4, true, false>((const e16*)dy, (const e16*)y, w, nullptr, (e16*)dx, B, Ho, H, T, st); break;
            case 8: rc = launch_p2<8, true, false>((const e16*)dy, (const e16*)y, w, nullptr, (e16*)dx, B, Ho, H, T, st); break;
            case 16: rc = launch_p2<16, true, false>((const e16*)dy, (const e16*)y, w, nullptr, (e16*)dx, B, Ho, H, T, st); break;
            case 32: rc = launch_p2<32, true, false>((const e16*)dy, (const e16*)y, w, nullptr, (e16*)dx, B, Ho, H, T, st); break;
        }
        if (rc) return rc;
    }
    TT_BY_C(C, (launch_w4<CC, true, false>((const e16*)dy, (const e16*)x, (const e16*)y, dw, db, (float*)ws, nullptr, nullptr, B, Ho, H, T, st)));
}

int tt_tconv16_fwd(const void* x, const float* w, const float* b, void* y, int B, int C, int H, int T, int out_pad, void* stream) {
    if (!x || !w || !b || !y || B <= 0 || H <= 0 || T <= 0 || out_pad < 0 || out_pad > 1) return TT_E_BADARG;
    const int Ho = 2 * H + 2 + out_pad;
    if (!ok_shape(B, C, Ho, T)) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    TT_BY_C(C, (launch_p2<CC, false, true>((const e16*)x, nullptr, w, b, (e16*)y, B, H, Ho, T, st)));
}

int tt_tconv16_bwd(const void* x, const void* y, const void* dy, const float* w, void* dx, float* dw, float* db, void* ws, int B,
                   int C, int H, int T, int out_pad, void* stream) {
    if (!x || !y || !dy || !w || !dw || !db || !ws || B <= 0 || H <= 0 || T <= 0 || out_pad < 0 || out_pad > 1) return TT_E_BADARG;
    const int Ho = 2 * H + 2 + out_pad;
    if (!ok_shape(B, C, Ho, T)) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    if (dx && w4x_enabled<32>()) {
        switch (C) {
            case 32: return launch_w4<32, false, true>((const e16*)x, (const e16*)dy, (const e16*)y, dw, db, (float*)ws, w, (e16*)dx, B, H, Ho, T, st);
            case 4: return launch_w4<4, false, true>((const e16*)x, (const e16*)dy, (const e16*)y, dw, db, (float*)ws, w, (e16*)dx, B, H, Ho, T, st);
            case 8: return launch_w4<8, false, true>((const e16*)x, (const e16*)dy, (const e16*)y, dw, db, (float*)ws, w, (e16*)dx, B, H, Ho, T, st);
            case 16: return launch_w4<16, false, true>((const e16*)x, (const e16*)dy, (const e16*)y, dw, db, (float*)ws, w, (e16*)dx, B, H, Ho, T, st);
        }
    }
    if (dx) {
        int rc = TT_E_UNSUPPORTED;
        switch (C) {
            case 4: rc = launch_s4<4, true, false>((const e16*)dy, (const e16*)y, w, nullptr, (e16*)dx, B, Ho, H, T, st); break;
            case 8: rc = launch_s4<8, true, false>((const e16*)dy, (const e16*)y, w, nullptr, (e16*)dx, B, Ho, H, T, st); break;
            case 16: rc = launch_s4<16, true, false>((const e16*)dy, (const e16*)y, w, nullptr, (e16*)dx, B, Ho, H, T, st); break;
            case 32: rc = launch_s4<32, true, false>((const e16*)dy, (const e16*)y, w, nullptr, (e16*)dx, B, Ho, H, T, st); break;
        }
        if (rc) return rc;
    }
    TT_BY_C(C, (launch_w4<CC, false, false>((const e16*)x, (const e16*)dy, (const e16*)y, dw, db, (float*)ws, nullptr, nullptr, B, H, Ho, T, st)));
}

// The same two backward passes with dy ALREADY gated (g = dy * ELU'(y), as tt_wide_level_bwd_gated leaves it): y is not read.
int tt_sconv16_bwd_pregated(const void* x, const void* g, const float* w, void* dx, float* dw, float* db, void* ws, int B, int C, int H,
                            int T, void* stream) {
    if (!x || !g || !w || !dw || !db || !ws || !ok_shape(B, C, H, T)) return TT_E_BADARG;
    const int Ho = (H - 4) / 2 + 1;
    hipStream_t st = tt_stream(stream);
    if (dx) { TT_BY_C(C, (launch_w4<CC, true, true, true>((const e16*)g, (const e16*)x, nullptr, dw, db, (float*)ws, w, (e16*)dx, B, Ho, H, T, st))); }
    TT_BY_C(C, (launch_w4<CC, true, false, true>((const e16*)g, (const e16*)x, nullptr, dw, db, (float*)ws, nullptr, nullptr, B, Ho, H, T, st)));
}

int tt_tconv16_bwd_pregated(const void* x, const void* g, const float* w, void* dx, float* dw, float* db, void* ws, int B, int C, int H,
                            int T, int out_pad, int gate_dx, void* stream) {
    if (!x || !g || !w || !dw || !db || !ws || B <= 0 || H <= 0 || T <= 0 || out_pad < 0 || out_pad > 1 || (gate_dx && !dx)) return TT_E_BADARG;
    const int Ho = 2 * H + 2 + out_pad;
    if (!ok_shape(B, C, Ho, T)) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    if (gate_dx) {                                               // dx * ELU'(x): the widths a latent head can sit in front of (tt_latent16_*)
        if (C == 32) return launch_w4<32, false, true, true, true>((const e16*)x, (const e16*)g, nullptr, dw, db, (float*)ws, w, (e16*)dx, B, H, Ho, T, st);
        if (C == 16) return launch_w4<16, false, true, true, true>((const e16*)x, (const e16*)g, nullptr, dw, db, (float*)ws, w, (e16*)dx, B, H, Ho, T, st);
        return TT_E_UNSUPPORTED;
    }
    if (dx) { TT_BY_C(C, (launch_w4<CC, false, true, true>((const e16*)x, (const e16*)g, nullptr, dw, db, (float*)ws, w, (e16*)dx, B, H, Ho, T, st))); }
    TT_BY_C(C, (launch_w4<CC, false, false, true>((const e16*)x, (const e16*)g, nullptr, dw, db, (float*)ws, nullptr, nullptr, B, H, Ho, T, st)));
}

}  // extern "C"
