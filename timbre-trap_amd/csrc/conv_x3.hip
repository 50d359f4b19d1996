// fp32-CLASS residual blocks of the wide levels (C = 16, 32; reference modules.py:721-777) on the 16-bit matrix pipe, for gfx950:
// the inference path that carries the 1e-4 output bar of BASELINE's north_star at more than the fp32 matrix rate.
//
// The fp32 kernels of conv_mfma.hip are bound by v_mfma_f32_16x16x4_f32 (a sixteenth of the 16-bit rate); the 16-bit kernels of
// conv_wide_bf16.hip keep 8 / 11 significant bits per stored activation.  Here every fp32 value v -- activations in HBM and weights in
// LDS -- is held as a PAIR of IEEE halves
//       hi = fp16(v),      lo = fp16((v - hi) * 2^11)                       v = hi + lo * 2^-11  to 2^-23 relative
// (below |v| = 2^-14 hi is a subnormal half -- v_mfma_f32_*_f16 takes subnormal inputs at full value on gfx950,
// tools/diag/mfma_denorm.hip -- and the pair is exact to 2^-36 absolute; |v| > 65504 does not fit: hi = inf and the result is
// non-finite, never silently wrong), and a product is three
// v_mfma_f32_16x16x32_f16 with fp32 accumulation:
//       W x  =  Whi xhi  +  2^-11 (Whi xlo + Wlo xhi)          (the dropped Wlo xlo term is 2^-22 relative)
// with the cross terms in their own accumulators.  Bias, ELU and the residual add are fp32, exactly like conv_mfma.hip.
//
// Layout "x3" of an activation tensor: [B][H][T][2][C] halves -- the hi and lo planes of one pixel are adjacent, 4 C bytes per pixel
// (as many bytes as fp32), so that a tile row with its halo is one contiguous run and goes to LDS by LDS-DMA like in the 16-bit
// kernels, and the K = channels slice of either plane is one 16-byte LDS read in B-operand order: no conversion on the read side.
// The split is done ONCE per element, by the producer's epilogue.
//
// Kernels: k_x3_pack / k_x3_unpack (fp32 planar <-> x3 at the two ends of a level), k_x3_conv<C, D> (one residual block:
// y = ELU(W2 . ELU(W1 (*)_D x + b1) + b2) + x; persistent, XCD-ordered tile walk; weights of both planes in LDS in operand order,
// written by every workgroup for itself).  Forward only: training saves fp32 activations for the fp32 backward.
#define TT_F16 1
#include "wide_common.h"

namespace {

constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;

__device__ __forceinline__ void split(float v, e16& hi, e16& lo) {
    hi = (e16)v;
    lo = (e16)__builtin_fmaf((float)hi, -LO_SCALE, v * LO_SCALE);      // = (v - hi) 2^11 exactly (every step is exact), one fma_mix
}
__device__ __forceinline__ float join(e16 hi, e16 lo) { return __builtin_fmaf((float)lo, LO_INV, (float)hi); }

// The same two steps on PAIRS, with the mixed-precision fma spelled out (round 5).  The kernels below are bound by vector issue, and the
// vectoriser turns the scalar forms above into packed fp32 arithmetic around explicit conversions -- hi back to fp32 (one instruction per
// element), packed fma, packed conversion of lo: three instructions per split element, and two conversions + an fma per joined element.
// v_fma_mix* takes the f16 halves as they are: split = packed conversion of hi (1/2) + packed v * 2^11 (1/2) + one v_fma_mix{lo,hi}_f16 per
// element = 2; join = ONE v_fma_mix_f32 per element.  The arithmetic is the scalar forms' (an fp32 fma, then one rounding to f16).
#ifndef TT_X3_MIX
#define TT_X3_MIX 1
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef e16 e16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// TOMM: the halves go straight into a matrix instruction as an operand.  The compiler's hazard recogniser does not see an asm block as a vector
// instruction, so the two wait states a matrix instruction needs behind a vector write of one of its sources are spelled out too (without
// them the C = 16 and C = 4 kernels, whose products follow the split most closely, read stale registers: tests/test_gpu_x3.py caught it).
template <bool TOMM>
__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi2, unsigned& lo2) {
    const f32x2 v = f32x2{v0, v1};
    hi2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, e16x2));
    const f32x2 sc = v * LO_SCALE;
    const float nk = -LO_SCALE;
    unsigned d;
    if constexpr (TOMM)
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "s_nop 1"
            : "=&v"(d) : "v"(hi2), "v"(nk), "v"(sc.x), "v"(sc.y));
    else
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(d) : "v"(hi2), "v"(nk), "v"(sc.x), "v"(sc.y));
    lo2 = d;
}
__device__ __forceinline__ void join2(unsigned hi2, unsigned lo2, float& o0, float& o1) {
    const float k = LO_INV;
    asm("v_fma_mix_f32 %0, %2, %4, %3 op_sel_hi:[1,0,1]\n\t"
        "v_fma_mix_f32 %1, %2, %4, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]"
        : "=&v"(o0), "=&v"(o1) : "v"(lo2), "v"(hi2), "v"(k));
}
template <bool TOMM = false>
__device__ __forceinline__ void split_v(const float (&v)[4], e16x4& hi, e16x4& lo) {
#if TT_X3_MIX
    u32x2 h, l;
    unsigned a, b;
    split2<TOMM>(v[0], v[1], a, b); h.x = a; l.x = b;
    split2<TOMM>(v[2], v[3], a, b); h.y = a; l.y = b;
    hi = __builtin_bit_cast(e16x4, h); lo = __builtin_bit_cast(e16x4, l);
#else
#pragma unroll
    for (int j = 0; j < 4; ++j) { e16 a_, b_; split(v[j], a_, b_); hi[j] = a_; lo[j] = b_; }
#endif
}
template <bool TOMM = false>
__device__ __forceinline__ void split_v(const float (&v)[8], e16x8& hi, e16x8& lo) {
#if TT_X3_MIX
    u32x4 h, l;
    unsigned a, b;
    split2<TOMM>(v[0], v[1], a, b); h.x = a; l.x = b;
    split2<TOMM>(v[2], v[3], a, b); h.y = a; l.y = b;
    split2<TOMM>(v[4], v[5], a, b); h.z = a; l.z = b;
    split2<TOMM>(v[6], v[7], a, b); h.w = a; l.w = b;
    hi = __builtin_bit_cast(e16x8, h); lo = __builtin_bit_cast(e16x8, l);
#else
#pragma unroll
    for (int j = 0; j < 8; ++j) { e16 a_, b_; split(v[j], a_, b_); hi[j] = a_; lo[j] = b_; }
#endif
}
__device__ __forceinline__ void join_v(e16x4 hi, e16x4 lo, float (&o)[4]) {
#if TT_X3_MIX
    const u32x2 h = __builtin_bit_cast(u32x2, hi), l = __builtin_bit_cast(u32x2, lo);
    join2(h.x, l.x, o[0], o[1]);
    join2(h.y, l.y, o[2], o[3]);
#else
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = join(hi[j], lo[j]);
#endif
}
__device__ __forceinline__ void join_v(e16x8 hi, e16x8 lo, float (&o)[8]) {
#if TT_X3_MIX
    const u32x4 h = __builtin_bit_cast(u32x4, hi), l = __builtin_bit_cast(u32x4, lo);
    join2(h.x, l.x, o[0], o[1]);
    join2(h.y, l.y, o[2], o[3]);
    join2(h.z, l.z, o[4], o[5]);
    join2(h.w, l.w, o[6], o[7]);
#else
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = join(hi[j], lo[j]);
#endif
}

template <int C> struct XL {
    static constexpr int PB = 4 * C;                             // bytes per pixel: [hi C][lo C]
    static constexpr int PP = PB / 16;                           // 16-byte pieces per pixel (C = 32: 8, C = 16: 4)
    static constexpr int CG = C / 8;                             // pieces per plane
    static constexpr int CPR = 256 / PB;                         // pixels per 256 bytes (all 64 banks)
};
// The piece `q` of the pixel in image column `col` sits at position q ^ xswz(col): the sixteen consecutive columns a k-group reads
// as B operand then cover all sixteen bank quads (same rule as cswz of conv_wide_bf16.hip at twice the pixel size).
template <int C> __device__ __forceinline__ int xswz(int col) { return (col / XL<C>::CPR) & (XL<C>::PP - 1); }

// ---- layout change at the ends of a level: one thread per eight channels of one pixel ----------------------------------------
template <int C>
__global__ __launch_bounds__(NT) void k_x3_pack(const float* __restrict__ x, e16* __restrict__ out, int H, int T, long npix) {
    constexpr int CG = C / 8;
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    const long pix = i / CG;
    if (pix >= npix) return;
    const int cg = (int)(i % CG);
    const long plane = (long)H * T;
    const long b = pix / plane, o = pix - b * plane;
    const float* s = x + (b * C + cg * 8) * plane + o;
    e16x8 qh, ql;
#pragma unroll
    for (int j = 0; j < 8; ++j) { e16 h, l; split(s[j * plane], h, l); qh[j] = h; ql[j] = l; }
    *reinterpret_cast<e16x8*>(out + pix * 2 * C + cg * 8) = qh;
    *reinterpret_cast<e16x8*>(out + pix * 2 * C + C + cg * 8) = ql;
}
template <int C>
__global__ __launch_bounds__(NT) void k_x3_unpack(const e16* __restrict__ in, float* __restrict__ y, int H, int T, long npix) {
    constexpr int CG = C / 8;
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    const long pix = i / CG;
    if (pix >= npix) return;
    const int cg = (int)(i % CG);
    const long plane = (long)H * T;
    const long b = pix / plane, o = pix - b * plane;
    const e16x8 qh = *reinterpret_cast<const e16x8*>(in + pix * 2 * C + cg * 8);
    const e16x8 ql = *reinterpret_cast<const e16x8*>(in + pix * 2 * C + C + cg * 8);
    float* d = y + (b * C + cg * 8) * plane + o;
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j * plane] = join(qh[j], ql[j]);
}

// ---- one residual block ---------------------------------------------------------------------------------------------------------
// A wave owns a 16-pixel column group of the tile and walks R rows of it at a time, so that the weight operands of a tap that live
// in LDS are read once per R rows.  Measured first (bench plane, profiles/r04_x3_ablation.txt): staging and products + epilogue of
// one workgroup do not overlap (0.13 + 0.30 ms at C = 32 on 16 x 32 tiles with one workgroup per CU), and the products' side is
// issue-bound (60 matrix + ~250 vector instructions per 16-pixel row) -- so the shapes are chosen for overlap and occupancy:
//   C = 32: TWO workgroups of four waves per CU on 6 x 32 tiles (R = 3: with R = 4 the registers spill): 36-61 KB of tile + 18 KB for the lo planes of W1
//           in LDS; the hi planes of W1 and both planes of W2 sit in registers (88 of the 256 a lane has at two waves per SIMD);
//   C = 16: two workgroups of EIGHT waves per CU on 8 x 64 tiles (64 KB + 11 KB of weights, both planes in LDS): four waves per
//           SIMD at <= 128 registers.
template <int C, int D> struct XT {
    using L = XL<C>;
    static constexpr bool WREG = C == 32;
    static constexpr int TW = C == 32 ? 32 : 64;
    static constexpr int TH = C == 32 ? 6 : 8;
    static constexpr int NTH = C == 32 ? 256 : 512;
    static constexpr int MINW = C == 32 ? 2 : 4;
    static constexpr int NCG = TW / 16, NRG = (NTH / 64) / NCG;  // column groups x row groups of waves
    static constexpr int R = TH / NRG;
    static constexpr int RW = TW + 2 * D, ROWS = TH + 2 * D;
    static constexpr int NP = ROWS * RW * L::PP;                 // 16-byte pieces of the tile with its halo
    static constexpr int NPR = (NP + NTH - 1) / NTH * NTH;       // whole DMA instructions for every wave
    static constexpr int TILE_BYTES = NPR * 16;
    static constexpr int NCT = C / 16;
    static constexpr int NK = C == 32 ? 9 : 5;                   // products per co-tile: one tap (C = 32) / two taps (C = 16)
    static constexpr int NPL = WREG ? 1 : 2;                     // planes of W1 kept in LDS (lo only / hi and lo)
    static constexpr int W1_BYTES = NK * NCT * NPL * 64 * 16;    // [k][ct][plane][lane] x 16 bytes
    static constexpr int W2_BYTES = WREG ? 0 : 2 * 64 * 8;       // C = 16: [plane][lane] x 8 bytes
    static constexpr int LDS_BYTES = TILE_BYTES + W1_BYTES + W2_BYTES;
    static_assert(R * NRG == TH, "rows per wave");
};

// PLANAR: the output goes out as fp32 (B,C,H,T) instead of x3 -- the last block of a level (saves the unpack pass).
template <int C, int D, bool PLANAR>
__global__ __launch_bounds__((XT<C, D>::NTH), (XT<C, D>::MINW)) void k_x3_conv(const e16* __restrict__ x, const float* __restrict__ w1,
                                                                               const float* __restrict__ b1, const float* __restrict__ w2,
                                                                               const float* __restrict__ b2, void* __restrict__ yout,
                                                                               int B, int H, int T, int tiles_h, int tiles_t, int ntiles) {
    using G = XT<C, D>;
    using L = XL<C>;
    constexpr int NTH = G::NTH, NCT = G::NCT, NK = G::NK, R = G::R, PB = L::PB, CG = L::CG, NPL = G::NPL;
    constexpr bool WREG = G::WREG;
    constexpr int NCH = C == 32 ? 8 : 4;                         // channels a lane ends up with
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* w1img = smem + G::TILE_BYTES;
    unsigned char* w2img = w1img + G::W1_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;

    // ---- weights in operand order, split into the two planes: to LDS, and (WREG) the hi planes to registers ----
    for (int e = tid; e < NK * NCT * 64; e += NTH) {
        const int l = e & 63, ct = (e >> 6) % NCT, k = e / (64 * NCT);
        const int ln = l & 15, lg = l >> 4;
        e16x8 qh, ql;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int tap, kc;                                         // tap and contraction channel of this lane's k = 8 g + j
            if (C == 32) { tap = k; kc = 8 * lg + j; }
            else { tap = 2 * k + (lg >> 1); kc = 8 * (lg & 1) + j; }
            const int mo = chan_of<C>(ct, ln);                   // output channel of row n
            const float wv = tap < 9 ? w1[(mo * C + kc) * 9 + tap] : 0.f;
            e16 h, lo_; split(wv, h, lo_); qh[j] = h; ql[j] = lo_;
        }
        if constexpr (!WREG) *reinterpret_cast<e16x8*>(w1img + ((long)((k * NCT + ct) * NPL + 0) * 64 + l) * 16) = qh;
        *reinterpret_cast<e16x8*>(w1img + ((long)((k * NCT + ct) * NPL + NPL - 1) * 64 + l) * 16) = ql;
    }
    e16x8 WH[WREG ? NK : 1][NCT], A2H[NCT], A2L[NCT];           // WREG: hi planes of W1, both planes of W2
    if constexpr (WREG) {
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int j = 0; j < 8; ++j) WH[k][ct][j] = (e16)w1[(chan_of<C>(ct, n) * C + 8 * g + j) * 9 + k];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                e16 h, lo_; split(w2[chan_of<C>(ct, n) * C + 8 * g + j], h, lo_); A2H[ct][j] = h; A2L[ct][j] = lo_;
            }
    } else {
        for (int e = tid; e < 64; e += NTH) {
            e16x4 qh, ql;
#pragma unroll
            for (int j = 0; j < 4; ++j) { e16 h, lo_; split(w2[(e & 15) * C + 4 * (e >> 4) + j], h, lo_); qh[j] = h; ql[j] = lo_; }
            *reinterpret_cast<e16x4*>(w2img + (long)e * 8) = qh;
            *reinterpret_cast<e16x4*>(w2img + (long)(64 + e) * 8) = ql;
        }
    }
    float b1r[NCH], b2r[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) { b1r[j] = b1[NCH * g + j]; b2r[j] = b2[NCH * g + j]; }

    const int c0 = (wave % G::NCG) * 16, r0 = (wave / G::NCG) * R;
    const e16* zero = reinterpret_cast<const e16*>(&g_wzero16);

    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        int tile = xcd_order(v, ntiles);
        const int tt = tile % tiles_t; tile /= tiles_t;
        const int th = tile % tiles_h;
        const int b = tile / tiles_h, h0 = th * G::TH, t0 = tt * G::TW;
        const unsigned char* xb = reinterpret_cast<const unsigned char*>(x) + (long)b * H * T * PB;

        __syncthreads();                                         // the previous tile has been consumed (first pass: weights written)
#ifdef X3_NOSTAGE
        if (v == (int)blockIdx.x)
#endif
        for (int i = wave * 64; i < G::NPR; i += NTH) {
            const int p = i + lane;
            const int row = p / (G::RW * L::PP), rem = p - row * (G::RW * L::PP);
            const int px = rem / L::PP, q = (rem - px * L::PP) ^ xswz<C>(px);       // 16-byte position -> piece held there
            const int h = h0 - D + row, t = t0 - D + px;
            const bool ok = p < G::NP && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
            glds16(ok ? (const void*)(xb + ((long)h * T + t) * PB + q * 16) : (const void*)zero, smem + (long)i * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        const int t = t0 + c0 + n;
        const bool valid = t < T;
#ifdef X3_NOCOMPUTE
        if (v != (int)blockIdx.x) continue;
#endif
        if (h0 + r0 >= H) continue;
        f32x4 am[R][NCT], al[R][NCT];                            // hi x hi (bias as initial value) and the two cross terms
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                am[rr][ct] = f32x4{b1r[4 * ct], b1r[4 * ct + 1], b1r[4 * ct + 2], b1r[4 * ct + 3]};
                al[rr][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            int tap = C == 32 ? k : 2 * k + (g >> 1);
            if (tap > 8) tap = 8;                                // the weights of the missing tenth tap are zero
            const int kh = tap / 3, kw = tap - 3 * kh;
            const int col = c0 + n + kw * D;
            const int gsel = C == 32 ? g : (g & 1), sw = xswz<C>(col);
            e16x8 wh[NCT], wl[NCT], bh[R], bl[R];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if constexpr (WREG) wh[ct] = WH[k][ct];
                else wh[ct] = *reinterpret_cast<const e16x8*>(w1img + ((long)((k * NCT + ct) * NPL + 0) * 64 + lane) * 16);
                wl[ct] = *reinterpret_cast<const e16x8*>(w1img + ((long)((k * NCT + ct) * NPL + NPL - 1) * 64 + lane) * 16);
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const unsigned char* pxp = smem + ((long)(r0 + rr + kh * D) * G::RW + col) * PB;
                bh[rr] = *reinterpret_cast<const e16x8*>(pxp + 16 * (gsel ^ sw));
                bl[rr] = *reinterpret_cast<const e16x8*>(pxp + 16 * ((CG + gsel) ^ sw));
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    am[rr][ct] = mma32(wh[ct], bh[rr], am[rr][ct]);
                    al[rr][ct] = mma32(wh[ct], bl[rr], al[rr][ct]);
                }
#pragma unroll
            for (int rr = 0; rr < R; ++rr)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) al[rr][ct] = mma32(wl[ct], bh[rr], al[rr][ct]);
        }
        // ---- ELU, 1x1 product, ELU, residual add, split, store ----
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int h = h0 + r0 + rr;
            if (h >= H) break;
            const long pix = ((long)b * H + h) * T + t;
            const int colc = c0 + n + D, swc = xswz<C>(colc);
            const unsigned char* pxc = smem + ((long)(r0 + rr + D) * G::RW + colc) * PB;
            float out[NCH];
            if constexpr (C == 32) {
                e16x8 hh, hl;
                float hv[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hv[j] = elu1(__builtin_fmaf(al[rr][0][j], LO_INV, am[rr][0][j]));
                    hv[4 + j] = elu1(__builtin_fmaf(al[rr][1][j], LO_INV, am[rr][1][j]));
                }
                split_v<true>(hv, hh, hl);
                f32x4 zm[2], zl[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    zm[ct] = mma32(A2H[ct], hh, f32x4{b2r[4 * ct], b2r[4 * ct + 1], b2r[4 * ct + 2], b2r[4 * ct + 3]});
                    zl[ct] = mma32(A2H[ct], hl, f32x4{0.f, 0.f, 0.f, 0.f});
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) zl[ct] = mma32(A2L[ct], hh, zl[ct]);
                const e16x8 ch = *reinterpret_cast<const e16x8*>(pxc + 16 * (g ^ swc));
                const e16x8 cl = *reinterpret_cast<const e16x8*>(pxc + 16 * ((CG + g) ^ swc));
                float res[8];
                join_v(ch, cl, res);
#pragma unroll
                for (int j = 0; j < 8; ++j) out[j] = elu1(__builtin_fmaf(zl[j >> 2][j & 3], LO_INV, zm[j >> 2][j & 3])) + res[j];
            } else {
                e16x4 hh, hl;
                float hv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) hv[j] = elu1(__builtin_fmaf(al[rr][0][j], LO_INV, am[rr][0][j]));
                split_v<true>(hv, hh, hl);
                const s16x4 a2h = *reinterpret_cast<const s16x4*>(w2img + (long)lane * 8);
                const s16x4 a2l = *reinterpret_cast<const s16x4*>(w2img + (long)(64 + lane) * 8);
                const f32x4 zm = mma16(a2h, __builtin_bit_cast(s16x4, hh), f32x4{b2r[0], b2r[1], b2r[2], b2r[3]});
                f32x4 zl = mma16(a2h, __builtin_bit_cast(s16x4, hl), f32x4{0.f, 0.f, 0.f, 0.f});
                zl = mma16(a2l, __builtin_bit_cast(s16x4, hh), zl);
                const e16x4 ch = *reinterpret_cast<const e16x4*>(pxc + 16 * ((g >> 1) ^ swc) + 8 * (g & 1));
                const e16x4 cl = *reinterpret_cast<const e16x4*>(pxc + 16 * ((CG + (g >> 1)) ^ swc) + 8 * (g & 1));
                float res[4];
                join_v(ch, cl, res);
#pragma unroll
                for (int j = 0; j < 4; ++j) out[j] = elu1(__builtin_fmaf(zl[j], LO_INV, zm[j])) + res[j];
            }
            if (!valid) continue;
            if constexpr (PLANAR) {
                float* yp = static_cast<float*>(yout) + (((long)b * C + NCH * g) * H + h) * T + t;
#pragma unroll
                for (int j = 0; j < NCH; ++j) yp[(long)j * H * T] = out[j];
            } else {
                e16* y = static_cast<e16*>(yout);
                typename std::conditional<C == 32, e16x8, e16x4>::type oh, ol;
                split_v(out, oh, ol);
                *reinterpret_cast<decltype(oh)*>(y + pix * 2 * C + NCH * g) = oh;
                *reinterpret_cast<decltype(ol)*>(y + pix * 2 * C + C + NCH * g) = ol;
            }
        }
    }
}

template <int C, int D, bool PLANAR>
int launch_x3(const e16* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y, int B, int H, int T,
              hipStream_t st) {
    using G = XT<C, D>;
    const int tiles_h = (H + G::TH - 1) / G::TH, tiles_t = (T + G::TW - 1) / G::TW, ntiles = B * tiles_h * tiles_t;
    static AttrOnce once;
    auto kern = k_x3_conv<C, D, PLANAR>;
    if (int rc = raise_lds(kern, G::LDS_BYTES, once)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for(ntiles, G::LDS_BYTES, 2)), dim3(G::NTH), G::LDS_BYTES, st, x, w1, b1, w2, b2, y, B, H, T,
                       tiles_h, tiles_t, ntiles);
    TT_LAUNCH_CHECK();
    return 0;
}
template <int C, bool PLANAR>
int x3_d(const e16* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y, int B, int H, int T, int d,
         hipStream_t st) {
    switch (d) {
        case 1: return launch_x3<C, 1, PLANAR>(x, w1, b1, w2, b2, y, B, H, T, st);
        case 2: return launch_x3<C, 2, PLANAR>(x, w1, b1, w2, b2, y, B, H, T, st);
        case 3: return launch_x3<C, 3, PLANAR>(x, w1, b1, w2, b2, y, B, H, T, st);
    }
    return TT_E_UNSUPPORTED;
}
// ---- narrow levels (C = 4, 8) with split operands (round 5) ---------------------------------------------------------------------
// The narrow levels of the no-grad fp32 forward ran on the exact-fp32 kernels of conv_small.hip (k_small_lds / k_small_fwd4:
// v_mfma_f32_4x4x1 at a sixteenth of the 16-bit matrix rate, 570-750 us per block at the 96 chunks of BASELINE configs[1] against a
// 270 us memory floor: 15 of its 37 ms).  Same arithmetic as k_x3_conv above -- every value a (hi, lo 2^11) pair of halves, three
// products with the cross terms in their own accumulators, bias / ELU / residual in fp32.
//
// Matrix shape.  A first cut used the lane-per-pixel form of the 16-bit narrow kernels (v_mfma_f32_4x4x4_16b_f16, sixteen independent
// 4 x 4 x 4 products per wave): 120 matrix instructions per 64 pixels at C = 8, and that instruction issues at ~16 cycles -- the
// kernel was bound by the matrix pipe's ISSUE rate at a quarter of its arithmetic (0.56-0.66 ms per block at C = 8, no better than the
// fp32 kernels; profiles/r05_x3n_*).  This cut feeds v_mfma_f32_16x16x16_f16 instead: the 16 rows / 16 k are S = 16 / C PIXELS x C
// channels with the weights as the block-diagonal matrix W (x) I_S (lane n, group g holds A[m = n][k = 4 g ..]: W[co][ci] where the
// pixel slots of m and k agree, 0 elsewhere), the 16 columns are 16 pixels per slot: one instruction = 16 S pixels x C x C of one tap
// and plane pair -- 60 (C = 8) / 30 (C = 4) instructions per 64 pixels instead of 120 / 30 of the slow shape, half / three quarters of
// each one's multiplies spent on the zero blocks, which the matrix pipe has to spare.  Pixel of (column n, slot s) = n + 16 s of the
// span: consecutive lanes read consecutive pixels (conflict-free 8- / 16-byte LDS reads), and the D rows a lane gets (4 g .. 4 g + 3
// = slot s, channels 4 (g & 1) ..) are exactly the k it supplies as B operand: the hidden activation goes from the accumulators of the
// 3x3 product into the 1x1 product without leaving the lane, like in the wide kernels.
//
// Layout "x3n" between the blocks of a level: [B][H][T][2][C] halves like x3 -- C = 4: one 16-byte piece per pixel [hi 4][lo 4];
// C = 8: two pieces [hi 8][lo 8].  The level's first block reads the fp32 planar tensor of the layer in front of it (PIN: C coalesced
// 4-byte loads per pixel, split in registers on the way to LDS) and its last block stores fp32 planar (POUT): no pack / unpack pass,
// the strided and boundary layers around the narrow levels stay as they are.
// LDS tile (16 x 64 pixels + D halo): C = 4 pixel-major; C = 8 [row][plane][column] (the DMA source addresses are permuted).  A wave owns
// rows 4 w .. 4 w + 3 and walks them R at a time.
#ifndef TT_X3N_R8
#define TT_X3N_R8 2              // rows a wave walks at a time at C = 8 (1 | 2 | 4)
#endif
#ifndef TT_X3N_R4
#define TT_X3N_R4 2              // ... at C = 4 (one span per row)
#endif
#ifndef TT_X3N_MINW8
#define TT_X3N_MINW8 3           // workgroups per CU the C = 8 kernel's registers are capped for
#endif
template <int C, int D> struct XN {
    static constexpr int TH = 16, TW = 64, NTH = 256;
    static constexpr int R = C == 8 ? TT_X3N_R8 : TT_X3N_R4;
    static constexpr int S = 16 / C;                             // pixel slots of a matrix column: 2 (C = 8), 4 (C = 4)
    static constexpr int SPAN = 16 * S, NSPAN = TW / SPAN;       // pixels per matrix instruction; instructions per 64-pixel row
    static constexpr int PXB = 4 * C, PPX = PXB / 16;            // bytes / 16-byte pieces per pixel
    static constexpr int RW = TW + 2 * D, ROWS = TH + 2 * D, NPIX = ROWS * RW;
    static constexpr int NP = NPIX * PPX;                        // pieces of the halo'd tile, [row][piece of the pixel][column]
    static constexpr int NPR = (NP + NTH - 1) / NTH * NTH;
    static constexpr int TILE_BYTES = NPR * 16;
    static constexpr int LDS_BYTES = TILE_BYTES;
    static constexpr int NITP = (NPIX + NTH - 1) / NTH;          // PIN: pixels per thread
};

template <int C, int D, bool PIN, bool POUT>
__global__ __launch_bounds__(256, C == 8 ? TT_X3N_MINW8 : 3) void k_x3n_conv(const void* __restrict__ xin, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, void* __restrict__ yout, int B, int H, int T,
                                                      int tiles_h, int tiles_t, int ntiles) {
    using G = XN<C, D>;
    constexpr int R = G::R, RW = G::RW, PPX = G::PPX, NTH = G::NTH, NSPAN = G::NSPAN, SPAN = G::SPAN;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    // this lane as B operand / D rows: pixel slot and first of its four channels;  as A operand: row n = (slot, output channel)
    const int slot = C == 8 ? (g >> 1) : g, c0 = C == 8 ? 4 * (g & 1) : 0;
    const int a_slot = n / C, a_co = n % C;

    // ---- weights, both planes, in registers.  C = 8: block-diagonal W (x) I_2, this lane's four k of row n (per tap; the 3x3 product
    // takes two taps per K = 32 instruction).  C = 4: the plain 4 x 4 matrix, row lane % 4 (the sixteen-block instruction). ----
    auto wsel = [&](int co_blockdiag, int ci) -> bool { (void)co_blockdiag; (void)ci; return C == 4 || a_slot == slot; };
    const int w_row = C == 4 ? (lane & 3) : a_co;
    s16x4 AH[9], AL[9], A2H, A2L;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        e16x4 qh, ql;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            e16 h_, l_;
            split(wsel(0, 0) ? w1[(w_row * C + c0 + k) * 9 + tap] : 0.f, h_, l_);
            qh[k] = h_; ql[k] = l_;
        }
        AH[tap] = __builtin_bit_cast(s16x4, qh);
        AL[tap] = __builtin_bit_cast(s16x4, ql);
    }
    {
        e16x4 qh, ql;
#pragma unroll
        for (int k = 0; k < 4; ++k) { e16 h_, l_; split(wsel(0, 0) ? w2[w_row * C + c0 + k] : 0.f, h_, l_); qh[k] = h_; ql[k] = l_; }
        A2H = __builtin_bit_cast(s16x4, qh);
        A2L = __builtin_bit_cast(s16x4, ql);
    }
    // one K = 16 / K = 4-per-block product of the shape this width uses
    auto mm = [&](s16x4 a, s16x4 b, f32x4 c) { if constexpr (C == 4) return mma4(a, b, c); else return mma16(a, b, c); };
    auto cat8 = [&](s16x4 lo4, s16x4 hi4) { return __builtin_bit_cast(e16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7)); };
    // C = 8: BOTH cross terms of a tap in one K = 32 instruction -- A = [Wlo | Whi] against B = [xhi ; xlo], the two planes of the lane's
    // channels exactly as one 16-byte LDS read pair returns them (no register shuffling); the main term Whi xhi stays a K = 16 product
    // on B's first half.  Two matrix instructions per tap and span instead of three.
    e16x8 AX8[C == 8 ? 9 : 1], A2X8;
    if constexpr (C == 8) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) AX8[tap] = cat8(AL[tap], AH[tap]);
        A2X8 = cat8(A2L, A2H);
    }
    float b1r[4], b2r[4];                                        // biases of the four channels this lane ends up with
#pragma unroll
    for (int r = 0; r < 4; ++r) { b1r[r] = b1[c0 + r]; b2r[r] = b2[c0 + r]; }
    // the lane's four channels (both planes) of the pixel in image (row, col)
    auto ldpx = [&](int row, int col, s16x4& hi, s16x4& lo) {
        if constexpr (C == 4) {
            const e16x8 v = *reinterpret_cast<const e16x8*>(smem + ((long)row * RW + col) * 16);
            hi = __builtin_bit_cast(s16x4, __builtin_shufflevector(v, v, 0, 1, 2, 3));
            lo = __builtin_bit_cast(s16x4, __builtin_shufflevector(v, v, 4, 5, 6, 7));
        } else {
            hi = *reinterpret_cast<const s16x4*>(smem + ((long)(row * 2 + 0) * RW + col) * 16 + 2 * c0);
            lo = *reinterpret_cast<const s16x4*>(smem + ((long)(row * 2 + 1) * RW + col) * 16 + 2 * c0);
        }
    };
    const e16* zero = reinterpret_cast<const e16*>(&g_wzero16);

    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        int tile = xcd_order(v, ntiles);
        const int tt = tile % tiles_t; tile /= tiles_t;
        const int th = tile % tiles_h;
        const int b = tile / tiles_h, h0 = th * G::TH, t0 = tt * G::TW;
        __syncthreads();                                         // the previous tile has been consumed
        if constexpr (PIN) {
            // fp32 planar -> split halves in LDS: C coalesced loads per pixel, all of a thread's pixels requested first
            const float* xp = static_cast<const float*>(xin) + (long)b * C * H * T;
            float f[G::NITP][C];
#pragma unroll
            for (int it = 0; it < G::NITP; ++it) {
                const int i = it * NTH + tid, row = i / RW, col = i - row * RW;
                const int h = h0 - D + row, t = t0 - D + col;
                const bool ok = i < G::NPIX && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
                const long o = ok ? (long)h * T + t : 0;
#pragma unroll
                for (int c = 0; c < C; ++c) { const float q = xp[(long)c * H * T + o]; f[it][c] = ok ? q : 0.f; }
            }
#pragma unroll
            for (int it = 0; it < G::NITP; ++it) {
                const int i = it * NTH + tid, row = i / RW, col = i - row * RW;
                if (i >= G::NPIX) continue;
                e16x8 ph, pl;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    e16 h_, l_; split(f[it][c], h_, l_);
                    if constexpr (C == 4) { ph[c] = h_; ph[4 + c] = l_; } else { ph[c] = h_; pl[c] = l_; }
                }
                if constexpr (C == 4) {
                    *reinterpret_cast<e16x8*>(smem + ((long)row * RW + col) * 16) = ph;
                } else {
                    *reinterpret_cast<e16x8*>(smem + ((long)(row * 2 + 0) * RW + col) * 16) = ph;
                    *reinterpret_cast<e16x8*>(smem + ((long)(row * 2 + 1) * RW + col) * 16) = pl;
                }
            }
        } else {
            const unsigned char* xb = static_cast<const unsigned char*>(xin) + (long)b * H * T * G::PXB;
            for (int i = wave * 64; i < G::NPR; i += NTH) {
                const int p = i + lane;
                const int rp = p / RW, col = p - rp * RW, row = rp / PPX, pc = rp - row * PPX;
                const int h = h0 - D + row, t = t0 - D + col;
                const bool ok = p < G::NP && (unsigned)h < (unsigned)H && (unsigned)t < (unsigned)T;
                glds16(ok ? (const void*)(xb + ((long)h * T + t) * G::PXB + pc * 16) : (const void*)zero, smem + (long)i * 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();

#pragma unroll 1
        for (int part = 0; part < 4 / R; ++part) {
            const int r0 = 4 * wave + R * part;                  // rows r0 .. r0 + R - 1 of the tile
            if (h0 + r0 >= H) break;
            f32x4 am[R][NSPAN], al[R][NSPAN];                    // hi x hi (bias as initial value) and the two cross terms
#pragma unroll
            for (int rr = 0; rr < R; ++rr)
#pragma unroll
                for (int sp = 0; sp < NSPAN; ++sp) {
                    am[rr][sp] = f32x4{b1r[0], b1r[1], b1r[2], b1r[3]};
                    al[rr][sp] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            // operands one step ahead; the empty asm statements pin that order (left alone, the compiler hoists the LDS reads of all nine
            // taps to the top of the loop and spills 300 bytes per lane at C = 8 to hold them).  A step = one tap (C = 4) or two (C = 8:
            // v_mfma_f32_16x16x32_f16 takes K = 32 = two taps x (2 slots x 8 channels) at the issue cost of one K = 16 instruction)
            typedef typename std::conditional<C == 8, e16x8, s16x4>::type bop_t;       // C = 8: [hi 4 | lo 4] of the lane's channels in one operand
            bop_t xh[R][NSPAN], nh[R][NSPAN];
            s16x4 xl[R][NSPAN], nl[R][NSPAN];                      // C = 4: the lo plane separately
            auto ldstep = [&](int tap, bop_t (&qh)[R][NSPAN], s16x4 (&ql)[R][NSPAN]) {
                const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
#pragma unroll
                    for (int sp = 0; sp < NSPAN; ++sp) {
                        s16x4 h4, l4;
                        ldpx(r0 + rr + kh * D, sp * SPAN + n + 16 * slot + kw * D, h4, l4);
                        if constexpr (C == 8) { qh[rr][sp] = cat8(h4, l4); ql[rr][sp] = l4; }
                        else { qh[rr][sp] = h4; ql[rr][sp] = l4; }
                    }
            };
            ldstep(0, xh, xl);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                asm volatile("" ::: "memory");
                if (tap + 1 < 9) ldstep(tap + 1, nh, nl);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
#pragma unroll
                    for (int sp = 0; sp < NSPAN; ++sp) {
                        if constexpr (C == 8) {
                            const s16x4 bh = __builtin_bit_cast(s16x4, __builtin_shufflevector(xh[rr][sp], xh[rr][sp], 0, 1, 2, 3));
                            am[rr][sp] = mma16(AH[tap], bh, am[rr][sp]);
                            al[rr][sp] = mma32(AX8[tap], xh[rr][sp], al[rr][sp]);
                        } else {
                            am[rr][sp] = mm(AH[tap], xh[rr][sp], am[rr][sp]);
                            al[rr][sp] = mm(AH[tap], xl[rr][sp], al[rr][sp]);
                            al[rr][sp] = mm(AL[tap], xh[rr][sp], al[rr][sp]);
                        }
                    }
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
#pragma unroll
                    for (int sp = 0; sp < NSPAN; ++sp) { xh[rr][sp] = nh[rr][sp]; xl[rr][sp] = nl[rr][sp]; }
            }
            // ---- ELU, 1x1 product (the accumulators of the 3x3 product ARE its B operand), ELU, residual add, store ----
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int h = h0 + r0 + rr;
                if (h >= H) break;
#pragma unroll
                for (int sp = 0; sp < NSPAN; ++sp) {
                    const int col = sp * SPAN + n + 16 * slot;   // this lane's pixel of the row
                    const int t = t0 + col;
                    e16x4 hh, hl;
                    float hv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) hv[r] = elu1(__builtin_fmaf(al[rr][sp][r], LO_INV, am[rr][sp][r]));
                    split_v<true>(hv, hh, hl);
                    const f32x4 zm = mm(A2H, __builtin_bit_cast(s16x4, hh), f32x4{b2r[0], b2r[1], b2r[2], b2r[3]});
                    f32x4 zl;
                    if constexpr (C == 8) {
                        zl = mma32(A2X8, __builtin_shufflevector(hh, hl, 0, 1, 2, 3, 4, 5, 6, 7), f32x4{0.f, 0.f, 0.f, 0.f});
                    } else {
                        zl = mm(A2H, __builtin_bit_cast(s16x4, hl), f32x4{0.f, 0.f, 0.f, 0.f});
                        zl = mm(A2L, __builtin_bit_cast(s16x4, hh), zl);
                    }
                    s16x4 ch_, cl_;
                    ldpx(r0 + rr + D, col + D, ch_, cl_);
                    const e16x4 c_h = __builtin_bit_cast(e16x4, ch_), c_l = __builtin_bit_cast(e16x4, cl_);
                    float out[4], res[4];
                    join_v(c_h, c_l, res);
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[r] = elu1(__builtin_fmaf(zl[r], LO_INV, zm[r])) + res[r];
                    if (t >= T) continue;
                    if constexpr (POUT) {
                        float* yp = static_cast<float*>(yout) + (((long)b * C + c0) * H + h) * T + t;
#pragma unroll
                        for (int r = 0; r < 4; ++r) yp[(long)r * H * T] = out[r];
                    } else {
                        e16* y = static_cast<e16*>(yout) + (((long)b * H + h) * T + t) * 2 * C;
                        e16x4 oh, ol;
                        split_v(out, oh, ol);
                        if constexpr (C == 4) {
                            *reinterpret_cast<e16x8*>(y) = __builtin_shufflevector(oh, ol, 0, 1, 2, 3, 4, 5, 6, 7);
                        } else {
                            *reinterpret_cast<e16x4*>(y + c0) = oh;
                            *reinterpret_cast<e16x4*>(y + C + c0) = ol;
                        }
                    }
                }
            }
        }
    }
}

template <int C, int D, bool PIN, bool POUT>
int launch_x3n(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y, int B, int H, int T, hipStream_t st) {
    using G = XN<C, D>;
    const int tiles_h = (H + G::TH - 1) / G::TH, tiles_t = (T + G::TW - 1) / G::TW, ntiles = B * tiles_h * tiles_t;
    static AttrOnce once;
    auto kern = k_x3n_conv<C, D, PIN, POUT>;
    if (int rc = raise_lds(kern, G::LDS_BYTES, once)) return rc;
    static const int per_cu = tt_tune("TTRAP_X3N_PER_CU", C == 8 ? 3 : 4);     // registers: <= 152 (C = 8) / <= 99 (C = 4) VGPRs
    hipLaunchKernelGGL(kern, dim3(grid_for(ntiles, G::LDS_BYTES, per_cu)), dim3(G::NTH), G::LDS_BYTES, st, x, w1, b1, w2, b2, y, B, H, T,
                       tiles_h, tiles_t, ntiles);
    TT_LAUNCH_CHECK();
    return 0;
}
template <int C, int D>
int x3n_io(const void* x, bool pin, const float* w1, const float* b1, const float* w2, const float* b2, void* y, bool pout, int B, int H,
           int T, hipStream_t st) {
    if (pin) return pout ? launch_x3n<C, D, true, true>(x, w1, b1, w2, b2, y, B, H, T, st) : launch_x3n<C, D, true, false>(x, w1, b1, w2, b2, y, B, H, T, st);
    return pout ? launch_x3n<C, D, false, true>(x, w1, b1, w2, b2, y, B, H, T, st) : launch_x3n<C, D, false, false>(x, w1, b1, w2, b2, y, B, H, T, st);
}
template <int C>
int x3n_d(const void* x, bool pin, const float* w1, const float* b1, const float* w2, const float* b2, void* y, bool pout, int B, int H, int T,
          int d, hipStream_t st) {
    switch (d) {
        case 1: return x3n_io<C, 1>(x, pin, w1, b1, w2, b2, y, pout, B, H, T, st);
        case 2: return x3n_io<C, 2>(x, pin, w1, b1, w2, b2, y, pout, B, H, T, st);
        case 3: return x3n_io<C, 3>(x, pin, w1, b1, w2, b2, y, pout, B, H, T, st);
    }
    return TT_E_UNSUPPORTED;
}
inline bool x3n_shape_ok(int B, int C, int H, int T) { return B > 0 && H > 0 && T > 0 && (C == 4 || C == 8) && (long)B * H * T < (1l << 34); }

// ---- strided layers between and above the wide levels ------------------------------------------------------------------------------
// EncoderBlock.sconv = ELU(Conv2d(C, 2C, (4,1), stride (2,1))) (reference modules.py:626-630) and DecoderBlock.tconv =
// ELU(ConvTranspose2d(2C, C, (4,1), stride (2,1), output_padding)) (modules.py:683-688) with split operands, so that a chain
// level -> strided layer -> level never leaves the x3 layout (no pack / unpack passes, no fp32 matrix instructions).
// These layers are pointwise in time: an output pixel (h, t) reads 4 (2) input pixels of the same frame.  No LDS for the activations: a
// wave owns 16 consecutive frames of one clip and a run of output rows, all weights of its co-tiles sit in registers as hi / lo pairs
// in operand order (split once per workgroup, through LDS), the K = 32 slices of the input rows come straight from HBM / L2 in
// B-operand order, the next unit's NEW slices are requested before this unit's products, and the slices two neighbouring units share
// stay in registers (sliding window).  Input: an x3 tensor (16 bytes per lane, plane and slice) or -- PIN, the layer that ENTERS the
// split-operand part of the network -- an fp32 planar tensor (eight 4-byte loads per slice, split in registers).
//   DOWN  C = 8 (PIN): its own kernel below (k_x3_sconv8: K = 16 half-slices of two rows, so that the window works).
//   DOWN  C = 16: two slices of two rows each; 32 output channels = 2 co-tiles.
//   DOWN  C = 32: four slices; 64 output channels = 4 co-tiles.
//   UP   2C = 32: output rows 2m, 2m + 1 both read input rows m (tap = parity) and m - 1 (tap = parity + 2): two slices per parity.
//   UP   2C = 64 (PIN): the same with two slices per input row; 32 output channels = 2 co-tiles.
template <int CIN, bool UP> struct XS {
    static constexpr int COUT = UP ? CIN / 2 : 2 * CIN;
    static constexpr int NCT = COUT / 16;                        // co-tiles in all
    static constexpr int NCTW = NCT;                             // co-tiles per wave: all of them (two waves with two co-tiles each on the
                                                                 // same frames loaded every operand twice: 32 -> 64 0.48 -> 0.35 ms)
    static constexpr int NSPLIT = NCT / NCTW;
    static constexpr int SPR = CIN > 32 ? CIN / 32 : 1;          // slices per input row
    static constexpr int NKS = UP ? 2 * SPR : (4 * CIN) / 32;    // K = 32 slices per unit
    static constexpr int NNEW = NKS == 1 ? 1 : NKS / 2;          // slices a unit does not share with its predecessor
    static constexpr int NSET = UP ? 2 : 1;                      // weight sets (output row parity)
    static constexpr int NCH = 4 * NCTW;                         // channels a lane ends up with
};

template <int CIN, bool UP, bool PIN, bool PLANAR>
__global__ __launch_bounds__(NT, (CIN >= 64 || (CIN == 32 && !UP)) ? 2 : 4) void k_x3_sconv(const void* __restrict__ xin, const float* __restrict__ w,
                                                                                      const float* __restrict__ bias, void* __restrict__ yout,
                                                                                      int B, int Hin, int Hout, int T, int nchunks, int rch) {
    using S = XS<CIN, UP>;
    constexpr int COUT = S::COUT, NCTW = S::NCTW, NKS = S::NKS, NSET = S::NSET, NCH = S::NCH, NNEW = S::NNEW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    // task of this workgroup: (clip, 64-frame group, co split, row chunk); the four waves take the four 16-frame groups
    int task = blockIdx.x;
    const int rc = task % nchunks; task /= nchunks;
    const int sp = task % S::NSPLIT; task /= S::NSPLIT;
    const int tgroups = (T + 63) / 64;
    const int tg = task % tgroups, b = task / tgroups;
    const int t = tg * 64 + wave * 16 + n;
    const bool tv = t < T;
    const int tc = tv ? t : T - 1;                               // clamped: loads stay inside the tensor, stores are masked

    // slice ks as seen by lane group lg: input row relative to the unit's first row (DOWN: 2u, UP: u), tap, first of its 8 channels
    auto row_off = [](int ks, int lg) { return UP ? -(ks / S::SPR) : CIN == 8 ? lg : CIN == 16 ? 2 * ks + (lg >> 1) : ks; };
    auto chan0 = [](int ks, int lg) { return CIN == 8 ? 0 : CIN == 16 ? 8 * (lg & 1) : CIN == 32 ? 8 * lg : 32 * (ks % S::SPR) + 8 * lg; };

    // ---- weights of this workgroup's co-tiles: split once by the four waves together (through LDS), then to registers ----
    // channel of row m of co-tile ct (global index): a lane's rows 4g..4g+3 of its co-tiles are consecutive channels
    // (COUT = 64: the two co-tiles of a wave cover 32 CONSECUTIVE channels, 8 per lane group: whole 64-byte runs per pixel and plane)
    auto cmap = [&](int ct, int m) { return COUT == 16 ? m : COUT == 32 ? 8 * (m >> 2) + 4 * ct + (m & 3) : 32 * (ct >> 1) + 8 * (m >> 2) + 4 * (ct & 1) + (m & 3); };
    constexpr int NE = NSET * NKS * NCTW;
    __shared__ __align__(16) unsigned char wimg[NE * 2 * 64 * 16];
    for (int e = threadIdx.x; e < NE * 64; e += NT) {
        const int l = e & 63, ent = e >> 6, ct = ent % NCTW, ks = (ent / NCTW) % NKS, st_ = ent / (NCTW * NKS);
        const int ln = l & 15, lg = l >> 4;
        const int co = cmap(sp * NCTW + ct, ln);
        e16x8 qh, ql;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ci = chan0(ks, lg) + j;
            const float wv = UP ? w[(ci * COUT + co) * 4 + st_ - 2 * row_off(ks, lg)]           // (Cin, Cout, 4, 1): tap = parity + 2 (m - row)
                                : w[(co * CIN + ci) * 4 + row_off(ks, lg)];                      // (Cout, Cin, 4, 1): tap = row - 2 ho
            e16 h, lo_; split(wv, h, lo_); qh[j] = h; ql[j] = lo_;
        }
        *reinterpret_cast<e16x8*>(wimg + ((long)(ent * 2 + 0) * 64 + l) * 16) = qh;
        *reinterpret_cast<e16x8*>(wimg + ((long)(ent * 2 + 1) * 64 + l) * 16) = ql;
    }
    __syncthreads();
    e16x8 AH[NSET][NKS][NCTW], AL[NSET][NKS][NCTW];
#pragma unroll
    for (int st_ = 0; st_ < NSET; ++st_)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int ct = 0; ct < NCTW; ++ct) {
                const int ent = (st_ * NKS + ks) * NCTW + ct;
                AH[st_][ks][ct] = *reinterpret_cast<const e16x8*>(wimg + ((long)(ent * 2 + 0) * 64 + lane) * 16);
                AL[st_][ks][ct] = *reinterpret_cast<const e16x8*>(wimg + ((long)(ent * 2 + 1) * 64 + lane) * 16);
            }
    float br[NCTW][4];
#pragma unroll
    for (int ct = 0; ct < NCTW; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) br[ct][r] = bias[cmap(sp * NCTW + ct, 4 * g + r)];

    // B operand (both planes) of slice ks for unit u; rows outside the input read row 0 and are zeroed
    auto ldb = [&](int u, int ks, e16x8& vh, e16x8& vl) {
        const int row = (UP ? u : 2 * u) + row_off(ks, g);
        const bool ok = (unsigned)row < (unsigned)Hin;
        const int rw = ok ? row : 0, c0_ = chan0(ks, g);
        if constexpr (PIN) {
            const float* xp = static_cast<const float*>(xin) + (((long)b * CIN + c0_) * Hin + rw) * T + tc;
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = xp[(long)j * Hin * T];
#pragma unroll
            for (int j = 0; j < 8; ++j) { e16 h, l; split(ok ? f[j] : 0.f, h, l); vh[j] = h; vl[j] = l; }
        } else {
            const e16* xp = static_cast<const e16*>(xin) + (((long)b * Hin + rw) * T + tc) * 2 * CIN + c0_;
            vh = *reinterpret_cast<const e16x8*>(xp);
            vl = *reinterpret_cast<const e16x8*>(xp + CIN);
            if (!ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { vh[j] = (e16)0.f; vl[j] = (e16)0.f; }
            }
        }
    };
    const int npairs = UP ? (Hout + 1) / 2 : Hout;               // units the row loop walks: output rows, or pairs of them
    const int u0 = rc * rch, u1 = u0 + rch < npairs ? u0 + rch : npairs;
    if (u0 >= npairs) return;
    // window positions: DOWN keeps the slices in ascending row order (new ones enter at the top end); UP has the slices of row m first
    // (new), then those of row m - 1 (the previous unit's first half)
    e16x8 bh[NKS], bl[NKS], nh[NNEW], nl[NNEW];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) ldb(u0, ks, bh[ks], bl[ks]);
    for (int u = u0; u < u1; ++u) {
        if (u + 1 < u1) {
#pragma unroll
            for (int i = 0; i < NNEW; ++i) ldb(u + 1, UP ? i : NKS - NNEW + i, nh[i], nl[i]);   // the slices unit u does not hold
        }
#pragma unroll
        for (int st_ = 0; st_ < NSET; ++st_) {
            const int ho = UP ? 2 * u + st_ : u;
            f32x4 am[NCTW], al[NCTW];
#pragma unroll
            for (int ct = 0; ct < NCTW; ++ct) { am[ct] = f32x4{br[ct][0], br[ct][1], br[ct][2], br[ct][3]}; al[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int ct = 0; ct < NCTW; ++ct) {
                    am[ct] = mma32(AH[st_][ks][ct], bh[ks], am[ct]);
                    al[ct] = mma32(AH[st_][ks][ct], bl[ks], al[ct]);
                }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int ct = 0; ct < NCTW; ++ct) al[ct] = mma32(AL[st_][ks][ct], bh[ks], al[ct]);
            if (ho >= Hout || !tv) continue;
            float out[NCH];
#pragma unroll
            for (int ct = 0; ct < NCTW; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[4 * ct + r] = elu1(__builtin_fmaf(al[ct][r], LO_INV, am[ct][r]));
            // a lane's rows 4g..4g+3 of two neighbouring co-tiles are eight consecutive channels (cmap): one 16-byte piece per plane
            constexpr int PW = NCTW >= 2 ? 2 : 1;
#pragma unroll
            for (int q = 0; q < NCTW / PW; ++q) {
                const int cb = cmap(sp * NCTW + PW * q, 4 * g);
                if constexpr (PLANAR) {
                    float* yp = static_cast<float*>(yout) + (((long)b * COUT + cb) * Hout + ho) * T + t;
#pragma unroll
                    for (int j = 0; j < 4 * PW; ++j) yp[(long)j * Hout * T] = out[4 * PW * q + j];
                } else {
                    e16* y = static_cast<e16*>(yout) + (((long)b * Hout + ho) * T + t) * 2 * COUT + cb;
                    typename std::conditional<PW == 2, e16x8, e16x4>::type oh, ol;
#pragma unroll
                    for (int j = 0; j < 4 * PW; ++j) { e16 a_, b_; split(out[4 * PW * q + j], a_, b_); oh[j] = a_; ol[j] = b_; }
                    *reinterpret_cast<decltype(oh)*>(y) = oh;
                    *reinterpret_cast<decltype(ol)*>(y + COUT) = ol;
                }
            }
        }
        // shift the window
        if constexpr (NNEW == NKS) {
#pragma unroll
            for (int i = 0; i < NKS; ++i) { bh[i] = nh[i]; bl[i] = nl[i]; }
        } else if constexpr (UP) {
#pragma unroll
            for (int i = 0; i < NNEW; ++i) { bh[NNEW + i] = bh[i]; bl[NNEW + i] = bl[i]; bh[i] = nh[i]; bl[i] = nl[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < NNEW; ++i) { bh[i] = bh[NNEW + i]; bl[i] = bl[NNEW + i]; bh[NNEW + i] = nh[i]; bl[NNEW + i] = nl[i]; }
        }
    }
}

// EncoderBlock.sconv 8 -> 16 from an fp32 planar tensor (the layer that enters the split-operand part of the encoder).  K = 4 rows x 8
// channels = 32 would be ONE slice with the four rows on the four lane groups -- no two units could share registers and every input row
// would be fetched twice, four bytes per lane at a time (0.49 ms).  Two K = 16 half-slices of two rows each (v_mfma_f32_16x16x16_f16)
// restore the sliding window: the lower half-slice of one output row is the upper one of the next.
template <bool PLANAR>
__global__ __launch_bounds__(NT, 4) void k_x3_sconv8(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                     void* __restrict__ yout, int B, int Hin, int Hout, int T, int nchunks, int rch) {
    constexpr int CIN = 8, COUT = 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    int task = blockIdx.x;
    const int rc = task % nchunks; task /= nchunks;
    const int tgroups = (T + 63) / 64;
    const int tg = task % tgroups, b = task / tgroups;
    const int t = tg * 64 + wave * 16 + n;
    const bool tv = t < T;
    const int tc = tv ? t : T - 1;
    // A operands: row n = output channel, k = 4 g + j  ->  input row (g >> 1) of the half-slice, channel 4 (g & 1) + j
    s16x4 AH[2], AL[2];
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        e16x4 qh, ql;
#pragma unroll
        for (int j = 0; j < 4; ++j) { e16 h, l; split(w[(n * CIN + 4 * (g & 1) + j) * 4 + 2 * hs + (g >> 1)], h, l); qh[j] = h; ql[j] = l; }
        AH[hs] = __builtin_bit_cast(s16x4, qh); AL[hs] = __builtin_bit_cast(s16x4, ql);
    }
    const float br[4] = {bias[4 * g], bias[4 * g + 1], bias[4 * g + 2], bias[4 * g + 3]};
    auto ldb = [&](int u, int hs, s16x4& vh, s16x4& vl) {
        const int row = 2 * u + 2 * hs + (g >> 1);
        const bool ok = row < Hin;
        const float* xp = x + (((long)b * CIN + 4 * (g & 1)) * Hin + (ok ? row : 0)) * T + tc;
        float f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = xp[(long)j * Hin * T];
        e16x4 qh, ql;
#pragma unroll
        for (int j = 0; j < 4; ++j) { e16 h, l; split(ok ? f[j] : 0.f, h, l); qh[j] = h; ql[j] = l; }
        vh = __builtin_bit_cast(s16x4, qh); vl = __builtin_bit_cast(s16x4, ql);
    };
    const int u0 = rc * rch, u1 = u0 + rch < Hout ? u0 + rch : Hout;
    if (u0 >= Hout) return;
    s16x4 bh[2], bl[2], nh, nl;
    ldb(u0, 0, bh[0], bl[0]); ldb(u0, 1, bh[1], bl[1]);
    for (int u = u0; u < u1; ++u) {
        if (u + 1 < u1) ldb(u + 1, 1, nh, nl);
        f32x4 am = f32x4{br[0], br[1], br[2], br[3]}, al = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) { am = mma16(AH[hs], bh[hs], am); al = mma16(AH[hs], bl[hs], al); }
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) al = mma16(AL[hs], bh[hs], al);
        if (tv) {
            float out[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] = elu1(__builtin_fmaf(al[r], LO_INV, am[r]));
            if constexpr (PLANAR) {
                float* yp = static_cast<float*>(yout) + (((long)b * COUT + 4 * g) * Hout + u) * T + t;
#pragma unroll
                for (int j = 0; j < 4; ++j) yp[(long)j * Hout * T] = out[j];
            } else {
                e16* y = static_cast<e16*>(yout) + (((long)b * Hout + u) * T + t) * 2 * COUT + 4 * g;
                e16x4 oh, ol;
#pragma unroll
                for (int j = 0; j < 4; ++j) { e16 a_, b_; split(out[j], a_, b_); oh[j] = a_; ol[j] = b_; }
                *reinterpret_cast<e16x4*>(y) = oh;
                *reinterpret_cast<e16x4*>(y + COUT) = ol;
            }
        }
        bh[0] = bh[1]; bl[0] = bl[1]; bh[1] = nh; bl[1] = nl;
    }
}

template <int CIN, bool UP, bool PIN>
int launch_x3s(const void* x, const float* w, const float* bias, void* y, bool planar_out, int B, int Hin, int Hout, int T, hipStream_t st) {
    using S = XS<CIN, UP>;
    // a task = (clip, 64 frames, co split, run of output rows): runs as long as the grid still holds ~16 workgroups per CU (the weight
    // split at the head of a workgroup is amortised over the run), never shorter than 8 rows
    const int units = UP ? (Hout + 1) / 2 : Hout;
    const long base = (long)B * ((T + 63) / 64) * S::NSPLIT;
    long want = (16l * tt_cus() + base - 1) / base;
    if (want < 1) want = 1;
    if (want > (units + 7) / 8) want = (units + 7) / 8;
    const int rch = (int)((units + want - 1) / want), nchunks = (units + rch - 1) / rch;
    const long grid = base * nchunks;
    if (grid > 0x7fffffffl) return TT_E_UNSUPPORTED;
    if constexpr (CIN == 8) {
        if (planar_out) hipLaunchKernelGGL(k_x3_sconv8<true>, dim3((unsigned)grid), dim3(NT), 0, st, (const float*)x, w, bias, y, B, Hin, Hout, T, nchunks, rch);
        else hipLaunchKernelGGL(k_x3_sconv8<false>, dim3((unsigned)grid), dim3(NT), 0, st, (const float*)x, w, bias, y, B, Hin, Hout, T, nchunks, rch);
    } else if (planar_out)
        hipLaunchKernelGGL((k_x3_sconv<CIN, UP, PIN, true>), dim3((unsigned)grid), dim3(NT), 0, st, x, w, bias, y, B, Hin, Hout, T, nchunks, rch);
    else
        hipLaunchKernelGGL((k_x3_sconv<CIN, UP, PIN, false>), dim3((unsigned)grid), dim3(NT), 0, st, x, w, bias, y, B, Hin, Hout, T, nchunks, rch);
    TT_LAUNCH_CHECK();
    return 0;
}

// ---- latent heads ---------------------------------------------------------------------------------------------------------------------
// Encoder.convlat = Conv2d(C, D, (E,1)) (reference modules.py:446): z[d][t] = b[d] + sum_{e,c} W[d][c][e] x[c][e][t], and Decoder.convin =
// ELU(ConvTranspose2d(D + 1, C, (E,1))) (modules.py:534): y[co][e][t] = ELU(b[co] + sum_d W[d][co][e] z[d][t]) -- per frame a
// (D x C E) and a (C E x D + 1) matrix-vector product; the fp32 GEMM runs them at 46 % of the fp32 matrix peak (0.69 ms each at the
// inference batch).  With split operands they are memory-bound: the x3 embedding (E rows of C channels) is read / written once.
// The weights are split ONCE per call into an operand-order image (k_x3_lat_wprep: [step][slice][co-tile][plane][lane] x 16 bytes; a
// step = one frequency row e) which every workgroup streams through LDS by LDS-DMA, double-buffered, one step ahead; a workgroup =
// eight waves = 128 consecutive frames of one clip.
//   encode: the K loop runs over the E rows (B operands straight from the x3 tensor, one row ahead), all D outputs accumulate in
//           registers (D / 16 co-tiles x two accumulator sets);
//   decode: the latent vector of the wave's 16 frames is split once into registers (K = D), the loop runs over the E OUTPUT rows; the
//           extra input channel (the transcription switch, D + 1) enters as a rank-1 term on the vector pipe.
template <int C, int D> struct XLat {
    static constexpr int SPR = C / 32;                           // K = 32 slices per frequency row (encode)
    static constexpr int KS = D / 32;                            // K = 32 slices of the latent vector (decode)
    static constexpr int NCT_E = D / 16, NCT_D = C / 16;         // co-tiles: encode (latent channels), decode (embedding channels)
    static constexpr int ENT_E = SPR * NCT_E, ENT_D = KS * NCT_D;     // 2-KB image entries (hi + lo plane of one operand) per step
};
constexpr int XLAT_NT = 512;

// MODE 0: encode image from w (D, C, E, 1); MODE 1: decode image from w (D + 1, C, E, 1) (rows d < D)
template <int C, int D, int MODE>
__global__ __launch_bounds__(256) void k_x3_lat_wprep(const float* __restrict__ w, unsigned char* __restrict__ img, int E) {
    using L = XLat<C, D>;
    constexpr int ENT = MODE == 0 ? L::ENT_E : L::ENT_D, NCT = MODE == 0 ? L::NCT_E : L::NCT_D;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= E * ENT * 64) return;
    const int l = i & 63, ent = (i >> 6) % ENT, e = i / (64 * ENT);
    const int ct = ent % NCT, ks = ent / NCT, ln = l & 15, lg = l >> 4;
    e16x8 qh, ql;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float wv;
        if (MODE == 0) wv = w[((long)(16 * ct + ln) * C + 32 * ks + 8 * lg + j) * E + e];
        else {
            const int co = C == 32 ? 8 * (ln >> 2) + 4 * ct + (ln & 3) : 16 * (ln >> 2) + 4 * ct + (ln & 3);
            wv = w[((long)(32 * ks + 8 * lg + j) * C + co) * E + e];
        }
        e16 h, lo_; split(wv, h, lo_); qh[j] = h; ql[j] = lo_;
    }
    unsigned char* dst = img + ((long)(e * ENT + ent) * 2) * 1024 + l * 16;
    *reinterpret_cast<e16x8*>(dst) = qh;
    *reinterpret_cast<e16x8*>(dst + 1024) = ql;
}

template <int C, int D>
__global__ __launch_bounds__(XLAT_NT, 2) void k_x3_latenc(const e16* __restrict__ x, const unsigned char* __restrict__ img,
                                                          const float* __restrict__ bias, float* __restrict__ z, int B, int E, int T) {
    using L = XLat<C, D>;
    constexpr int SPR = L::SPR, NCT = L::NCT_E, STEP = L::ENT_E * 2048;
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int tgroups = (T + 127) / 128;
    const int tg = blockIdx.x % tgroups, b = blockIdx.x / tgroups;
    const int t = tg * 128 + wave * 16 + n;
    const bool tv = t < T;
    const int tc = tv ? t : T - 1;
    auto stage = [&](int e, int buf) {
        for (int i = wave * 1024; i < STEP; i += XLAT_NT * 16) glds16(img + (long)e * STEP + i + lane * 16, smem + buf * STEP + i);
    };
    f32x4 am[NCT], al[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        am[ct] = bias ? f32x4{bias[16 * ct + 4 * g], bias[16 * ct + 4 * g + 1], bias[16 * ct + 4 * g + 2], bias[16 * ct + 4 * g + 3]}
                      : f32x4{0.f, 0.f, 0.f, 0.f};
        al[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const e16* xb = x + ((long)b * E * T + tc) * 2 * C + 8 * g;
    e16x8 bh[SPR], bl[SPR], nh[SPR], nl[SPR];
#pragma unroll
    for (int s_ = 0; s_ < SPR; ++s_) { bh[s_] = *reinterpret_cast<const e16x8*>(xb + 32 * s_); bl[s_] = *reinterpret_cast<const e16x8*>(xb + C + 32 * s_); }
    stage(0, 0);
    for (int e = 0; e < E; ++e) {
        const int buf = e & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                         // step e has landed; everyone has left step e - 1 (buffer buf ^ 1)
        if (e + 1 < E) {
            stage(e + 1, buf ^ 1);
            const e16* xn = xb + (long)(e + 1) * T * 2 * C;
#pragma unroll
            for (int s_ = 0; s_ < SPR; ++s_) { nh[s_] = *reinterpret_cast<const e16x8*>(xn + 32 * s_); nl[s_] = *reinterpret_cast<const e16x8*>(xn + C + 32 * s_); }
        }
        const unsigned char* A = smem + buf * STEP;
#pragma unroll
        for (int s_ = 0; s_ < SPR; ++s_)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const e16x8 wh = *reinterpret_cast<const e16x8*>(A + ((s_ * NCT + ct) * 2) * 1024 + lane * 16);
                const e16x8 wl = *reinterpret_cast<const e16x8*>(A + ((s_ * NCT + ct) * 2 + 1) * 1024 + lane * 16);
                am[ct] = mma32(wh, bh[s_], am[ct]);
                al[ct] = mma32(wh, bl[s_], al[ct]);
                al[ct] = mma32(wl, bh[s_], al[ct]);
            }
#pragma unroll
        for (int s_ = 0; s_ < SPR; ++s_) { bh[s_] = nh[s_]; bl[s_] = nl[s_]; }
    }
    if (!tv) return;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) z[((long)b * D + 16 * ct + 4 * g + r) * T + t] = __builtin_fmaf(al[ct][r], LO_INV, am[ct][r]);
}

template <int C, int D, bool PLANAR>
__global__ __launch_bounds__(XLAT_NT, 2) void k_x3_latdec(const float* __restrict__ z, int Dz, float fill, const float* __restrict__ w,
                                                          const unsigned char* __restrict__ img, const float* __restrict__ bias,
                                                          void* __restrict__ yout, int B, int E, int T) {
    using L = XLat<C, D>;
    constexpr int KS = L::KS, NCT = L::NCT_D, STEP = L::ENT_D * 2048, NCH = C == 32 ? 8 : 16;
    extern __shared__ __align__(16) unsigned char smem[];
    float* wx = reinterpret_cast<float*>(smem + 2 * STEP);       // [E][C]: the weights of the extra input channel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int tgroups = (T + 127) / 128;
    const int tg = blockIdx.x % tgroups, b = blockIdx.x / tgroups;
    const int t = tg * 128 + wave * 16 + n;
    const bool tv = t < T;
    const int tc = tv ? t : T - 1;
    auto stage = [&](int e, int buf) {
        for (int i = wave * 1024; i < STEP; i += XLAT_NT * 16) glds16(img + (long)e * STEP + i + lane * 16, smem + buf * STEP + i);
    };
    stage(0, 0);
    for (int i = threadIdx.x; i < E * C; i += XLAT_NT) { const int e = i / C, co = i - e * C; wx[i] = w[((long)D * C + co) * E + e]; }
    // the latent vector of this wave's frames, split once
    e16x8 bh[KS], bl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = z[((long)b * Dz + 32 * ks + 8 * g + j) * T + tc];
#pragma unroll
        for (int j = 0; j < 8; ++j) { e16 h, l; split(f[j], h, l); bh[ks][j] = h; bl[ks][j] = l; }
    }
    const float zx = Dz > D ? z[((long)b * Dz + D) * T + tc] : fill;
    auto cmap = [&](int ct, int m) { return C == 32 ? 8 * (m >> 2) + 4 * ct + (m & 3) : 16 * (m >> 2) + 4 * ct + (m & 3); };
    float br[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) br[ct][r] = bias ? bias[cmap(ct, 4 * g + r)] : 0.f;
    const int chbase = NCH * g;                                  // the lane's NCH consecutive channels: cmap(ct, 4 g + r) = chbase + 4 ct + r
    for (int e = 0; e < E; ++e) {
        const int buf = e & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (e + 1 < E) stage(e + 1, buf ^ 1);
        const unsigned char* A = smem + buf * STEP;
        f32x4 am[NCT], al[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { am[ct] = f32x4{br[ct][0], br[ct][1], br[ct][2], br[ct][3]}; al[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const e16x8 wh = *reinterpret_cast<const e16x8*>(A + ((ks * NCT + ct) * 2) * 1024 + lane * 16);
                const e16x8 wl = *reinterpret_cast<const e16x8*>(A + ((ks * NCT + ct) * 2 + 1) * 1024 + lane * 16);
                am[ct] = mma32(wh, bh[ks], am[ct]);
                al[ct] = mma32(wh, bl[ks], al[ct]);
                al[ct] = mma32(wl, bh[ks], al[ct]);
            }
        if (!tv) continue;
        float out[NCH];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[4 * ct + r] = elu1(__builtin_fmaf(wx[e * C + chbase + 4 * ct + r], zx, __builtin_fmaf(al[ct][r], LO_INV, am[ct][r])));
        if constexpr (PLANAR) {
            float* yp = static_cast<float*>(yout) + (((long)b * C + chbase) * E + e) * T + t;
#pragma unroll
            for (int j = 0; j < NCH; ++j) yp[(long)j * E * T] = out[j];
        } else {
            e16* y = static_cast<e16*>(yout) + (((long)b * E + e) * T + t) * 2 * C + chbase;
#pragma unroll
            for (int q = 0; q < NCH / 8; ++q) {
                e16x8 oh, ol;
#pragma unroll
                for (int j = 0; j < 8; ++j) { e16 a_, b_; split(out[8 * q + j], a_, b_); oh[j] = a_; ol[j] = b_; }
                *reinterpret_cast<e16x8*>(y + 8 * q) = oh;
                *reinterpret_cast<e16x8*>(y + C + 8 * q) = ol;
            }
        }
    }
}

template <int C, int D>
int launch_latenc(const e16* x, const float* w, const float* bias, float* z, unsigned char* ws, int B, int E, int T, hipStream_t st) {
    using L = XLat<C, D>;
    const int n = E * L::ENT_E * 64;
    hipLaunchKernelGGL((k_x3_lat_wprep<C, D, 0>), dim3((n + 255) / 256), dim3(256), 0, st, w, ws, E);
    TT_LAUNCH_CHECK();
    static AttrOnce once;
    auto kern = k_x3_latenc<C, D>;
    constexpr int LDS = 2 * L::ENT_E * 2048;
    if (int rc = raise_lds(kern, LDS, once)) return rc;
    hipLaunchKernelGGL(kern, dim3(B * ((T + 127) / 128)), dim3(XLAT_NT), LDS, st, x, (const unsigned char*)ws, bias, z, B, E, T);
    TT_LAUNCH_CHECK();
    return 0;
}
template <int C, int D>
int launch_latdec(const float* z, int Dz, float fill, const float* w, const float* bias, void* y, bool planar, unsigned char* ws, int B,
                  int E, int T, hipStream_t st) {
    using L = XLat<C, D>;
    const int n = E * L::ENT_D * 64;
    hipLaunchKernelGGL((k_x3_lat_wprep<C, D, 1>), dim3((n + 255) / 256), dim3(256), 0, st, w, ws, E);
    TT_LAUNCH_CHECK();
    const int LDS = 2 * L::ENT_D * 2048 + E * C * 4;
    static AttrOnce once_p, once_x;
    const dim3 grid(B * ((T + 127) / 128));
    if (planar) {
        auto kern = k_x3_latdec<C, D, true>;
        if (int rc = raise_lds(kern, 160 * 1024, once_p)) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(XLAT_NT), LDS, st, z, Dz, fill, w, (const unsigned char*)ws, bias, y, B, E, T);
    } else {
        auto kern = k_x3_latdec<C, D, false>;
        if (int rc = raise_lds(kern, 160 * 1024, once_x)) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(XLAT_NT), LDS, st, z, Dz, fill, w, (const unsigned char*)ws, bias, y, B, E, T);
    }
    TT_LAUNCH_CHECK();
    return 0;
}

int x3_block(const e16* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y, bool planar, int B, int C,
             int H, int T, int d, hipStream_t st) {
    if (C == 16) return planar ? x3_d<16, true>(x, w1, b1, w2, b2, y, B, H, T, d, st) : x3_d<16, false>(x, w1, b1, w2, b2, y, B, H, T, d, st);
    return planar ? x3_d<32, true>(x, w1, b1, w2, b2, y, B, H, T, d, st) : x3_d<32, false>(x, w1, b1, w2, b2, y, B, H, T, d, st);
}
bool x3_shape_ok(int B, int C, int H, int T) { return B > 0 && H > 0 && T > 0 && (C == 16 || C == 32); }      // widths with block kernels
bool x3_layout_ok(int B, int C, int H, int T) { return B > 0 && H > 0 && T > 0 && (C == 16 || C == 32 || C == 64); }   // + the strided layer's output

}  // namespace

extern "C" {

int64_t tt_x3_bytes(int B, int C, int H, int T) {
    if (!x3_layout_ok(B, C, H, T)) return -1;
    return (int64_t)B * H * T * C * 4;
}

int tt_x3_pack(const float* x, void* out, int B, int C, int H, int T, void* stream) {
    if (!x || !out || !x3_layout_ok(B, C, H, T)) return TT_E_BADARG;
    const long npix = (long)B * H * T, pieces = npix * C / 8;
    const unsigned grid = (unsigned)((pieces + NT - 1) / NT);
    hipStream_t st = tt_stream(stream);
    if (C == 16) hipLaunchKernelGGL(k_x3_pack<16>, dim3(grid), dim3(NT), 0, st, x, (e16*)out, H, T, npix);
    else if (C == 64) hipLaunchKernelGGL(k_x3_pack<64>, dim3(grid), dim3(NT), 0, st, x, (e16*)out, H, T, npix);
    else hipLaunchKernelGGL(k_x3_pack<32>, dim3(grid), dim3(NT), 0, st, x, (e16*)out, H, T, npix);
    TT_LAUNCH_CHECK();
    return 0;
}

int tt_x3_unpack(const void* in, float* y, int B, int C, int H, int T, void* stream) {
    if (!in || !y || !x3_layout_ok(B, C, H, T)) return TT_E_BADARG;
    const long npix = (long)B * H * T, pieces = npix * C / 8;
    const unsigned grid = (unsigned)((pieces + NT - 1) / NT);
    hipStream_t st = tt_stream(stream);
    if (C == 16) hipLaunchKernelGGL(k_x3_unpack<16>, dim3(grid), dim3(NT), 0, st, (const e16*)in, y, H, T, npix);
    else if (C == 64) hipLaunchKernelGGL(k_x3_unpack<64>, dim3(grid), dim3(NT), 0, st, (const e16*)in, y, H, T, npix);
    else hipLaunchKernelGGL(k_x3_unpack<32>, dim3(grid), dim3(NT), 0, st, (const e16*)in, y, H, T, npix);
    TT_LAUNCH_CHECK();
    return 0;
}

int tt_x3_rb_fwd(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y, int planar_out, int B,
                 int C, int H, int T, int dilation, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || x == y || !x3_shape_ok(B, C, H, T)) return TT_E_BADARG;
    return x3_block((const e16*)x, w1, b1, w2, b2, y, planar_out != 0, B, C, H, T, dilation, tt_stream(stream));
}

int tt_x3_level_fwd(int nblocks, const void* x, int x3_in, void* y, int x3_out, const float* const* w1, const float* const* b1,
                    const float* const* w2, const float* const* b2, const int* dilations, void* ws, int B, int C, int H, int T,
                    void* stream) {
    if (nblocks < 1 || !x || !y || !w1 || !b1 || !w2 || !b2 || !dilations || !ws || !x3_shape_ok(B, C, H, T)) return TT_E_BADARG;
    for (int i = 0; i < nblocks; ++i)
        if (!w1[i] || !b1[i] || !w2[i] || !b2[i] || dilations[i] < 1 || dilations[i] > 3) return TT_E_BADARG;
    const int64_t bytes = tt_x3_bytes(B, C, H, T);
    unsigned char* buf[2] = {static_cast<unsigned char*>(ws), static_cast<unsigned char*>(ws) + ((bytes + 255) / 256) * 256};
    const void* cur = x;
    if (!x3_in) {
        if (int rc = tt_x3_pack(static_cast<const float*>(x), buf[0], B, C, H, T, stream)) return rc;
        cur = buf[0];
    }
    for (int i = 0; i < nblocks; ++i) {                          // the last block writes into y: x3, or fp32 planar straight away
        const bool last = i == nblocks - 1;
        void* dst = last ? y : (cur == buf[0] ? buf[1] : buf[0]);
        if (dst == cur) return TT_E_BADARG;                      // x3_in with y == x
        if (int rc = tt_x3_rb_fwd(cur, w1[i], b1[i], w2[i], b2[i], dst, last && !x3_out, B, C, H, T, dilations[i], stream)) return rc;
        cur = dst;
    }
    return 0;
}

int tt_x3_sconv_fwd(const void* x, int planar_in, const float* w, const float* bias, void* y, int planar_out, int B, int C, int H, int T,
                    void* stream) {
    if (!x || !w || !bias || !y || B <= 0 || T <= 0 || H < 4) return TT_E_BADARG;
    const int Hout = (H - 4) / 2 + 1;
    hipStream_t st = tt_stream(stream);
    if (planar_in) return C == 8 ? launch_x3s<8, false, true>(x, w, bias, y, planar_out, B, H, Hout, T, st) : TT_E_UNSUPPORTED;
    if (C == 16) return launch_x3s<16, false, false>(x, w, bias, y, planar_out, B, H, Hout, T, st);
    if (C == 32) return launch_x3s<32, false, false>(x, w, bias, y, planar_out, B, H, Hout, T, st);
    return TT_E_UNSUPPORTED;
}

int tt_x3_tconv_fwd(const void* x, int planar_in, const float* w, const float* bias, void* y, int planar_out, int B, int C, int H, int T,
                    int out_pad, void* stream) {
    if (!x || !w || !bias || !y || B <= 0 || H <= 0 || T <= 0 || (out_pad != 0 && out_pad != 1)) return TT_E_BADARG;
    const int Hout = 2 * H + 2 + out_pad;                        // C = output channels; the input has 2 C
    hipStream_t st = tt_stream(stream);
    if (planar_in) return C == 32 ? launch_x3s<64, true, true>(x, w, bias, y, planar_out, B, H, Hout, T, st) : TT_E_UNSUPPORTED;
    if (C == 32) return launch_x3s<64, true, false>(x, w, bias, y, planar_out, B, H, Hout, T, st);
    return C == 16 ? launch_x3s<32, true, false>(x, w, bias, y, planar_out, B, H, Hout, T, st) : TT_E_UNSUPPORTED;
}

int64_t tt_x3_latent_scratch_bytes(int C, int E, int D) {
    if (!((C == 64 && D == 128) || (C == 32 && D == 32)) || E < 1) return -1;
    const int64_t enc = (int64_t)E * (C / 32) * (D / 16) * 2048, dec = (int64_t)E * (D / 32) * (C / 16) * 2048;
    return enc > dec ? enc : dec;
}

int tt_x3_latent_encode(const void* x, const float* w, const float* bias, float* z, void* ws, int B, int C, int E, int D, int T,
                        void* stream) {
    if (!x || !w || !z || !ws || B <= 0 || E <= 0 || T <= 0) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    if (C == 64 && D == 128) return launch_latenc<64, 128>((const e16*)x, w, bias, z, (unsigned char*)ws, B, E, T, st);
    if (C == 32 && D == 32) return launch_latenc<32, 32>((const e16*)x, w, bias, z, (unsigned char*)ws, B, E, T, st);
    return TT_E_UNSUPPORTED;
}

int tt_x3_latent_decode(const float* z, int Dz, float fill, const float* w, const float* bias, void* y, int planar_out, void* ws, int B,
                        int C, int E, int D, int T, void* stream) {
    if (!z || !w || !y || !ws || B <= 0 || E <= 0 || T <= 0 || (Dz != D && Dz != D + 1)) return TT_E_BADARG;
    if ((int64_t)E * C * 4 + 2 * (int64_t)(D / 32) * (C / 16) * 2048 > 160 * 1024) return TT_E_UNSUPPORTED;
    hipStream_t st = tt_stream(stream);
    if (C == 64 && D == 128) return launch_latdec<64, 128>(z, Dz, fill, w, bias, y, planar_out != 0, (unsigned char*)ws, B, E, T, st);
    if (C == 32 && D == 32) return launch_latdec<32, 32>(z, Dz, fill, w, bias, y, planar_out != 0, (unsigned char*)ws, B, E, T, st);
    return TT_E_UNSUPPORTED;
}

int64_t tt_x3_level_scratch_bytes(int B, int C, int H, int T) {
    const int64_t bytes = tt_x3_bytes(B, C, H, T);
    return bytes < 0 ? -1 : 2 * (((bytes + 255) / 256) * 256);
}

int64_t tt_x3n_level_scratch_bytes(int B, int C, int H, int T) {
    if (!x3n_shape_ok(B, C, H, T)) return -1;
    const int64_t bytes = (int64_t)B * H * T * C * 4;
    return 2 * (((bytes + 255) / 256) * 256);
}

int tt_x3n_rb_fwd(const void* x, int planar_in, const float* w1, const float* b1, const float* w2, const float* b2, void* y, int planar_out,
                  int B, int C, int H, int T, int dilation, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || x == y || !x3n_shape_ok(B, C, H, T)) return TT_E_BADARG;
    hipStream_t st = tt_stream(stream);
    return C == 4 ? x3n_d<4>(x, planar_in != 0, w1, b1, w2, b2, y, planar_out != 0, B, H, T, dilation, st)
                  : x3n_d<8>(x, planar_in != 0, w1, b1, w2, b2, y, planar_out != 0, B, H, T, dilation, st);
}

int tt_x3n_level_fwd(int nblocks, const float* x, float* y, const float* const* w1, const float* const* b1, const float* const* w2,
                     const float* const* b2, const int* dilations, void* ws, int B, int C, int H, int T, void* stream) {
    if (nblocks < 1 || !x || !y || !w1 || !b1 || !w2 || !b2 || !dilations || !ws || !x3n_shape_ok(B, C, H, T)) return TT_E_BADARG;
    for (int i = 0; i < nblocks; ++i)
        if (!w1[i] || !b1[i] || !w2[i] || !b2[i] || dilations[i] < 1 || dilations[i] > 3) return TT_E_BADARG;
    const int64_t bytes = (int64_t)B * H * T * C * 4;
    unsigned char* buf[2] = {static_cast<unsigned char*>(ws), static_cast<unsigned char*>(ws) + ((bytes + 255) / 256) * 256};
    const void* cur = x;
    for (int i = 0; i < nblocks; ++i) {                          // first block: fp32 planar in; last block: fp32 planar out
        const bool first = i == 0, last = i == nblocks - 1;
        void* dst = last ? static_cast<void*>(y) : static_cast<void*>(buf[i & 1]);
        if (dst == cur) return TT_E_BADARG;
        if (int rc = tt_x3n_rb_fwd(cur, first, w1[i], b1[i], w2[i], b2[i], dst, last, B, C, H, T, dilations[i], stream)) return rc;
        cur = dst;
    }
    return 0;
}

}  // extern "C"
